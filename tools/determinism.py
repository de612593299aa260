#!/usr/bin/env python3
"""Run the small hooked UNet several times with different batch sizes and report bitwise reproducibility."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
from vface_amd.ldm.models.diffusion.ddpm import LatentDiffusion
from vface_amd.ldm.models.pnp_utils import register_spa_attn_injection as reg
from vface_amd.utils import synth
cfg = dict(image_size=32, in_channels=9, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
           channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768, legacy=False)
dev = "cuda:0"
ldm = LatentDiffusion(cfg); synth.fill_module_(ldm.unet, seed=0); ldm = ldm.to(dev)
sampler = DDIMSampler(ldm)
total, h, w = 4, 32, 32
xs = [synth.synth_normal(f"shard.x.{c}", (total, 9, h, w)) for c in range(3)]
cs = [synth.synth_normal(f"shard.c.{c}", (total, 1, 768)) for c in range(3)]
mode = sys.argv[1] if len(sys.argv) > 1 else "off"
def run(f0, fc):
    x = torch.cat([t[f0:f0 + fc] for t in xs]).to(dev); ctx = torch.cat([t[f0:f0 + fc] for t in cs]).to(dev)
    tt = torch.full((3 * fc,), 481, dtype=torch.long, device=dev)
    reg(sampler, 1, switch_on=False, input_blocks=True, middle_block=True, output_blocks=True)
    if mode != "off":
        reg(sampler, 1, switch_on=True, input_blocks=True, middle_block=False, output_blocks=False, chunks=3, block_indices=list(range(9)), fusion=mode)
    return ldm.apply_model(x, tt, ctx).float().cpu()
a = run(0, 4); b = run(0, 4)
print("same batch twice: bitwise equal =", torch.equal(a, b), "max diff", float((a - b).abs().max()))
c = run(0, 2)
ref = torch.cat([a[k * 4:k * 4 + 2] for k in range(3)])
print("F=2 vs slices of F=4: equal =", torch.equal(c, ref), "max diff", float((c - ref).abs().max()))
# layer-level: trace where they start to differ
