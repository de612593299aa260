"""Does any kernel read memory nobody wrote?  Poison the caching allocator's pool with NaN / large values, then run the
small UNet (all hook modes) twice and compare with a run on a zeroed pool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd.utils import synth
from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
from vface_amd.ldm.models.diffusion.ddpm import LatentDiffusion
from vface_amd.ldm.models.pnp_utils import register_spa_attn_injection as reg
DEV = "cuda:0"
cfg = dict(image_size=32, in_channels=9, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
           channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768, legacy=False)
ldm = LatentDiffusion(cfg); synth.fill_module_(ldm.unet, seed=0); ldm = ldm.to(DEV); sampler = DDIMSampler(ldm)
def poison(val):
    t = torch.full((3 * 2 ** 28,), val, dtype=torch.float32, device=DEV)   # 3 GiB block, returned to the pool (not the driver)
    del t
def run(F_, fusion):
    h = w = 32
    x = torch.cat([synth.synth_normal(f"shard.x.{c}", (4, 9, h, w))[:F_] for c in range(3)]).to(DEV)
    ctx = torch.cat([synth.synth_normal(f"shard.c.{c}", (4, 1, 768))[:F_] for c in range(3)]).to(DEV)
    tt = torch.full((3 * F_,), 481, dtype=torch.long, device=DEV)
    flow = synth.synth_flow(3, h, w)[:F_ - 1]
    reg(sampler, 1, switch_on=False, input_blocks=True, middle_block=True, output_blocks=True)
    reg(sampler, 1, switch_on=True, input_blocks=True, middle_block=False, output_blocks=False, chunks=3,
        flow=[f[None] for f in flow] if fusion == "flow_fix" else None, block_indices=list(range(9)), fusion=fusion)
    return ldm.apply_model(x, tt, ctx).float().cpu()
for fusion in ("flow_fix", "replace"):
    for F_ in (2, 4):
        poison(0.0); a = run(F_, fusion)
        poison(float("nan")); b = run(F_, fusion)
        poison(3.0e4); c = run(F_, fusion)
        print(fusion, F_, "zero-vs-nan equal:", torch.equal(a, b), "nan in out:", bool(torch.isnan(b).any()),
              "zero-vs-3e4 equal:", torch.equal(a, c), "max diff", (a - c).abs().max().item(), flush=True)
