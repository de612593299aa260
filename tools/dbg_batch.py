import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.utils import synth
from vface_amd.ldm.models.diffusion.ddpm import FFHQ_UNET_CONFIG, LatentDiffusion
from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
from vface_amd.ldm.models.pnp_utils import register_spa_attn_injection as reg
DEV = "cuda:0"
ldm = LatentDiffusion(dict(FFHQ_UNET_CONFIG)); synth.fill_module_(ldm.unet, seed=0); ldm = ldm.to(DEV)
sampler = DDIMSampler(ldm)
def run(fusion, f0, fc, h=64):
    pick = lambda name, shape: torch.stack([synth.synth_normal(f"cfg.{name}.{c}.{f}", shape) for c in range(3) for f in range(f0, f0 + fc)])
    x, ctx = pick("x", (9, h, h)).to(DEV), pick("c", (1, 768)).to(DEV)
    t = torch.full((3 * fc,), 481, dtype=torch.long, device=DEV)
    reg(sampler, 1, switch_on=False, input_blocks=True, middle_block=True, output_blocks=True, chunks=3)
    if fusion != "plain":
        reg(sampler, 1, switch_on=True, input_blocks=True, middle_block=False, output_blocks=False, chunks=3, flow=None,
            block_indices=list(range(9)), fusion=fusion, split_ratio_fft=0.8, alpha=0.8)
    out = ldm.apply_model(x, t, ctx).float()
    return out.reshape(3, fc, *out.shape[1:]).cpu()
orig = hip.splitk_workspace
def make(filter_fn):
    def f(device, M, N, K, flags=0, rows_per_sample=1):
        if filter_fn(M, N, K, rows_per_sample):
            return orig(device, M, N, K, flags, rows_per_sample)
        return None, 0
    return f
cases = {"conv only (rps>1,K>=9*320)": lambda M, N, K, r: r > 1 and K % 9 == 0 and K >= 2880,
         "rps==1 gemms only": lambda M, N, K, r: r <= 1,
         "rps>1 gemms only": lambda M, N, K, r: r > 1 and not (K % 9 == 0 and K >= 2880)}
for name, fn in cases.items():
    hip.splitk_workspace = make(fn)
    w = run("plain", 0, 16); p = run("plain", 0, 8)
    print(f"{name:30s}: max|diff| = {(w[:, :8] - p).abs().max().item():.3e}", flush=True)
seen = set()
def spy(device, M, N, K, flags=0, rows_per_sample=1):
    r = orig(device, M, N, K, flags, rows_per_sample)
    if r[1]: seen.add((M, N, K, rows_per_sample, hip.load().vface_splitk_workspace_bytes(M, N, K, flags, rows_per_sample) // (M * N * 4)))
    return r
hip.splitk_workspace = spy
run("plain", 0, 8); print("F=8 splits:", sorted(seen)); seen.clear()
run("plain", 0, 16); print("F=16 splits:", sorted(seen))
