#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself (read-only at
/root/reference) on CPU in the build container.  The reference never travels: only inputs' seeds and the
resulting tensors are committed.  Re-run with:  python tests/golden/make_golden.py [--full]

Third-party packages the reference imports but this image lacks (torchvision, kornia, omegaconf; no network)
are replaced by empty stand-ins that contain no reference code: RAFT is never called, and
``kornia.utils.create_meshgrid`` is the documented pixel-coordinate (x, y) grid.

Inputs are produced by ``vface_amd.utils.synth`` (name-keyed deterministic fill), so tests regenerate them
instead of storing them.
"""
import argparse
import os
import sys
import tempfile
import time
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from vface_amd.utils import synth  # noqa: E402

REF = "/root/reference/REFace"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    class _NoRaft(torch.nn.Module):
        def forward(self, *a, **k):
            raise RuntimeError("RAFT stand-in: optical flow is supplied synthetically")

    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.transforms.functional = _stub("torchvision.transforms.functional")
    tv.io = _stub("torchvision.io", read_video=None, write_video=None)
    tv.models = _stub("torchvision.models")
    tv.models.optical_flow = _stub("torchvision.models.optical_flow", raft_large=lambda **k: _NoRaft())
    tv.utils = _stub("torchvision.utils", flow_to_image=None)
    oc = _stub("omegaconf")
    oc.listconfig = _stub("omegaconf.listconfig", ListConfig=type("ListConfig", (list,), {}))

    def create_meshgrid(H, W, normalized_coordinates=False, device=None, dtype=None):
        assert not normalized_coordinates
        xs = torch.linspace(0, W - 1, W)
        ys = torch.linspace(0, H - 1, H)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        return torch.stack([gx, gy], -1)[None]

    k = _stub("kornia")
    k.utils = _stub("kornia.utils", create_meshgrid=create_meshgrid)
    sys.path.insert(0, REF)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    sys.stdout.write(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)\n")


from cases import make_flows as _make_flows  # noqa: E402


def make_flows(pairs, h, w):
    return _make_flows(h, w)


@torch.no_grad()
def gen_fsai():
    import scripts.face_swap_utils as fsu
    out = {}
    for d in (320, 640, 1280):
        for ratio in (0.8, 0.5):
            q1 = synth.synth_normal(f"fsai.q1.{d}", (2, 5, d), seed=1)
            q2 = synth.synth_normal(f"fsai.q2.{d}", (2, 5, d), seed=2)
            out[f"d{d}_r{ratio}"] = fsu.combine_fft_high_low(q1, q2, split_ratio=ratio)
            out[f"d{d}_r{ratio}_h"] = fsu.combine_fft_high_low(q1.half(), q2.half(), split_ratio=ratio)
    save("fsai", **out)


@torch.no_grad()
def gen_warp():
    import scripts.temporal_flow as tf
    import torch.nn.functional as F
    h = w = 64
    cases = make_flows(1, h, w)
    img = synth.synth_normal("warp.img", (3, 8, h, w), seed=3)
    out = {}
    captured = {}
    real_gs = F.grid_sample

    def spy(inp, grid, **kw):
        captured["grid"] = grid.clone()
        return real_gs(inp, grid, **kw)

    for name, fl in cases.items():
        flow = torch.from_numpy(fl)[None]
        tf.F.grid_sample = spy
        try:
            o = tf.warp_image(img[:1], flow)
        finally:
            tf.F.grid_sample = real_gs
        out[f"warp_{name}"] = o[0]
        g = captured["grid"][0]  # [H,W,2] normalised grid as the reference computed it
        # ATen grid_sampler (align_corners=True, border): unnormalise, clamp, floor
        ix = ((g[..., 0] + 1.0) / 2.0) * float(w - 1)
        iy = ((g[..., 1] + 1.0) / 2.0) * float(h - 1)
        ix = ix.clamp(0.0, float(w - 1))
        iy = iy.clamp(0.0, float(h - 1))
        out[f"x0_{name}"] = torch.floor(ix).to(torch.int32)
        out[f"y0_{name}"] = torch.floor(iy).to(torch.int32)
        out[f"ix_{name}"] = ix
        out[f"iy_{name}"] = iy
    flows = [torch.from_numpy(cases["pm3"])[None], torch.from_numpy(cases["smooth"])[None]]
    out["align_a0.8"] = tf.align_by_flow(img, flow=flows, alpha=0.8)
    out["align_a0.5"] = tf.align_by_flow(img, flow=flows, alpha=0.5)
    save("warp", **out)


def gen_warp_cuda_form():
    """Gather indices of the six flow cases under CUDA-ATen's scalar division (multiply by the fp32 reciprocal) -- EMULATED
    by the oracle (oracle/flow.py::sample_coords(cuda_recip_div=True)): the reference cannot run on its native device in
    this container, so this fixture pins the kernel's `flags bit 0` path to the restated rule, not to a reference run."""
    from oracle import flow as oflow
    out = {}
    for name, fl in make_flows(1, 64, 64).items():
        f = torch.from_numpy(fl)
        x0, y0 = oflow.gather_indices(f, cuda_recip_div=True)
        cx0, cy0 = oflow.gather_indices(f, cuda_recip_div=False)
        out[f"x0_{name}"], out[f"y0_{name}"] = x0, y0
        out[f"ndiff_{name}"] = int(((x0 != cx0) | (y0 != cy0)).sum())
    save("warp_cuda_form", **out)


def ref_unet(model_channels, fill_seed=0):
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    m = UNetModel(image_size=32, in_channels=9, out_channels=4, model_channels=model_channels,
                  attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8,
                  use_spatial_transformer=True, transformer_depth=1, context_dim=768, use_checkpoint=True,
                  legacy=False, add_conv_in_front_of_unet=False)
    synth.fill_module_(m, seed=fill_seed)
    return m.eval()


class FakeLDM:
    """Stands in for LatentDiffusion: the sampler only touches these attributes."""

    def __init__(self, unet):
        from ldm.modules.diffusionmodules.util import make_beta_schedule
        betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
        self.num_timesteps = 1000
        self.device = torch.device("cpu")
        self.model = types.SimpleNamespace(diffusion_model=unet)
        self.parameterization = "eps"

    def apply_model(self, x, t, c):
        return self.model.diffusion_model(x, t, context=c)


def make_sampler(unet):
    import ldm.models.diffusion.ddim_w_inv as dd

    class CpuSampler(dd.DDIMSampler):
        def register_buffer(self, name, attr):  # the reference forces .to("cuda") here (:149-153)
            setattr(self, name, attr)

    return CpuSampler(FakeLDM(unet)), dd


def unet_inputs(F_, h, w, tag):
    x = synth.synth_normal(f"{tag}.x", (3 * F_, 9, h, w))
    ctx = synth.synth_normal(f"{tag}.ctx", (3 * F_, 1, 768))
    return x, ctx


@torch.no_grad()
def gen_tiny_unet():
    import ldm.models.pnp_utils as pnp
    unet = ref_unet(32)
    sampler, dd = make_sampler(unet)
    F_, h, w = 2, 64, 64
    x, ctx = unet_inputs(F_, h, w, "tiny")
    t = torch.full((3 * F_,), 481, dtype=torch.long)
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    out = {}
    # ordinal table (vi)
    for g in ("input_blocks", "middle_block", "output_blocks"):
        _, names = pnp.find_all_modules_by_name(getattr(unet, g), "attn1")
        out[f"names_{g}"] = np.array(names)
    out["plain"] = unet(x, t, context=ctx)
    all_idx = list(range(9))

    def off():
        pnp.register_spa_attn_injection(sampler, 1, switch_on=False, input_blocks=True, middle_block=True,
                                        output_blocks=True, attn_component="attn1", chunks=3)

    off()
    out["off"] = unet(x, t, context=ctx)
    for fusion in ("replace", "fft", "flow_fix", "temporal", "adaIn", "mix", "fft_vfixed"):
        off()
        pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=True, middle_block=False,
                                        output_blocks=False, attn_component="attn1", flow=flow, chunks=3,
                                        block_indices=all_idx, fusion=fusion, split_ratio_fft=0.8, alpha=0.8)
        out[f"in_{fusion}"] = unet(x, t, context=ctx)
    # the pre-loop registration of the shipped sampler (fft on the 9 output blocks), and all groups on
    off()
    pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=False, middle_block=False,
                                    output_blocks=True, attn_component="attn1", chunks=3, block_indices=all_idx,
                                    fusion="fft", split_ratio_fft=0.8, alpha=0.8)
    out["out_fft"] = unet(x, t, context=ctx)
    off()
    pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=True, middle_block=True,
                                    output_blocks=True, attn_component="attn1", chunks=3, block_indices=[0, 2, 5],
                                    fusion="replace")
    out["sel_replace_025"] = unet(x, t, context=ctx)
    # chunks == 2 (inversion-time variant, pnp_utils.py:259-262)
    off()
    pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=True, middle_block=False,
                                    output_blocks=True, attn_component="attn1", chunks=2, block_indices=[0, 1, 2])
    out["chunks2"] = unet(x[:4], t[:4], context=ctx[:4])
    save("tiny_unet", **out)


@torch.no_grad()
def gen_attn_module():
    """Hooked attn1 at the real level-0 shape (d=320, 8 heads, n=4096), strided token slice of the output."""
    from ldm.modules.attention import CrossAttention
    import ldm.models.pnp_utils as pnp
    attn = CrossAttention(query_dim=320, heads=8, dim_head=40).eval()
    synth.fill_module_(attn, seed=0, prefix="attn1.")
    holder = torch.nn.Module()
    holder.blk = torch.nn.Module()
    holder.blk.attn1 = attn
    unet = types.SimpleNamespace(input_blocks=holder, middle_block=torch.nn.Module(), output_blocks=torch.nn.Module())
    sampler = types.SimpleNamespace(model=types.SimpleNamespace(model=types.SimpleNamespace(diffusion_model=unet)))
    F_, n, d = 2, 4096, 320
    x = synth.synth_normal("attnmod.x", (3 * F_, n, d))
    flow = [synth.synth_flow(F_ - 1, 64, 64)[i][None] for i in range(F_ - 1)]
    out = {"plain": attn(x)[:, ::128]}
    for fusion in ("replace", "fft", "flow_fix"):
        pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=True, middle_block=False,
                                        output_blocks=False, attn_component="attn1", flow=flow, chunks=3,
                                        block_indices=None, fusion=fusion, split_ratio_fft=0.8, alpha=0.8)
        out[fusion] = attn(x)[:, ::128]
    save("attn_module", **out)


@torch.no_grad()
def gen_ddim():
    unet = ref_unet(32)
    sampler, dd = make_sampler(unet)
    F_, h, w = 2, 64, 64
    out = {}
    for S in (50, 20, 25):
        sampler.make_schedule(ddim_num_steps=S, ddim_eta=0.0, verbose=False)
        out[f"S{S}_timesteps"] = np.asarray(sampler.ddim_timesteps)
        out[f"S{S}_alphas"] = np.asarray(sampler.ddim_alphas, dtype=np.float64)
        out[f"S{S}_alphas_prev"] = np.asarray(sampler.ddim_alphas_prev, dtype=np.float64)
        out[f"S{S}_sqrt_1m"] = np.asarray(sampler.ddim_sqrt_one_minus_alphas, dtype=np.float64)
        out[f"S{S}_sigmas"] = np.asarray(sampler.ddim_sigmas, dtype=np.float64)
    out["alphas_cumprod"] = sampler.alphas_cumprod
    x_T = synth.synth_normal("ddim.xT", (F_, 4, h, w))
    c = synth.synth_normal("ddim.c", (F_, 1, 768))
    uc = synth.synth_normal("ddim.uc", (F_, 1, 768))
    tc = synth.synth_normal("ddim.tc", (F_, 1, 768))
    inp = synth.synth_normal("ddim.inpaint", (F_, 4, h, w)) * 0.18215
    mask = synth.synth_mask(F_, h, w)
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    kwargs = {"inpaint_image": inp, "inpaint_mask": mask}
    with tempfile.TemporaryDirectory() as td:
        sampler.make_schedule(ddim_num_steps=50, ddim_eta=0.0, verbose=False)
        for step in sampler.ddim_timesteps:
            torch.save(synth.synth_normal(f"ddim.inv.{int(step)}", (F_, 4, h, w)),
                       os.path.join(td, f"ddim_latents_{int(step)}.pt"))
        # (v) a 3-step run of the shipped loop (hooks: all off, then input blocks flow_fix; :303,:305)
        import ldm.models.diffusion.ddim_w_inv as ddm
        real_tqdm = ddm.tqdm
        ddm.tqdm = lambda it, **k: list(it)[:3]
        try:
            img, inter = sampler.sample(S=50, batch_size=F_, shape=[4, h, w], conditioning=c,
                                        target_conditioning=tc, inverse_results_dir=td, verbose=False,
                                        unconditional_guidance_scale=3.0, unconditional_conditioning=uc, eta=0.0,
                                        x_T=x_T, flow=flow, test_model_kwargs=kwargs, log_every_t=1)
        finally:
            ddm.tqdm = real_tqdm
        out["sample3_final"] = img
        out["sample3_x_inter"] = torch.stack(inter["x_inter"][1:])
        out["sample3_pred_x0"] = torch.stack(inter["pred_x0"][1:])
        # inversion: 2 steps, batch 2F (target ; source), saves the target half
        x0 = synth.synth_normal("ddim.z2", (2 * F_, 4, h, w))
        cond2 = torch.cat([tc, c], 0)
        kw2 = {"inpaint_image": torch.cat([inp, inp], 0), "inpaint_mask": torch.cat([mask, mask], 0)}
        inv_dir = os.path.join(td, "inv")
        os.makedirs(inv_dir)
        ddm.tqdm = lambda it, **k: list(it)[:2]
        try:
            xn, _ = sampler.ddim_invert(x=x0, cond=cond2, S=50, shape=[4, h, w], eta=0.0,
                                        unconditional_guidance_scale=3.0, unconditional_conditioning=None,
                                        inverse_dir=inv_dir, batch_size=F_, test_model_kwargs=kw2)
        finally:
            ddm.tqdm = real_tqdm
        out["invert2_final"] = xn
        out["invert2_saved_1"] = torch.load(os.path.join(inv_dir, "ddim_latents_1.pt"))
        out["invert2_saved_21"] = torch.load(os.path.join(inv_dir, "ddim_latents_21.pt"))
    save("ddim", **out)


@torch.no_grad()
def gen_full_unet():
    """The real 859.5 M-parameter configuration (project_ffhq.yaml:33-56), F=2 at 64x64, shipped hook
    schedule (input-block attn1 flow_fix).  ~4 GB of weights, about a minute on 8 cores."""
    import ldm.models.pnp_utils as pnp
    t0 = time.time()
    unet = ref_unet(320)
    print(f"full unet built+filled in {time.time() - t0:.1f}s; params "
          f"{sum(p.numel() for p in unet.parameters()) / 1e6:.1f} M")
    sampler, dd = make_sampler(unet)
    F_, h, w = 2, 64, 64
    x, ctx = unet_inputs(F_, h, w, "full")
    t = torch.full((3 * F_,), 481, dtype=torch.long)
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    out = {}
    t0 = time.time()
    out["plain"] = unet(x, t, context=ctx)
    out["plain_seconds"] = time.time() - t0
    pnp.register_spa_attn_injection(sampler, 1, switch_on=False, input_blocks=True, middle_block=True,
                                    output_blocks=True, attn_component="attn1", chunks=3)
    pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=True, middle_block=False,
                                    output_blocks=False, attn_component="attn1", flow=flow, chunks=3,
                                    block_indices=list(range(9)), fusion="flow_fix", split_ratio_fft=0.8, alpha=0.8)
    t0 = time.time()
    out["flow_fix"] = unet(x, t, context=ctx)
    out["flow_fix_seconds"] = time.time() - t0
    pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=True, middle_block=False,
                                    output_blocks=False, attn_component="attn1", flow=None, chunks=3,
                                    block_indices=list(range(9)), fusion="replace")
    out["replace"] = unet(x, t, context=ctx)
    out["threads"] = torch.get_num_threads()
    save("full_unet", **out)


@torch.no_grad()
def gen_full_unet_fft():
    """Round 4: BASELINE config 3's own schedule -- frequency-spectrum attention interpolation alone
    (fusion="fft" on the input-block attn1) -- on the real 859.5 M-parameter UNet: the fp32 output, the output under fp16
    autocast and the output with fp16-rounded weights, ADDED to full_unet.npz / lowp.npz (the other entries are kept as they
    are: same seeded inputs, ``full.x`` / ``full.ctx``)."""
    import ldm.models.pnp_utils as pnp
    unet = ref_unet(320)
    sampler, dd = make_sampler(unet)
    F_, h, w = 2, 64, 64
    x, ctx = unet_inputs(F_, h, w, "full")
    t = torch.full((3 * F_,), 481, dtype=torch.long)

    def hook():
        pnp.register_spa_attn_injection(sampler, 1, switch_on=False, input_blocks=True, middle_block=True,
                                        output_blocks=True, attn_component="attn1", chunks=3)
        pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=True, middle_block=False,
                                        output_blocks=False, attn_component="attn1", flow=None, chunks=3,
                                        block_indices=list(range(9)), fusion="fft", split_ratio_fft=0.8, alpha=0.8)
    full = dict(np.load(os.path.join(HERE, "full_unet.npz"), allow_pickle=False))
    lowp = dict(np.load(os.path.join(HERE, "lowp.npz"), allow_pickle=False))
    hook()
    full["fft"] = unet(x, t, context=ctx).numpy()
    hook()
    with torch.autocast("cpu", dtype=torch.float16):
        lowp["full.fft_autocast_f16"] = unet(x, t, context=ctx).float().numpy()
    _round_weights_(unet)
    hook()
    lowp["full.fft_w16"] = unet(x, t, context=ctx).numpy()
    save("full_unet", **full)
    save("lowp", **lowp)


@torch.no_grad()
def gen_full_unet_sampler_batch():
    """Round 6: the 859.5 M-parameter UNet on the batch the reference's SAMPLER assembles
    (ddim_w_inv.py:632-655: ``x_in = cat([x, x, ddim_inv_t])`` with ``inpaint_image`` / ``inpaint_mask`` appended, ``t_in = cat([t] * 3)``,
    ``c_in = cat([uc, c, tc])``) under fusion="fft" on the input-block attn1 -- chunks 0 and 1 enter with identical inputs, which is
    what the build's shared uncond/cond prefix relies on.  F = 2 at 64 x 64; fp32 eps -> full_unet_sampler_batch.npz."""
    import ldm.models.pnp_utils as pnp
    unet = ref_unet(320)
    sampler, dd = make_sampler(unet)
    F_, h, w = 2, 64, 64
    x = synth.synth_normal("sb.x", (F_, 4, h, w))
    inv = synth.synth_normal("sb.inv", (F_, 4, h, w))
    inp = synth.synth_normal("sb.inp", (F_, 4, h, w)) * 0.18215
    mask = synth.synth_mask(F_, h, w)
    ctx = synth.synth_normal("sb.ctx", (3 * F_, 1, 768))
    x_in = torch.cat([x, x, inv])
    x_in = torch.cat([x_in, torch.cat([inp] * 3), torch.cat([mask] * 3)], dim=1)          # (:632-633, :654-655)
    t = torch.full((3 * F_,), 481, dtype=torch.long)
    pnp.register_spa_attn_injection(sampler, 1, switch_on=False, input_blocks=True, middle_block=True,
                                    output_blocks=True, attn_component="attn1", chunks=3)
    pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=True, middle_block=False,
                                    output_blocks=False, attn_component="attn1", flow=None, chunks=3,
                                    block_indices=list(range(9)), fusion="fft", split_ratio_fft=0.8, alpha=0.8)
    save("full_unet_sampler_batch", fft=unet(x_in, t, context=ctx).numpy())


def _round_weights_(module, dtype=torch.float16):
    for prm in module.parameters():
        prm.data = prm.data.to(dtype).float()


@torch.no_grad()
def gen_lowp(full=False):
    """Evidence for the whole-network tolerance: the REFERENCE's own low-precision behaviour, made by the
    reference itself.  For each case two outputs beside the fp32 ones of tiny_unet.npz / full_unet.npz:
      *_autocast_f16  the reference UNet under ``torch.autocast("cpu", dtype=torch.float16)`` -- the arithmetic the shipped
                      entry point runs it in (VFace_inference_batch.py:407-408 autocast) with CPU-ATen kernels;
      *_w16           the reference UNet in fp32 arithmetic with every parameter rounded to fp16 first: the error floor of
                      ANY implementation that feeds fp16 weights to the matrix cores.
    Only outputs are stored (same seeded inputs as the fp32 fixtures)."""
    import ldm.models.pnp_utils as pnp
    out = {}
    for tag, mc in (("tiny", 32),) + ((("full", 320),) if full else ()):
        unet = ref_unet(mc)
        sampler, dd = make_sampler(unet)
        F_, h, w = 2, 64, 64
        x, ctx = unet_inputs(F_, h, w, tag)
        t = torch.full((3 * F_,), 481, dtype=torch.long)
        flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]

        def hook(fusion):
            pnp.register_spa_attn_injection(sampler, 1, switch_on=False, input_blocks=True, middle_block=True,
                                            output_blocks=True, attn_component="attn1", chunks=3)
            if fusion is not None:
                pnp.register_spa_attn_injection(sampler, 1, switch_on=True, input_blocks=True, middle_block=False,
                                                output_blocks=False, attn_component="attn1",
                                                flow=flow if fusion == "flow_fix" else None, chunks=3,
                                                block_indices=list(range(9)), fusion=fusion, split_ratio_fft=0.8, alpha=0.8)

        modes = (("plain", None), ("flow_fix", "flow_fix"), ("replace", "replace"))
        for name, fusion in modes:
            hook(fusion)
            with torch.autocast("cpu", dtype=torch.float16):
                out[f"{tag}.{name}_autocast_f16"] = unet(x, t, context=ctx).float()
        _round_weights_(unet)
        for name, fusion in modes:
            hook(fusion)
            out[f"{tag}.{name}_w16"] = unet(x, t, context=ctx)
    save("lowp", **out)


def gen_vae():
    """First-stage KL-VAE: the reference's own Encoder / Decoder (diffusionmodules/model.py:368-568) and
    DiagonalGaussianDistribution, wired as AutoencoderKL.encode / decode wires them (autoencoder.py:301-302,323-333 --
    the class itself derives from pytorch_lightning, absent here): a small configuration with every block type, and
    the shipped configuration (project_ffhq.yaml:57-78) on a 64x64 image."""
    from ldm.modules.diffusionmodules.model import Decoder, Encoder
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    import builtins
    out, lowp = {}, {}
    for tag, dd, res, nb in (("small", dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32,
                                            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0), 32, 2),
                             ("ffhq", dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                                           ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0), 64, 1)):
        class KL(torch.nn.Module):
            def __init__(self):
                super().__init__()
                pr, builtins.print = builtins.print, (lambda *a, **k: None)   # the constructors print
                self.encoder, self.decoder = Encoder(**dd), Decoder(**dd)
                builtins.print = pr
                self.quant_conv = torch.nn.Conv2d(2 * dd["z_channels"], 2 * 4, 1)
                self.post_quant_conv = torch.nn.Conv2d(4, dd["z_channels"], 1)
        m = KL().eval()
        synth.fill_module_(m, seed=0, prefix="vae.")
        x = synth.synth_normal(f"vae.{tag}.x", (nb, 3, res, res)).clamp(-1, 1)
        noise = synth.synth_normal(f"vae.{tag}.noise", (nb, 4, res // 8, res // 8))
        with torch.no_grad():
            moments = m.quant_conv(m.encoder(x))
            post = DiagonalGaussianDistribution(moments)
            z_mode = post.mode() * 0.18215
            z_samp = (post.mean + post.std * noise) * 0.18215      # sample() with the noise drawn outside
            dec = m.decoder(m.post_quant_conv(z_samp / 0.18215))
        out[f"{tag}.moments"], out[f"{tag}.z_mode"], out[f"{tag}.z_sample"], out[f"{tag}.dec"] = moments, z_mode, z_samp, dec
        out[f"{tag}.n_params"] = float(sum(p.numel() for p in m.parameters()))
        # The reference's own LOW-PRECISION behaviour on the same inputs (the evidence the GPU test's whole-network bound
        # rests on, as lowp.npz is for the UNet): under torch.autocast(fp16) -- what the entry point runs the first stage in
        # (VFace_inference_batch.py:407-408 wraps encode / decode too) -- and in fp32 arithmetic with fp16-rounded parameters.
        def run(mm):
            mo = mm.quant_conv(mm.encoder(x)).float()
            po = DiagonalGaussianDistribution(mo)
            return po.mode() * 0.18215, (po.mean + po.std * noise) * 0.18215, mm.decoder(mm.post_quant_conv(z_samp / 0.18215)).float()
        with torch.no_grad():
            with torch.autocast("cpu", dtype=torch.float16):
                a_mode, a_samp, a_dec = run(m)
            _round_weights_(m)
            w_mode, w_samp, w_dec = run(m)
        for k, v in (("z_mode_autocast_f16", a_mode), ("z_sample_autocast_f16", a_samp), ("dec_autocast_f16", a_dec),
                     ("z_mode_w16", w_mode), ("z_sample_w16", w_samp), ("dec_w16", w_dec)):
            lowp[f"{tag}.{k}"] = v.float()
    save("vae", **out)
    save("vae_lowp", **lowp)


def gen_paste():
    """Paste-back (VFace_inference_batch.py:597-636): what PILLOW (the reference's own dependency, 12.2.0 in this image) returns
    for that block's calls on seeded inputs -- `Image.resize(.., BILINEAR)` up / down / one axis, and `putalpha(255)` +
    `transform(size, PERSPECTIVE, coeffs, BILINEAR)` + `alpha_composite` for three quads (inside, crossing the border, covering)."""
    import PIL
    from PIL import Image
    out = {"pillow_version": np.array(PIL.__version__)}
    rng = np.random.default_rng(20260401)
    for i, (h, w, ow, oh) in enumerate([(32, 32, 64, 64), (50, 70, 33, 91), (37, 41, 100, 17), (64, 48, 48, 64), (96, 96, 192, 192)]):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out[f"resize{i}.in"] = img
        out[f"resize{i}.size"] = np.array([ow, oh])
        out[f"resize{i}.out"] = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    sw = sh = 64
    crop = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    bg = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    out["persp.crop"], out["persp.bg"] = crop, bg
    quads = [[(20.3, 15.1), (80.7, 18.2), (85.1, 70.9), (18.5, 66.5)], [(-15, -8), (75, 4), (70, 105), (8, 68)], [(0, 0), (120, 0), (120, 90), (0, 90)]]
    for i, quad in enumerate(quads):
        A, B = [], []
        for (x, y), (u, v) in zip(quad, [(0, 0), (sw, 0), (sw, sh), (0, sh)]):
            A += [[x, y, 1, 0, 0, 0, -u * x, -u * y], [0, 0, 0, x, y, 1, -v * x, -v * y]]
            B += [u, v]
        co = np.linalg.solve(np.array(A, float), np.array(B, float))
        s = Image.fromarray(crop).convert("RGBA")
        pasted = Image.fromarray(bg).convert("RGBA")
        s.putalpha(255)
        pasted.alpha_composite(s.transform((120, 90), Image.PERSPECTIVE, co, Image.BILINEAR))
        out[f"persp{i}.coeffs"] = co
        out[f"persp{i}.out"] = np.asarray(pasted)
    save("paste", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also the 860 M-parameter UNet fixture")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    install_stubs()
    import builtins
    _print = builtins.print
    gens = {"fsai": gen_fsai, "warp": gen_warp, "warp_cuda": gen_warp_cuda_form, "attn": gen_attn_module, "tiny": gen_tiny_unet, "ddim": gen_ddim,
            "vae": gen_vae, "paste": gen_paste}
    gens["lowp"] = lambda: gen_lowp(a.full)
    if a.full:
        gens["full"] = gen_full_unet
        gens["full_fft"] = gen_full_unet_fft
        gens["full_sampler_batch"] = gen_full_unet_sampler_batch
    for name, fn in gens.items():
        if a.only and name not in a.only.split(","):
            continue
        t0 = time.time()
        builtins.print = lambda *x, **k: None  # the reference prints inside the hook (pnp_utils.py:143,222)
        try:
            fn()
        finally:
            builtins.print = _print
        print(f"{name}: {time.time() - t0:.1f}s")
