"""Whole-path parity on the MI355X: the hooked UNet and the DDIM loop through the product modules (HIP kernels)
against the CPU oracle on the same seeded inputs, and against the reference-generated full-size fixture.

Tolerance.  The north-star asks for rel-L2 <= 1e-3 against the reference.  Per kernel that is asserted
(tests/test_kernels_gpu.py).  For a whole 16-bit network it is not reachable by ANY implementation that feeds fp16
weights to the matrix cores, and the fixtures the reference itself produced show it (tests/golden/lowp.npz,
make_golden.py::gen_lowp): the reference UNet in fp32 arithmetic with nothing but its parameters rounded to fp16 is
already 9.6e-4 (full UNet) / 1.1e-3 (small configuration) from its fp32 output, and the reference in its own shipped
arithmetic (fp16 autocast) is 1.59e-3 / 2.06e-3 away.  What is asserted here instead, per case:
  * the HIP path is CLOSER to the fp32 reference than the reference's own fp16-autocast run (fixture, not emulation);
  * an absolute bound a little above the measured value (fp32 residual stream: DESIGN 6), so a regression shows."""
import os

import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import ddim as oddim
from oracle import hooks as ohooks
from oracle import unet as ounet
from vface_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
SMALL = ounet.UNetSpec(model_channels=64)


def _diff_pattern(a, b):
    """Which frames / how many elements / how large: enough to tell a stale or torn boundary slab (frame F/2 first, later frames
    only after further DDIM steps; rounding-sized) from an ordering bug in a half's own work (any frame of that half) at one look."""
    d = (a.double() - b.double()).abs()
    per = [(i, int((d[i] > 0).sum()), float(d[i].max()), float(d[i].max() / b[i].double().abs().max().clamp_min(1e-30)))
           for i in range(a.shape[0]) if bool((d[i] > 0).any())]
    return "; ".join(f"frame {i}: {n} elements differ, max abs {m:.3e} (rel {r:.1e})" for i, n, m, r in per) or "equal"


def small_cfg(mc=64):
    return dict(image_size=32, in_channels=9, out_channels=4, model_channels=mc, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                transformer_depth=1, context_dim=768, legacy=False)


@pytest.fixture(scope="module")
def small():
    from vface_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
    ldm = LatentDiffusion(small_cfg())
    synth.fill_module_(ldm.unet, seed=0)
    sd = {k: v.clone() for k, v in ldm.unet.state_dict().items()}
    ldm = ldm.to(DEV)
    sampler = DDIMSampler(ldm)
    sampler.flow_gate = "flow_hw"   # 32 x 32 latents: the reference's n == 4096 test would never warp (see the gate test below)
    return ldm, sampler, sd


def _register(sampler, mode, flow):
    from vface_amd.ldm.models.pnp_utils import register_spa_attn_injection as reg
    reg(sampler, 1, switch_on=False, input_blocks=True, middle_block=True, output_blocks=True, chunks=3)
    if mode == "off":
        return
    kw = dict(switch_on=True, attn_component="attn1", flow=flow, split_ratio_fft=0.8, alpha=0.8)
    if mode.startswith("in_"):
        reg(sampler, 1, input_blocks=True, middle_block=False, output_blocks=False, chunks=3,
            block_indices=list(range(9)), fusion=mode[3:], **kw)
    elif mode == "out_fft":
        reg(sampler, 1, input_blocks=False, middle_block=False, output_blocks=True, chunks=3,
            block_indices=list(range(9)), fusion="fft", **kw)
    elif mode == "sel_replace_025":
        reg(sampler, 1, input_blocks=True, middle_block=True, output_blocks=True, chunks=3, block_indices=[0, 2, 5],
            fusion="replace", **kw)
    elif mode == "chunks2":
        reg(sampler, 1, input_blocks=True, middle_block=False, output_blocks=True, chunks=2, block_indices=[0, 1, 2], **kw)


def _oracle_registry(mode, flow):
    names = ounet.attn1_names(SMALL)
    r = {}
    ohooks.register_spa_attn_injection(r, names, 1, switch_on=False, input_blocks=True, middle_block=True,
                                       output_blocks=True, chunks=3)
    kw = dict(switch_on=True, flow=flow, split_ratio_fft=0.8, alpha=0.8)
    if mode.startswith("in_"):
        ohooks.register_spa_attn_injection(r, names, 1, input_blocks=True, middle_block=False, output_blocks=False,
                                           chunks=3, block_indices=list(range(9)), fusion=mode[3:], **kw)
    elif mode == "out_fft":
        ohooks.register_spa_attn_injection(r, names, 1, input_blocks=False, middle_block=False, output_blocks=True,
                                           chunks=3, block_indices=list(range(9)), fusion="fft", **kw)
    elif mode == "sel_replace_025":
        ohooks.register_spa_attn_injection(r, names, 1, input_blocks=True, middle_block=True, output_blocks=True,
                                           chunks=3, block_indices=[0, 2, 5], fusion="replace", **kw)
    elif mode == "chunks2":
        ohooks.register_spa_attn_injection(r, names, 1, input_blocks=True, middle_block=False, output_blocks=True,
                                           chunks=2, block_indices=[0, 1, 2], **kw)
    return r


MODES = ["off", "in_replace", "in_fft", "in_flow_fix", "in_fft_vfixed", "in_mix", "in_temporal", "in_adaIn", "out_fft",
         "sel_replace_025", "chunks2"]


@pytest.mark.parametrize("mode", MODES)
def test_small_unet_hook_modes_vs_oracle(small, mode):
    """model_channels=64 (head dims 8/16/32) at a 32x32 latent; the flow gate fires at level 0 (n = 1024)."""
    ldm, sampler, sd = small
    F_, h, w = 2, 32, 32
    n = 4 if mode == "chunks2" else 6
    x = synth.synth_normal("small.x", (6, 9, h, w))[:n]
    ctx = synth.synth_normal("small.ctx", (6, 1, 768))[:n]
    t = torch.full((n,), 481, dtype=torch.long)
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    _register(sampler, mode, flow)
    got = ldm.apply_model(x.to(DEV), t.to(DEV), ctx.to(DEV)).float().cpu()
    ref = ounet.unet_forward(sd, SMALL, x, t, ctx, _oracle_registry(mode, flow))
    err = rel_l2(got, ref)
    print(f"{mode}: rel-L2 {err:.3e}")
    assert err < SMALL_BOUND, (mode, err)


def test_cpu_tensors_fail_loudly(small):
    ldm, sampler, _ = small
    x = synth.synth_normal("small.x", (6, 9, 32, 32))
    ctx = synth.synth_normal("small.ctx", (6, 1, 768))
    t = torch.full((6,), 481, dtype=torch.long)
    _register(sampler, "off", None)
    from vface_amd import hip
    with pytest.raises(hip.VFaceHipError):
        ldm.apply_model(x, t, ctx)


def test_standalone_functions_vs_reference_golden():
    """combine_fft_high_low / align_by_flow / warp_image as stand-alone GPU functions against the reference's outputs."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from cases import make_flows
    from vface_amd.scripts.face_swap_utils import combine_fft_high_low
    from vface_amd.scripts.temporal_flow import align_by_flow, warp_image
    g = load_golden("fsai")
    for d in (320, 640, 1280):
        q1 = synth.synth_normal(f"fsai.q1.{d}", (2, 5, d), seed=1).to(DEV)
        q2 = synth.synth_normal(f"fsai.q2.{d}", (2, 5, d), seed=2).to(DEV)
        out = combine_fft_high_low(q1, q2, split_ratio=0.8).cpu()
        assert rel_l2(out, g[f"d{d}_r0.8"]) < 1e-3
    gw = load_golden("warp")
    fl = {k: torch.from_numpy(v) for k, v in make_flows(64, 64).items()}
    img = synth.synth_normal("warp.img", (3, 8, 64, 64), seed=3)
    out = align_by_flow(img.to(DEV), [fl["pm3"][None], fl["smooth"][None]], alpha=0.8).cpu()
    assert (out - gw["align_a0.8"]).abs().max() < 6e-3  # fp16 storage of O(1) values
    for case in ("subpixel", "oob"):
        w = warp_image(img[:1].to(DEV), fl[case][None].to(DEV)).cpu()
        assert (w[0] - gw[f"warp_{case}"]).abs().max() < 6e-3, case


@pytest.mark.parametrize("drop,vae", [(False, False), (True, False), (True, True)])
def test_cli_pipelined_inversion_is_bit_identical(tmp_path, drop, vae):
    """--pipeline_inversion (DDIMSampler.sample_while_inverting: batch k + 1's DDIM inversion beside batch k's sampling, on two
    streams, hooks switched per step) writes the frames of the sequential order, bit for bit -- three batches (inversion alone,
    two overlapped pairs, sampling alone), shipped hook schedule incl. the flow warp, with and without dead-branch elimination."""
    import yaml
    from vface_amd.scripts import VFace_inference_batch as cli
    cfg = {"model": {"params": {"unet_config": {"params": small_cfg()}}}}
    ypath = tmp_path / "small.yaml"
    ypath.write_text(yaml.safe_dump(cfg))
    outs = {}
    for mode in ("seq", "pipe"):
        args = ["--synthetic", "--config", str(ypath), "--n_frames", "12", "--n_samples", "4", "--H", "256", "--W", "256", "--max_steps", "3",
                "--fusion", "flow_fix", "--flow_gate", "flow_hw", "--Base_dir", str(tmp_path / mode), "--ddim_steps", "50"]
        args += ["--pipeline_inversion"] if mode == "pipe" else []
        args += ["--drop_dead_branches"] if drop else []
        args += ["--with_vae"] if vae else []        # (the VAE-in-the-loop path -- encode, DDIM, decode -- pipelined too)
        res = cli.main(args)
        assert len(res["batches"]) == 3 and all(b["finite"] for b in res["batches"])
        if mode == "pipe":
            assert [("sampling_beside_next_inversion" in b["stage_seconds"]) for b in res["batches"]] == [True, True, False]
        outs[mode] = [torch.load(tmp_path / mode / f"samples_batch{k}.pt") for k in range(3)]
        if vae:
            outs[mode] += [torch.load(tmp_path / mode / f"pixels_batch{k}.pt") for k in range(3)]
    for k in range(len(outs["seq"])):
        assert torch.equal(outs["seq"][k], outs["pipe"][k]), f"output {k}: pipelined != sequential ({_diff_pattern(outs['pipe'][k], outs['seq'][k])})"


def test_cli_synthetic_smoke(tmp_path):
    """scripts/VFace_inference_batch.py --synthetic with a small UNet config: inversion + sampling, 2 steps each."""
    import yaml
    from vface_amd.scripts import VFace_inference_batch as cli
    cfg = {"model": {"params": {"unet_config": {"params": small_cfg()}}}}
    ypath = tmp_path / "small.yaml"
    ypath.write_text(yaml.safe_dump(cfg))
    res = cli.main(["--synthetic", "--config", str(ypath), "--n_frames", "4", "--n_samples", "2", "--H", "256", "--W", "256",
                    "--max_steps", "2", "--Base_dir", str(tmp_path / "out"), "--ddim_steps", "50"])
    assert len(res["batches"]) == 2 and all(b["finite"] for b in res["batches"])
    assert (tmp_path / "out" / "samples_batch1.pt").exists()
    # the same run with the first-stage VAE in the loop: synthetic images -> encode -> DDIM -> decode -> pixels in [0, 1]
    res = cli.main(["--synthetic", "--with_vae", "--config", str(ypath), "--n_frames", "2", "--n_samples", "2", "--H", "256",
                    "--W", "256", "--max_steps", "2", "--Base_dir", str(tmp_path / "out_vae"), "--ddim_steps", "50"])
    assert res["batches"][0]["finite"] and res["batches"][0]["pixels"] == [2, 3, 256, 256]
    px = torch.load(tmp_path / "out_vae" / "pixels_batch0.pt")
    assert px.min() >= 0 and px.max() <= 1
    # ... with the flow computed from the target frames by the RAFT-shaped producer (:550-553 `return_flow`), pixel resolution,
    # resampled to the latent map and used by the flow_fix hooks
    res = cli.main(["--synthetic", "--raft_flow", "--fusion", "flow_fix", "--flow_gate", "flow_hw", "--config", str(ypath), "--n_frames", "2",
                    "--n_samples", "2", "--H", "256", "--W", "256", "--max_steps", "2", "--Base_dir", str(tmp_path / "out_raft"),
                    "--ddim_steps", "50"])
    assert res["batches"][0]["finite"]
    # ... and the paste-back of the decoded crops into (synthetic) 320 x 320 original frames (:603-636), on the GPU
    res = cli.main(["--synthetic", "--with_vae", "--paste_back", "--frame_size", "320", "--config", str(ypath), "--n_frames", "2",
                    "--n_samples", "2", "--H", "256", "--W", "256", "--max_steps", "2", "--Base_dir", str(tmp_path / "out_paste"),
                    "--ddim_steps", "50"])
    assert res["batches"][0]["pasted"] == [2, 320, 320, 3]
    from PIL import Image
    import numpy as np
    im = np.asarray(Image.open(tmp_path / "out_paste" / "pasted_b0_f1.png"))
    assert im.shape == (320, 320, 3) and im.dtype == np.uint8


@pytest.mark.parametrize("mode", ["in_flow_fix"])
def test_small_unet_bf16_measured(small, mode):
    """bf16 compute (the north-star's MFMA type): parity is ~8x looser than fp16 (bf16 has 3 fewer mantissa bits);
    the CPU emulation of bf16 rounding at the reference's autocast points is 1.6e-2 away from fp32 on this model."""
    from vface_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
    _, _, sd = small
    ldm = LatentDiffusion(dict(small_cfg(), compute_dtype=torch.bfloat16))
    ldm.unet.load_state_dict(sd)
    ldm = ldm.to(DEV)
    sampler = DDIMSampler(ldm)
    sampler.flow_gate = "flow_hw"
    F_, h, w = 2, 32, 32
    x = synth.synth_normal("small.x", (6, 9, h, w)); ctx = synth.synth_normal("small.ctx", (6, 1, 768))
    t = torch.full((6,), 481, dtype=torch.long)
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    _register(sampler, mode, flow)
    got = ldm.apply_model(x.to(DEV), t.to(DEV), ctx.to(DEV)).float().cpu()
    reg = _oracle_registry(mode, flow)
    ref = ounet.unet_forward(sd, SMALL, x, t, ctx, reg)
    err = rel_l2(got, ref)
    # the bound is what the emulation of this build's rounding points PREDICTS for bf16 (tests/precision_budget.py: bf16 weights,
    # one bf16 rounding per matrix-core operand, fp32 residual stream) on the same inputs, + 25 % -- not a round number
    # (the same emulation reproduces the fp16 measurement of the full UNet to 4 %: test_precision_budget.py)
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import precision_budget as pb
    with torch.no_grad():
        emu = pb.rel(pb.forward(pb.Emu(sd, w16=True, act16=True, stream16=False, half=torch.bfloat16), SMALL, x, t, ctx, reg), ref)
    print(f"bf16 {mode}: rel-L2 {err:.3e} (emulated rounding points: {emu:.3e})")
    assert err < 1.25 * emu and err < 2e-2, (err, emu)


@pytest.mark.parametrize("mode", ["plain", "flow_fix", "replace", "fft"])
def test_full_unet_vs_reference_golden(mode):
    """The real 859.5 M-parameter UNet, F=2 at 64x64, against the fixture the reference itself produced ("fft" = BASELINE
    config 3's own schedule, frequency-spectrum attention interpolation alone: round 4)."""
    from vface_amd.ldm.models.diffusion.ddpm import FFHQ_UNET_CONFIG, LatentDiffusion
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
    from vface_amd.ldm.models.pnp_utils import register_spa_attn_injection as reg
    g = load_golden("full_unet")
    ldm = _full_model()
    sampler = DDIMSampler(ldm)
    F_, h, w = 2, 64, 64
    x = synth.synth_normal("full.x", (3 * F_, 9, h, w)).to(DEV)
    ctx = synth.synth_normal("full.ctx", (3 * F_, 1, 768)).to(DEV)
    t = torch.full((3 * F_,), 481, dtype=torch.long, device=DEV)
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    reg(sampler, 1, switch_on=False, input_blocks=True, middle_block=True, output_blocks=True, chunks=3)
    if mode != "plain":
        reg(sampler, 1, switch_on=True, input_blocks=True, middle_block=False, output_blocks=False, chunks=3,
            flow=flow if mode == "flow_fix" else None, block_indices=list(range(9)), fusion=mode,
            split_ratio_fft=0.8, alpha=0.8)
    got = ldm.apply_model(x, t, ctx).float().cpu()
    err = rel_l2(got, g[mode])
    lp = load_golden("lowp")
    e_auto, e_w16 = rel_l2(lp[f"full.{mode}_autocast_f16"], g[mode]), rel_l2(lp[f"full.{mode}_w16"], g[mode])
    print(f"full UNet {mode}: rel-L2 vs reference {err:.3e}  (reference under fp16 autocast {e_auto:.3e}; reference with "
          f"fp16-rounded weights only {e_w16:.3e})")
    # north_star: "within 1e-3 relative fp16".  The reference's fp16 arithmetic is its autocast run: the distance between
    # this build and THAT output is printed and bounded too (two fp16 evaluations of one network differ by about the sum of
    # their rounding noises: measured 1.9e-3 = sqrt(1.18e-3^2 + 1.59e-3^2), i.e. the two errors are uncorrelated)
    e_vs_auto = rel_l2(got, lp[f"full.{mode}_autocast_f16"])
    print(f"full UNet {mode}: rel-L2 vs the reference's fp16-autocast output {e_vs_auto:.3e}")
    assert err < e_auto, (err, e_auto)        # closer to fp32 than the reference's own shipped arithmetic
    assert err < FULL_BOUND, err
    assert e_vs_auto < (err ** 2 + e_auto ** 2) ** 0.5 * 1.05, (e_vs_auto, err, e_auto)


SMALL_BOUND = 1.5e-3   # small UNet / DDIM loop / config 1: measured 1.36-1.39e-3 / 5.1e-4 / 9.4e-4 (the reference's own autocast: 2.06e-3)
FULL_BOUND = 1.35e-3   # measured 1.2e-3 with the fp32 residual stream (1.45e-3 without); the fp16-weight floor is 9.6e-4
_FULL = {}


def _full_model():
    if "m" not in _FULL:
        from vface_amd.ldm.models.diffusion.ddpm import FFHQ_UNET_CONFIG, LatentDiffusion
        ldm = LatentDiffusion(dict(FFHQ_UNET_CONFIG))
        synth.fill_module_(ldm.unet, seed=0)
        _FULL["m"] = ldm.to(DEV)
    return _FULL["m"]


def test_ddim_three_steps_and_inversion_vs_oracle(small):
    ldm, sampler, sd = small
    F_, h, w = 2, 32, 32
    x_T = synth.synth_normal("ddim.xT", (F_, 4, h, w))
    c, uc, tc = (synth.synth_normal(f"ddim.{k}", (F_, 1, 768)) for k in ("c", "uc", "tc"))
    inp = synth.synth_normal("ddim.inpaint", (F_, 4, h, w)) * 0.18215
    mask = synth.synth_mask(F_, h, w)
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    inv = {int(s): synth.synth_normal(f"ddim.inv.{int(s)}", (F_, 4, h, w)) for s in oddim.ddim_timesteps(50)}
    d = lambda v: v.to(DEV)
    img, inter = sampler.sample(S=50, batch_size=F_, shape=[4, h, w], conditioning=d(c), target_conditioning=d(tc),
                                inverse_results_dir={k: d(v) for k, v in inv.items()}, verbose=False,
                                unconditional_guidance_scale=3.0, unconditional_conditioning=d(uc), eta=0.0,
                                x_T=d(x_T), flow=flow, test_model_kwargs={"inpaint_image": d(inp), "inpaint_mask": d(mask)},
                                log_every_t=1, max_steps=3)
    names = ounet.attn1_names(SMALL)

    def apply_model(x, t, cc, reg):
        return ounet.unet_forward(sd, SMALL, x, t, cc, reg)

    ref, trace = oddim.sample(apply_model, names, 50, x_T, c, uc, tc, inv, inp, mask, scale=3.0, eta=0.0, flow=flow,
                              steps_limit=3)
    err = rel_l2(img.cpu(), ref)
    print(f"ddim 3 steps: rel-L2 {err:.3e}")
    assert err < SMALL_BOUND
    assert len(inter["x_inter"]) == 4
    # inversion, 2 steps, hooks off, stores the target half device-resident
    x0 = synth.synth_normal("ddim.z2", (2 * F_, 4, h, w))
    cond2 = torch.cat([tc, c], 0)
    store = {}
    xn, _ = sampler.ddim_invert(x=d(x0), cond=d(cond2), S=50, shape=[4, h, w], inverse_dir=store, batch_size=F_,
                                test_model_kwargs={"inpaint_image": d(torch.cat([inp] * 2)),
                                                   "inpaint_mask": d(torch.cat([mask] * 2))}, max_steps=2)
    rxn, rsaved = oddim.invert(lambda x, t, cc, reg: ounet.unet_forward(sd, SMALL, x, t, cc, None), 50, x0, cond2,
                               torch.cat([inp] * 2), torch.cat([mask] * 2), batch_size=F_, steps_limit=2)
    e_inv, e_saved = rel_l2(xn.cpu(), rxn), rel_l2(store[21].cpu(), rsaved[21])
    print(f"ddim inversion 2 steps: rel-L2 {e_inv:.3e} (saved latent {e_saved:.3e})")
    assert e_inv < SMALL_BOUND
    assert sorted(store) == [1, 21] and e_saved < SMALL_BOUND


def test_flow_resolution_mismatch_raises_like_reference(small):
    """a 512x512 flow against a 64x64 map is a RuntimeError in the reference's warp_image."""
    ldm, sampler, _ = small
    F_, h, w = 2, 32, 32
    z = torch.zeros(F_, 4, h, w, device=DEV)
    cc = torch.zeros(F_, 1, 768, device=DEV)
    with pytest.raises(RuntimeError):
        sampler.sample(S=50, batch_size=F_, shape=[4, h, w], conditioning=cc, target_conditioning=cc,
                       inverse_results_dir={}, verbose=False, unconditional_guidance_scale=3.0,
                       unconditional_conditioning=cc, x_T=z, flow=[torch.zeros(1, 2, 8 * h, 8 * w)],
                       test_model_kwargs={"inpaint_image": z, "inpaint_mask": z[:, :1]})


def test_bitwise_reproducible_and_batch_invariant(small):
    """No atomics anywhere on the path: the same inputs give the same bits, and a frame's result does not depend on
    which other frames share its batch (what makes frame sharding exact)."""
    ldm, sampler, _ = small
    h = w = 32
    xs = [synth.synth_normal(f"shard.x.{c}", (4, 9, h, w)) for c in range(3)]
    cs = [synth.synth_normal(f"shard.c.{c}", (4, 1, 768)) for c in range(3)]
    _register(sampler, "in_fft", None)

    def run(f0, fc):
        x = torch.cat([t[f0:f0 + fc] for t in xs]).to(DEV)
        ctx = torch.cat([t[f0:f0 + fc] for t in cs]).to(DEV)
        tt = torch.full((3 * fc,), 481, dtype=torch.long, device=DEV)
        return ldm.apply_model(x, tt, ctx).float().cpu()

    a, b = run(0, 4), run(0, 4)
    assert torch.equal(a, b)
    c = run(1, 2)
    assert torch.equal(c, torch.cat([a[k * 4 + 1:k * 4 + 3] for k in range(3)]))


def _full_run(fusion, frames, f0, fc, h, flow_all=None):
    """The 859.5 M-parameter UNet on frames [f0, f0+fc) of a synthetic `frames`-frame clip at h x h latents."""
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
    from vface_amd.ldm.models.pnp_utils import register_spa_attn_injection as reg
    ldm = _full_model()
    sampler = DDIMSampler(ldm)
    sampler.flow_gate = "flow_hw"
    pick = lambda name, shape: torch.stack([synth.synth_normal(f"cfg.{name}.{c}.{f}", shape)
                                            for c in range(3) for f in range(f0, f0 + fc)])
    x, ctx = pick("x", (9, h, h)).to(DEV), pick("c", (1, 768)).to(DEV)
    t = torch.full((3 * fc,), 481, dtype=torch.long, device=DEV)
    reg(sampler, 1, switch_on=False, input_blocks=True, middle_block=True, output_blocks=True, chunks=3)
    flow = None
    if fusion == "flow_fix":
        flow = [flow_all[i][None] for i in range(f0, f0 + fc - 1)]
    reg(sampler, 1, switch_on=True, input_blocks=True, middle_block=False, output_blocks=False, chunks=3, flow=flow,
        block_indices=list(range(9)), fusion=fusion, split_ratio_fft=0.8, alpha=0.8)
    out = ldm.apply_model(x, t, ctx).float()
    assert torch.isfinite(out).all()
    return out.reshape(3, fc, *out.shape[1:]).cpu()


@pytest.mark.parametrize("fusion,frames,h", [("fft", 32, 64), ("replace", 16, 64)])
def test_full_size_clip_is_batch_invariant(fusion, frames, h):
    """BASELINE configs 3 (32-frame clip, FSAI) at full size, and `replace` at 16 frames: FSAI / injection couple a frame
    only with its own chunk 0, so the clip computed in one batch must equal -- bit for bit -- the same frames computed
    8 at a time (the property frame sharding relies on; no oracle finishes at this size)."""
    whole = _full_run(fusion, frames, 0, frames, h)
    for f0 in (0, frames - 8):
        part = _full_run(fusion, frames, f0, 8, h)
        assert torch.equal(part, whole[:, f0:f0 + 8]), (fusion, f0)


def test_full_size_768_latent_flow_fix_dependency_window():
    """BASELINE config 5's geometry (768x768 -> 96x96 latents, n = 9216 / 2304 / 576 / 144) with all three modules; the
    flow gate is generalised to n == h*w of the flow field.  Smoothing is not recurrent inside a layer (frame f blends
    its own FSAI'd map with frame f-1's, temporal_flow.py:229-236), and two level-0 layers are hooked, so a frame depends
    on exactly two predecessors: in a run that starts at frame 1, frame 3 must come out bit for bit as in the whole
    clip, frame 2 must not (its second layer sees a frame 1 that lost its predecessor), and frame 0 equals an
    unsmoothed FSAI-only run."""
    h, frames = 96, 4
    flow_all = synth.synth_flow(frames - 1, h, h)
    whole = _full_run("flow_fix", frames, 0, frames, h, flow_all)
    tail = _full_run("flow_fix", frames, 1, 3, h, flow_all)
    assert torch.equal(tail[:, 2], whole[:, 3])
    assert not torch.equal(tail[:, 1], whole[:, 2])
    assert not torch.equal(tail[:, 0], whole[:, 1])
    single = _full_run("fft", frames, 0, 1, h)
    assert torch.equal(single[:, 0], whole[:, 0])        # frame 0 is never smoothed (temporal_flow.py:229)


def test_reference_flow_gate_is_the_default_and_pixel_flow_is_resampled(small):
    """(a) pnp_utils.py:201 warps only 4096-token maps: with the default gate a 32x32 clip's `flow_fix` equals `fft` bit for
    bit (the reference silently skips the warp there); the generalised gate (`flow_gate="flow_hw"`) does warp.
    (b) SURVEY F8 / 8f-3: a pixel-resolution field raises like the reference unless `flow_resample="area"`, and then equals a
    run that was handed the latent-resolution field the oracle's resample produces."""
    from oracle import flow as oflow
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
    ldm, _, _ = small
    h = w = 32
    F_ = 2
    x = synth.synth_normal("small.x", (6, 9, h, w)).to(DEV)
    ctx = synth.synth_normal("small.ctx", (6, 1, 768)).to(DEV)
    t = torch.full((6,), 481, dtype=torch.long, device=DEV)
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    outs = {}
    for gate in (None, "flow_hw"):
        sampler = DDIMSampler(ldm)
        if gate:
            sampler.flow_gate = gate
        for mode in ("in_fft", "in_flow_fix"):
            _register(sampler, mode, flow)
            outs[(gate, mode)] = ldm.apply_model(x, t, ctx).float().cpu()
    assert torch.equal(outs[(None, "in_flow_fix")], outs[(None, "in_fft")])
    assert torch.equal(outs[("flow_hw", "in_fft")], outs[(None, "in_fft")])
    assert not torch.equal(outs[("flow_hw", "in_flow_fix")], outs[("flow_hw", "in_fft")])
    # (b) through the sampler's loop, two steps
    z = synth.synth_normal("gate.z", (F_, 4, h, w)).to(DEV)
    cc = synth.synth_normal("gate.c", (F_, 1, 768)).to(DEV)
    inv = {int(s): z for s in oddim.ddim_timesteps(50)}
    fpx = synth.synth_flow(F_ - 1, 8 * h, 8 * w) * 8.0
    kw = dict(S=50, batch_size=F_, shape=[4, h, w], conditioning=cc, target_conditioning=cc, inverse_results_dir=inv,
              verbose=False, unconditional_guidance_scale=3.0, unconditional_conditioning=cc, x_T=z,
              test_model_kwargs={"inpaint_image": z, "inpaint_mask": z[:, :1]}, max_steps=2)
    s1 = DDIMSampler(ldm); s1.flow_gate = "flow_hw"
    with pytest.raises(RuntimeError):
        s1.sample(flow=[f[None] for f in fpx], **kw)
    s1.flow_resample = "area"
    a, _ = s1.sample(flow=[f[None] for f in fpx], **kw)
    s2 = DDIMSampler(ldm); s2.flow_gate = "flow_hw"
    b, _ = s2.sample(flow=[f[None] for f in oflow.flow_to_latent(fpx, 8)], **kw)
    assert (a - b).abs().max() < 2e-3 and torch.isfinite(a).all()


def test_full_size_config4_share_flow_fix_16_frames():
    """BASELINE configs[3] -- 64 frames, flow-guided smoothing, 4 GPUs -- at its per-GPU size: 16 frames at 64x64 latents
    with the shipped schedule (flow_fix on the input-block attn1).  No oracle finishes at this size; checked through the
    path's own invariants: a frame depends on exactly two predecessors (two hooked level-0 layers, each reading one
    neighbour), so frames 8..15 computed from a run that starts at frame 6 must equal the whole clip bit for bit -- which
    is also what a shard boundary relies on -- and the first frames of that run must not."""
    h, frames = 64, 16
    flow_all = synth.synth_flow(frames - 1, h, h)
    whole = _full_run("flow_fix", frames, 0, frames, h, flow_all)
    part = _full_run("flow_fix", frames, 6, 10, h, flow_all)
    assert torch.equal(part[:, 2:], whole[:, 8:])
    assert not torch.equal(part[:, 0], whole[:, 6]) and not torch.equal(part[:, 1], whole[:, 7])


def test_full_size_config5_share_32_frames_at_768():
    """BASELINE configs[4] -- 256 frames at 768x768, all three modules, 8 GPUs -- at its per-GPU size: 32 frames at 96x96
    latents (96 samples, n = 9216 tokens at level 0), flow_fix with the generalised gate.  Same invariants as above."""
    h, frames = 96, 32
    flow_all = synth.synth_flow(frames - 1, h, h)
    whole = _full_run("flow_fix", frames, 0, frames, h, flow_all)
    part = _full_run("flow_fix", frames, 22, 10, h, flow_all)
    assert torch.equal(part[:, 2:], whole[:, 24:])
    assert not torch.equal(part[:, 1], whole[:, 23])


def test_hipgraph_replay_equals_kernel_by_kernel_launches(small):
    """a DDIM step's UNet forward captured into a hipGraph (UNetEngine.step_forward_nhwc) and replayed gives
    the bits of the kernel-by-kernel launch sequence; one capture serves every step of every clip of the same shape and hook
    plan (new flow / conditioning tensors are copied into the graph's own buffers, the caller's are never written);
    another hook plan gets its own graph; inversion (hooks off, batch 2F) too."""
    from vface_amd.ldm.models.diffusion.ddim_w_inv import HookPlan
    ldm, sampler, sd = small
    eng = ldm.unet.engine
    F_, h, w = 2, 32, 32
    d = lambda v: v.to(DEV)
    x_T = d(synth.synth_normal("graph.xT", (F_, 4, h, w)))
    c, uc, tc = (d(synth.synth_normal(f"graph.{k}", (F_, 1, 768))) for k in ("c", "uc", "tc"))
    inp = d(synth.synth_normal("graph.inpaint", (F_, 4, h, w)) * 0.18215)
    mask = d(synth.synth_mask(F_, h, w))
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    inv = {int(s): d(synth.synth_normal(f"graph.inv.{int(s)}", (F_, 4, h, w))) for s in oddim.ddim_timesteps(50)}

    def run(plan, steps=4):
        nonlocal c
        sampler.hook_plan = plan
        img, _ = sampler.sample(S=50, batch_size=F_, shape=[4, h, w], conditioning=c, target_conditioning=tc,
                                inverse_results_dir=inv, verbose=False, unconditional_guidance_scale=3.0,
                                unconditional_conditioning=uc, eta=0.0, x_T=x_T, flow=flow,
                                test_model_kwargs={"inpaint_image": inp, "inpaint_mask": mask}, max_steps=steps)
        return img.clone()

    def invert():
        x0 = d(synth.synth_normal("graph.z2", (2 * F_, 4, h, w)))
        xn, _ = sampler.ddim_invert(x=x0, cond=torch.cat([tc, c], 0), S=50, shape=[4, h, w], inverse_dir={}, batch_size=F_,
                                    test_model_kwargs={"inpaint_image": torch.cat([inp] * 2), "inpaint_mask": torch.cat([mask] * 2)},
                                    max_steps=2)
        return xn.clone()

    old_plan, old_flag = sampler.hook_plan, eng.use_graph
    try:
        eng.use_graph = False
        plans = [HookPlan(fusion="flow_fix"), HookPlan(fusion="replace")]
        eager = [run(p) for p in plans] + [invert()]
        eng.use_graph, eng._graphs = True, {}
        g1 = run(plans[0])
        assert len(eng._graphs) == 1, "one capture must serve every step of the clip"
        g2 = run(plans[1])
        assert len(eng._graphs) == 2
        g1b = run(plans[0])                       # back to the first plan: its graph is still cached and still right
        assert len(eng._graphs) == 2
        gi = invert()
        assert len(eng._graphs) == 3
        # another clip (other flow fields, other conditioning) through the first plan's graph
        flow_b = [f * -0.5 for f in flow]
        c_b = d(synth.synth_normal("graph.c_b", (F_, 1, 768)))
        flow_keep = [f.clone() for f in flow]
        flow[:], c_a, c = flow_b, c, c_b
        g3 = run(plans[0])
        assert len(eng._graphs) == 3, "a new clip of the same shape must not be captured again"
        eng.use_graph = False
        e3 = run(plans[0])
        eng.use_graph = True
        flow[:], c = flow_keep, c_a
        g1c = run(plans[0])
        assert eng.use_graph and not eng._graph_failed, "capture fell back to the eager path"
        assert all(g["bytes"] > 0 for g in eng._graphs.values()) and sum(g["bytes"] for g in eng._graphs.values()) <= eng.graph_budget_bytes
        assert torch.equal(g1, eager[0]) and torch.equal(g1b, eager[0]) and torch.equal(g2, eager[1]) and torch.equal(gi, eager[2])
        assert torch.equal(g3, e3) and not torch.equal(g3, eager[0]) and torch.equal(g1c, eager[0])
    finally:
        sampler.hook_plan, eng.use_graph, eng._graphs = old_plan, old_flag, {}
        sampler.make_schedule(50, ddim_eta=0.0, verbose=False)


@pytest.mark.parametrize("fusion", ["replace", "fft", "mix", "none", "flow_fix"])
def test_two_launch_streams_equal_one(small, fusion):
    """The graph-replayed forward as two frame halves on two HIP streams (UNetEngine._step_forward_split) gives the bits of the
    single launch sequence -- sampling (batch 3F), with the dead branches dropped (2F), inversion (2F, unhooked).  Hook modes
    that never read another frame split freely; flow_fix, whose warp reads the previous frame, runs its halves as two in-process
    frame shards (parallel.StreamShard: half 0 hands its last frame's fused q|k to half 1 at every hooked flow layer); the
    modes that couple frames more widely stay whole."""
    from vface_amd.ldm.models.diffusion.ddim_w_inv import HookPlan
    ldm, sampler, sd = small
    eng = ldm.unet.engine
    F_, h, w = 6, 32, 32
    d = lambda v: v.to(DEV)
    x_T = d(synth.synth_normal("two.xT", (F_, 4, h, w)))
    c, uc, tc = (d(synth.synth_normal(f"two.{k}", (F_, 1, 768))) for k in ("c", "uc", "tc"))
    inp = d(synth.synth_normal("two.inpaint", (F_, 4, h, w)) * 0.18215)
    mask = d(synth.synth_mask(F_, h, w))
    inv = {int(s): d(synth.synth_normal(f"two.inv.{int(s)}", (F_, 4, h, w))) for s in oddim.ddim_timesteps(50)}
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)] if fusion == "flow_fix" else None
    old_gate = getattr(sampler, "flow_gate", None)
    sampler.flow_gate = "flow_hw"        # (32 x 32 latents: the reference's 4096-token gate would never fire)

    def run(drop):
        sampler.drop_dead_branches = drop
        img, _ = sampler.sample(S=50, batch_size=F_, shape=[4, h, w], conditioning=c, target_conditioning=tc,
                                inverse_results_dir=inv, verbose=False, unconditional_guidance_scale=3.0,
                                unconditional_conditioning=uc, eta=0.0, x_T=x_T, flow=flow,
                                test_model_kwargs={"inpaint_image": inp, "inpaint_mask": mask}, max_steps=3)
        return img.clone()

    def invert():
        x0 = d(synth.synth_normal("two.z2", (2 * F_, 4, h, w)))
        xn, _ = sampler.ddim_invert(x=x0, cond=torch.cat([tc, c], 0), S=50, shape=[4, h, w], inverse_dir={}, batch_size=F_,
                                    test_model_kwargs={"inpaint_image": torch.cat([inp] * 2), "inpaint_mask": torch.cat([mask] * 2)},
                                    max_steps=2)
        return xn.clone()

    old = sampler.hook_plan, eng.use_graph, eng._graphs, eng.split_streams, sampler.drop_dead_branches
    try:
        sampler.hook_plan = HookPlan(fusion=fusion, enabled=fusion != "none")
        res = {}
        for streams in (1, 2):
            eng.use_graph, eng._graphs, eng._split_state, eng.split_streams = True, {}, {}, streams
            eng._split_off.clear(); eng.split_timing.clear()       # (a configuration an earlier case timed faster as ONE sequence would not split at all)
            res[streams] = (run(False), run(True), invert())
            assert not eng._graph_failed
            split = [k for k in eng._split_state if k[0] != "plan"]
            assert (len(split) > 0) == (streams == 2), "the halves must actually have run on their own streams"
            if streams == 2 and fusion == "flow_fix":
                assert any("shard_objs" in v for k, v in eng._split_state.items() if k[0] != "plan"), "the coupled halves ran as stream shards"
                # ... and the engine timed the configuration both ways on its second step (it keeps the faster form: either is exact)
                t2, t1 = eng.split_timing[("coupled", 3 * F_, h, w)]
                assert t2 > 0 and t1 > 0 and ((("coupled", 3 * F_, h, w) in eng._split_off) == (t1 < 0.995 * t2))
        for what, a_, b_ in zip(("sampling 3F", "sampling 2F (dead branches dropped)", "inversion"), res[1], res[2]):
            assert torch.equal(a_, b_), f"{fusion}, {what}: two streams != one -- {_diff_pattern(b_, a_)}"
        if fusion == "replace":
            # halves that do NOT overlap (here: both on one stream, as two streams sharing a hardware queue would behave) are
            # detected on the second split step and the engine returns to one launch sequence -- with the right bits throughout
            one = torch.cuda.Stream()
            eng.use_graph, eng._graphs, eng._split_state, eng.split_streams = True, {}, {}, 2
            eng._split_off.clear(); eng.split_timing.clear()
            eng._split_pair, eng._split_verified, eng.split_overlap = [one, one], False, None
            with pytest.warns(UserWarning, match="do not overlap"):
                r = run(False)
            assert eng.split_streams == 1 and eng.split_overlap > 0.8 and torch.equal(r, res[1][0])
            eng._split_pair, eng._split_verified = None, False
        # flow_fix splits as two coupled shards (VFACE_SPLIT_COUPLED=0 keeps it whole); wider couplings keep the batch whole
        sampler.hook_plan = HookPlan(fusion="flow_fix")
        sampler._register_step_hooks([synth.synth_flow(F_ - 1, h, w)[i][None].to(DEV) for i in range(F_ - 1)])
        eng.split_streams = 2
        assert eng._split_plan(3 * F_) is not None and eng._split_coupled
        os.environ["VFACE_SPLIT_COUPLED"] = "0"
        try:
            assert eng._split_plan(3 * F_) is None
        finally:
            del os.environ["VFACE_SPLIT_COUPLED"]
        sampler.hook_plan = HookPlan(fusion="temporal")
        sampler._register_step_hooks(None)
        assert eng._split_plan(3 * F_) is None
        eng.split_streams = 2
        sampler.hook_plan = HookPlan(fusion="replace")
        sampler._register_step_hooks(None)
        assert eng._split_plan(3 * F_) is not None and eng._split_plan(3 * 5) is None and eng._split_plan(6) is None
    finally:
        sampler.hook_plan, eng.use_graph, eng._graphs, eng.split_streams, sampler.drop_dead_branches = old
        eng._split_state = {}
        sampler.flow_gate = old_gate
        sampler.make_schedule(50, ddim_eta=0.0, verbose=False)


def test_config1_single_frame_256x256_twenty_steps_vs_oracle(small):
    """BASELINE configs[0] (the reference's own CPU-runnable case): ONE 256 x 256 frame (32 x 32 latent), the whole 20-step DDIM
    loop with the shipped hook schedule -- a clip of one frame has no flow field and no neighbour, every hook degenerates to
    its chunk-0 form -- against the CPU oracle running the same 20 steps."""
    ldm, sampler, sd = small
    F_, h, w, S = 1, 32, 32, 20
    x_T = synth.synth_normal("cfg1.xT", (F_, 4, h, w))
    c, uc, tc = (synth.synth_normal(f"cfg1.{k}", (F_, 1, 768)) for k in ("c", "uc", "tc"))
    inp = synth.synth_normal("cfg1.inpaint", (F_, 4, h, w)) * 0.18215
    mask = synth.synth_mask(F_, h, w)
    inv = {int(s): synth.synth_normal(f"cfg1.inv.{int(s)}", (F_, 4, h, w)) for s in oddim.ddim_timesteps(S)}
    d = lambda v: v.to(DEV)
    try:
        img, inter = sampler.sample(S=S, batch_size=F_, shape=[4, h, w], conditioning=d(c), target_conditioning=d(tc),
                                    inverse_results_dir={k: d(v) for k, v in inv.items()}, verbose=False,
                                    unconditional_guidance_scale=3.0, unconditional_conditioning=d(uc), eta=0.0,
                                    x_T=d(x_T), flow=[], test_model_kwargs={"inpaint_image": d(inp), "inpaint_mask": d(mask)},
                                    log_every_t=1000)
    finally:
        sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
    names = ounet.attn1_names(SMALL)
    ref, _ = oddim.sample(lambda x, t, cc, reg: ounet.unet_forward(sd, SMALL, x, t, cc, reg), names, S, x_T, c, uc, tc, inv, inp, mask,
                          scale=3.0, eta=0.0, flow=[])
    err = rel_l2(img.cpu(), ref)
    print(f"config 1, 20 DDIM steps: rel-L2 {err:.3e}")
    assert torch.isfinite(img).all() and err < SMALL_BOUND


def test_config1_full_unet_single_frame_256x256_twenty_steps_vs_oracle():
    """BASELINE configs[0] on the REAL model: the 859.5 M-parameter UNet (project_ffhq.yaml:33-56), ONE 256 x 256 frame
    (32 x 32 latent, F = 1: batch [uncond ; cond ; recon] = 3), the whole 20-step DDIM loop with the shipped hook schedule,
    against the CPU oracle running the same 20 steps (60 sample-forwards of ~0.8 s).  At this resolution the UNet's bottom
    level is 4 x 4 pixels: 3x3 convolutions at 1280 / 2560 channels on 16-pixel images and attention over n = 16 tokens at
    head dim 160 (middle block), n = 64 / 256 / 1024 above it -- shape classes no other test reaches
    (openaimodel.py:860-907; ddim_w_inv.py:254-355 with S = 20, SURVEY F12: inversion steps = sampling steps)."""
    import os
    import time
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
    ldm = _full_model()
    sampler = DDIMSampler(ldm)
    spec = ounet.UNetSpec()
    sd = {k: v.float().cpu() for k, v in ldm.unet.state_dict().items()}
    F_, h, w, S = 1, 32, 32, 20
    x_T = synth.synth_normal("cfg1f.xT", (F_, 4, h, w))
    c, uc, tc = (synth.synth_normal(f"cfg1f.{k}", (F_, 1, 768)) for k in ("c", "uc", "tc"))
    inp = synth.synth_normal("cfg1f.inpaint", (F_, 4, h, w)) * 0.18215
    mask = synth.synth_mask(F_, h, w)
    inv = {int(s): synth.synth_normal(f"cfg1f.inv.{int(s)}", (F_, 4, h, w)) for s in oddim.ddim_timesteps(S)}
    d = lambda v: v.to(DEV)
    try:
        img, _ = sampler.sample(S=S, batch_size=F_, shape=[4, h, w], conditioning=d(c), target_conditioning=d(tc),
                                inverse_results_dir={k: d(v) for k, v in inv.items()}, verbose=False,
                                unconditional_guidance_scale=3.0, unconditional_conditioning=d(uc), eta=0.0,
                                x_T=d(x_T), flow=[], test_model_kwargs={"inpaint_image": d(inp), "inpaint_mask": d(mask)},
                                log_every_t=1000)
    finally:
        sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
    names = ounet.attn1_names(spec)
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))   # the GPU box's share is 16 cores of a 256-thread host
    t0 = time.time()
    try:
        with torch.no_grad():
            ref, _ = oddim.sample(lambda x, t, cc, reg: ounet.unet_forward(sd, spec, x, t, cc, reg), names, S, x_T, c, uc, tc, inv,
                                  inp, mask, scale=3.0, eta=0.0, flow=[])
    finally:
        torch.set_num_threads(nthr)
    err = rel_l2(img.cpu(), ref)
    print(f"config 1 on the 859.5M UNet, 20 DDIM steps: rel-L2 {err:.3e}  (oracle {time.time() - t0:.0f} s)")
    assert torch.isfinite(img).all() and err < SMALL_BOUND


DROP_MODES = ["off", "replace", "fft", "flow_fix", "fft_vfixed", "mix", "temporal", "adaIn"]


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("fusion", DROP_MODES)
def test_drop_recon_is_bit_identical(small, fusion, graph):
    """exact dead-branch elimination.  The recon third of every sampling batch is a pure sink
    (REFace ddim_w_inv.py:667,703-707: e_t_recon only feeds x_prev_recon, and :738 returns x_prev, pred_x0 without it;
    pnp_utils.py:136-142,195-199,255-256: every hook mode writes INTO chunk 2, never reads it), so
    ``sampler.drop_dead_branches`` runs the UNet on [uncond ; cond] only: x_prev and pred_x0 of a 3-step loop must be
    ``torch.equal`` to the full-batch run for EVERY fusion mode, kernel by kernel and through hipGraph replay."""
    from vface_amd.ldm.models.diffusion.ddim_w_inv import HookPlan
    ldm, sampler, sd = small
    eng = ldm.unet.engine
    F_, h, w = 2, 32, 32
    d = lambda v: v.to(DEV)
    x_T = d(synth.synth_normal("drop.xT", (F_, 4, h, w)))
    c, uc, tc = (d(synth.synth_normal(f"drop.{k}", (F_, 1, 768))) for k in ("c", "uc", "tc"))
    inp = d(synth.synth_normal("drop.inpaint", (F_, 4, h, w)) * 0.18215)
    mask = d(synth.synth_mask(F_, h, w))
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    inv = {int(s_): d(synth.synth_normal(f"drop.inv.{int(s_)}", (F_, 4, h, w))) for s_ in oddim.ddim_timesteps(50)}

    def run(drop):
        sampler.drop_dead_branches = drop
        sampler.hook_plan = HookPlan(fusion=fusion, enabled=fusion != "off")
        torch.manual_seed(7)
        img, inter = sampler.sample(S=50, batch_size=F_, shape=[4, h, w], conditioning=c, target_conditioning=tc,
                                    inverse_results_dir=inv, verbose=False, unconditional_guidance_scale=3.0,
                                    unconditional_conditioning=uc, eta=0.0, x_T=x_T, flow=flow, log_every_t=1,
                                    test_model_kwargs={"inpaint_image": inp, "inpaint_mask": mask}, max_steps=3)
        return img.clone(), [t.clone() for t in inter["pred_x0"]]

    old = sampler.hook_plan, eng.use_graph, eng._graphs, sampler.drop_dead_branches
    try:
        eng.use_graph, eng._graphs = graph, {}
        full, full_p0 = run(False)
        cut, cut_p0 = run(True)
        assert torch.equal(full, cut), (fusion, (full - cut).abs().max().item())
        assert len(full_p0) == len(cut_p0) and all(torch.equal(a, b) for a, b in zip(full_p0, cut_p0))
        assert eng.live_chunks is None
        if graph:
            assert not eng._graph_failed and len(eng._graphs) == 2      # one graph per batch shape
    finally:
        sampler.hook_plan, eng.use_graph, eng._graphs, sampler.drop_dead_branches = old


def test_drop_source_half_of_inversion_is_bit_identical(small):
    """The inversion batch is [target ; source] but only ``nosie[:batch_size]`` is ever saved (REFace ddim_w_inv.py:464-486) and
    the entry point re-loads x_noisy from those files (VFace_inference_batch.py:531-543): with ``drop_dead_branches`` the loop
    runs on the target half alone and must save the same bits."""
    ldm, sampler, sd = small
    F_, h, w = 2, 32, 32
    d = lambda v: v.to(DEV)
    c, tc = (d(synth.synth_normal(f"drop.{k}", (F_, 1, 768))) for k in ("c", "tc"))
    inp = d(synth.synth_normal("drop.inpaint", (F_, 4, h, w)) * 0.18215)
    mask = d(synth.synth_mask(F_, h, w))
    x0 = d(synth.synth_normal("drop.z2", (2 * F_, 4, h, w)))
    stores = []
    old = sampler.drop_dead_branches
    try:
        for drop in (False, True):
            sampler.drop_dead_branches = drop
            st = {}
            xn, _ = sampler.ddim_invert(x=x0, cond=torch.cat([tc, c], 0), S=50, shape=[4, h, w], inverse_dir=st, batch_size=F_,
                                        test_model_kwargs={"inpaint_image": torch.cat([inp] * 2), "inpaint_mask": torch.cat([mask] * 2)},
                                        max_steps=3)
            stores.append((st, xn.clone()))
    finally:
        sampler.drop_dead_branches = old
    (a, xa), (b, xb) = stores
    assert sorted(a) == sorted(b) == [1, 21, 41]
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert xb.shape[0] == F_ and torch.equal(xa[:F_], xb)


@pytest.mark.parametrize("mode", ["off", "in_replace", "in_fft", "in_flow_fix", "in_fft_vfixed"])
def test_decomposed_attn1_equals_one_call_form_bit_for_bit(small, mode):
    """bench.py's instrumented pass issues the launches of ``vface_attn1_forward`` call by call (UNetEngine._attn1_decomposed)
    to time the projections and the attention kernel separately: same kernels, same parameters, same order -> same bits."""
    ldm, sampler, _ = small
    F_, h, w = 2, 32, 32
    x = synth.synth_normal("small.x", (6, 9, h, w)).to(DEV)
    ctx = synth.synth_normal("small.ctx", (6, 1, 768)).to(DEV)
    t = torch.full((6,), 481, dtype=torch.long, device=DEV)
    flow = [synth.synth_flow(F_ - 1, h, w)[i][None] for i in range(F_ - 1)]
    _register(sampler, mode, flow)
    eng = ldm.unet.engine
    try:
        eng.decompose_attn1 = False
        a = ldm.apply_model(x, t, ctx).clone()
        eng.decompose_attn1 = True
        b = ldm.apply_model(x, t, ctx).clone()
    finally:
        eng.decompose_attn1 = False
    assert torch.equal(a, b), (a - b).abs().max().item()


def _shared_prefix_pair(ldm, sampler, mode, F_, h, graph, tag):
    """eps of ONE forward of the sampler's own batch -- x_in = [x ; x ; inv_t], t_in = [t] * 3 (ddim_w_inv.py:632-655) -- with the
    chunk-0 / chunk-1 prefix computed once (UNetEngine._shared_block) and with every chunk computed on its own.  Returns
    ``(unshared, shared, (x9, t, ctx, flow))``: eps as [3, F, 4, h, h] on the CPU and the NCHW inputs of the same batch for an oracle."""
    from vface_amd import hip
    from vface_amd.engine import Act
    eng = ldm.unet.engine
    d = lambda v: v.to(DEV)
    x = synth.synth_normal(f"{tag}.x", (F_, 4, h, h))
    inv = synth.synth_normal(f"{tag}.inv", (F_, 4, h, h))
    inp = synth.synth_normal(f"{tag}.inp", (F_, 4, h, h)) * 0.18215
    mask = synth.synth_mask(F_, h, h)
    ctx = synth.synth_normal(f"{tag}.ctx", (3 * F_, 1, 768))
    flow = [synth.synth_flow(F_ - 1, h, h)[i][None] for i in range(F_ - 1)]
    x9 = torch.cat([torch.cat([x, x, inv]), torch.cat([inp] * 3), torch.cat([mask] * 3)], dim=1)
    _register(sampler, mode, [d(f) for f in flow])
    x_in = torch.empty(3 * F_ * h * h, 16, dtype=eng.dtype, device=DEV)
    hip.pack_unet_input(d(x).contiguous(), d(inv).contiguous(), d(inp).contiguous(), d(mask).float().contiguous(), x_in,
                        F=F_, h=h, w=h, cpad=16)
    t = torch.full((3 * F_,), 481, dtype=torch.long)
    calls = []
    orig = eng._shared_block
    eng._shared_block = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    old = eng.use_graph, eng._graphs, eng.split_streams, eng.share_prefix
    out = {}
    try:
        eng.use_graph, eng._graphs, eng.split_streams = graph, {}, 1
        for share in (False, True):
            eng.share_prefix = share
            for _ in range(2 if graph else 1):      # (graph: the second call replays)
                e = eng.step_forward_nhwc(Act(x_in, 3 * F_, h, h), d(t), d(ctx))
            out[share] = e.float().reshape(3, F_, h, h, -1).permute(0, 1, 4, 2, 3).cpu().clone()
            assert bool(calls) == (share and mode not in ("in_fft_vfixed", "in_temporal", "in_adaIn")), \
                "the shared block runs exactly when the sampler states its batch (and the hook mode allows it)"
            calls.clear()
    finally:
        eng.use_graph, eng._graphs, eng.split_streams, eng.share_prefix = old
        del eng._shared_block
    return out[False], out[True], (x9, t, ctx, flow)


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("mode", ["off", "in_replace", "in_fft", "in_flow_fix", "in_mix", "out_fft", "in_fft_vfixed", "in_temporal"])
def test_shared_uncond_cond_prefix_small(small, mode, graph):
    """VERDICT r5 next #1.  Chunks 0 and 1 of the sampler's batch are the same x at the same t up to the first attn2: with the
    prefix run once, chunks 0 and 2 keep their bits; chunk 1 keeps them wherever the hook's edit of identical inputs is computed by
    the same launches (no hook, replace, flow_fix: its fused projection reads A's rows twice).  Under fft / mix q1 = FSAI(q0, q0) is
    taken as q0 -- what the reference's FFT round trip returns up to fp32 rounding -- instead of the folded-weight projection of
    the same rows: chunk 1 then moves by the folded weights' own rounding (measured 1.1-1.2e-3 of its eps here) and must be
    CLOSER to the oracle than before.  Modes the shared block does not take (fft_vfixed, temporal) run whole: equal throughout."""
    ldm, sampler, sd = small
    a, b, (x9, t, ctx, flow) = _shared_prefix_pair(ldm, sampler, mode, 2, 32, graph, "share")
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]), f"{mode}: chunk 0 / chunk 2 -- {_diff_pattern(b[0], a[0])} / {_diff_pattern(b[2], a[2])}"
    if mode in ("in_fft", "in_mix"):
        ref = ounet.unet_forward(sd, SMALL, x9, t, ctx, _oracle_registry(mode, flow)).reshape(3, 2, 4, 32, 32)
        e_un, e_sh, e1 = rel_l2(a[1], ref[1]), rel_l2(b[1], ref[1]), rel_l2(b[1], a[1])
        print(f"{mode} graph={graph}: chunk 1 vs oracle: unshared {e_un:.3e}, shared {e_sh:.3e}; shared vs unshared {e1:.3e}")
        assert e_sh <= e_un * 1.02 and e_sh < SMALL_BOUND and e1 < 2e-3, (mode, e_un, e_sh, e1)
    else:
        assert torch.equal(a[1], b[1]), f"{mode}: chunk 1 -- {_diff_pattern(b[1], a[1])}"


@pytest.mark.parametrize("mode", ["in_fft", "in_flow_fix", "in_replace"])
def test_shared_uncond_cond_prefix_full_unet(mode):
    """The same on the 859.5 M UNet at 64 x 64 latents (n = 4096, C = 320: the production front / attention / tail launches), F = 2;
    under fft against the REFERENCE's own output on this very batch (fixture full_unet_sampler_batch.npz: make_golden.py runs the
    reference UNet on ``cat([x, x, inv_t])`` exactly as its sampler assembles it): every chunk within the whole-network bound, chunk 1
    no further from the reference than the unshared launches put it."""
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
    ldm = _full_model()
    sampler = DDIMSampler(ldm)
    a, b, _ = _shared_prefix_pair(ldm, sampler, mode, 2, 64, True, "sb")
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]), f"{mode}: chunk 0 / chunk 2 -- {_diff_pattern(b[0], a[0])} / {_diff_pattern(b[2], a[2])}"
    if mode == "in_fft":
        ref = load_golden("full_unet_sampler_batch")["fft"].reshape(3, 2, 4, 64, 64)
        e_all_un, e_all_sh = rel_l2(a, ref), rel_l2(b, ref)
        e_un, e_sh, e1 = rel_l2(a[1], ref[1]), rel_l2(b[1], ref[1]), rel_l2(b[1], a[1])
        print(f"full UNet fft, the sampler's batch vs the reference: unshared {e_all_un:.3e}, shared {e_all_sh:.3e}; chunk 1: unshared "
              f"{e_un:.3e}, shared {e_sh:.3e}; shared vs unshared {e1:.3e}")
        assert e_all_sh < FULL_BOUND and e_all_un < FULL_BOUND, (e_all_sh, e_all_un)
        assert e_sh <= e_un * 1.02 and e1 < 2e-3, (e_un, e_sh, e1)
    else:
        assert torch.equal(a[1], b[1]), _diff_pattern(b[1], a[1])
