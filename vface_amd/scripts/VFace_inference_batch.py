"""Entry point of the VFace path with the reference's flags (``REFace/scripts/VFace_inference_batch.py:673-860``).

What runs here is the part of ``run_inference`` that is the hot path (``:529-594``): optional DDIM inversion of
the target latents into a latent cache, then ``sampler.sample`` over batches of ``--n_samples`` frames with the
shipped hook schedule.  Everything around it in the reference -- video decoding, dlib crop/align, face parsing,
CLIP/ArcFace/landmark conditioning, the KL-VAE and paste-back (``:193-528, 596-670``) -- is out of scope of this
build: it needs checkpoints and third-party models that are not available offline.  ``--synthetic``
therefore stands in for those stages with seeded synthetic latents / conditioning / flow of the right shapes and
writes the denoised latents; without it the script stops with a message saying what is missing.

    python -m vface_amd.scripts.VFace_inference_batch --synthetic --n_frames 24 --n_samples 8 --ddim_steps 50
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    # flags of the reference (same names, defaults and meaning)
    p.add_argument("--prompt", type=str, nargs="?", default="a photograph of an astronaut riding a horse")
    p.add_argument("--data_config", type=str, nargs="?", default=None, help="yaml of (video dir, source image) pairs")
    p.add_argument("--Base_dir", type=str, nargs="?", default="results_video_new", help="dir to write results to")
    p.add_argument("--skip_grid", action="store_true")
    p.add_argument("--skip_save", action="store_true")
    p.add_argument("--ddim_steps", type=int, default=50, help="number of ddim sampling steps")
    p.add_argument("--plms", action="store_true")
    p.add_argument("--laion400m", action="store_true")
    p.add_argument("--fixed_code", action="store_true")
    p.add_argument("--Start_from_target", action="store_true", default=True)
    p.add_argument("--only_target_crop", action="store_true", default=True)
    p.add_argument("--target_start_noise_t", type=int, default=1000)
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=2)
    p.add_argument("--H", type=int, default=512)
    p.add_argument("--W", type=int, default=512)
    p.add_argument("--C", type=int, default=4, help="latent channels")
    p.add_argument("--f", type=int, default=8, help="downsampling factor")
    p.add_argument("--n_samples", type=int, default=6, help="frames per sampler batch")
    p.add_argument("--n_frames", type=int, default=24)
    p.add_argument("--n_rows", type=int, default=0)
    p.add_argument("--scale", type=float, default=3.0, help="unconditional guidance scale")
    p.add_argument("--src_image_mask", type=str, default=None)
    p.add_argument("--from-file", type=str, default=None)
    p.add_argument("--config", type=str, default=None, help="project_ffhq.yaml (UNet hyper-parameters)")
    p.add_argument("--ckpt", type=str, default=None, help="last.ckpt; its model.diffusion_model.* keys are loaded")
    p.add_argument("--ckpt_stub_unknown_globals", action="store_true",
                   help="read a pytorch_lightning last.ckpt whose non-tensor entries (callbacks, hyper_parameters) the "
                        "weights-only loader rejects: tensors through torch's own allow-list, every other pickled global as an "
                        "inert placeholder (nothing outside the allow-list is ever called)")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--rank", type=int, default=0)
    p.add_argument("--precision", type=str, choices=["full", "autocast"], default="autocast")
    # additions of this build
    p.add_argument("--synthetic", action="store_true", help="synthetic latents/conditioning/flow instead of video I/O")
    p.add_argument("--raft_flow", action="store_true",
                   help="compute the optical flow from the (synthetic) target frames with the RAFT-shaped producer "
                        "(temporal_flow.return_flow, VFace_inference_batch.py:550-553) instead of a synthetic field; implies "
                        "--flow_pixels.  --raft_ckpt: a raft_large state dict (torchvision layout); default synthetic weights")
    p.add_argument("--raft_ckpt", type=str, default=None)
    p.add_argument("--paste_back", action="store_true",
                   help="with --with_vae: paste the decoded crops into (synthetic) original frames on the GPU, the block of "
                        "VFace_inference_batch.py:603-636 (scripts/paste_back.py); --only_target_crop is implied")
    p.add_argument("--frame_size", type=int, default=1024, help="side of the synthetic original frames of --paste_back")
    p.add_argument("--with_vae", action="store_true",
                   help="synthetic run through the first-stage KL-VAE too: the inpaint latents come from encode_first_stage of "
                        "synthetic images (:456-457) and the samples are decoded to pixels (:596-600)")
    p.add_argument("--fusion", type=str, default="flow_fix", help="hook mode on the input-block attn1 modules")
    p.add_argument("--no_inversion", action="store_true", help="use random recon latents instead of DDIM inversion")
    p.add_argument("--flow_pixels", action="store_true",
                   help="hand the sampler the flow at PIXEL resolution, as the reference's return_flow produces it "
                        "(temporal_flow.py:163-188), and let it be resampled to the latent map (area mean / 8, vface_flow_to_latent) "
                        "instead of failing like the reference does (SURVEY F8)")
    p.add_argument("--flow_gate", choices=["reference", "flow_hw"], default="reference",
                   help="which attention maps the flow smoothing touches: 'reference' = exactly pnp_utils.py:201 (the 4096-token "
                        "maps of a 512x512 clip, nothing at any other resolution); 'flow_hw' = the level whose token count equals "
                        "h*w of the flow field (needed for the module to act at e.g. 768x768)")
    p.add_argument("--compute_dtype", choices=["fp16", "bf16"], default="fp16")
    p.add_argument("--hip_graph", action="store_true",
                   help="(default since round 3; kept for older command lines) replay each DDIM step's UNet forward from a "
                        "hipGraph: one capture per clip shape and hook plan, bit-equal to kernel-by-kernel launches")
    p.add_argument("--no_hip_graph", action="store_true", help="launch every kernel of a step from the host (VFACE_GRAPH=0)")
    p.add_argument("--drop_dead_branches", action="store_true",
                   help="exact dead-branch elimination (results bit-identical): sampling runs the UNet on [uncond ; cond] without "
                        "the recon third -- its x_prev is dropped by the sampler and no hook mode reads chunk 2 (ddim_w_inv.py:"
                        "667,703-707,738; pnp_utils.py:136-142,195-199,255-256) -- and the inversion runs on the target half only "
                        "(only nosie[:batch_size] is saved, ddim_w_inv.py:464-486): -33 %% / -50 %% of the UNet work")
    p.add_argument("--max_steps", type=int, default=None, help="stop after this many DDIM steps (smoke runs)")
    p.add_argument("--pipeline_inversion", action="store_true",
                   help="run the DDIM inversion of batch k + 1 beside the sampling of batch k, on two HIP streams (the batches are "
                        "independent: VFace_inference_batch.py:413, 529-553); frames bit-identical to the sequential order")
    return p


def load_unet_config(path):
    from ..ldm.models.diffusion.ddpm import FFHQ_UNET_CONFIG
    if path is None:
        return dict(FFHQ_UNET_CONFIG)
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    return dict(cfg["model"]["params"]["unet_config"]["params"])


class _StubUnpickler:
    """``pickle_module`` for ``torch.load`` of a checkpoint written by the reference's pytorch_lightning==1.4.2
    (``REFace/environment.yml``): its ``callbacks`` / ``hyper_parameters`` entries pickle CLASS objects (the ModelCheckpoint
    class as a dict key) and omegaconf containers, which ``weights_only=True`` rejects and which are not importable here.

    ``find_class`` resolves EXACTLY the (module, name) pairs of torch's own ``weights_only`` allow-list
    (``torch._weights_only_unpickler._get_allowed_globals``: the ``_rebuild_*`` helpers, storages, dtypes, ``torch.Size``,
    ``collections.OrderedDict`` ..) plus the numpy scalar / dtype reconstructors -- never a whole package: every other global,
    including every other ``torch.*`` / ``numpy.*`` / ``builtins`` name (``torch.utils.collect_env.run``, ``torch.hub.load``,
    ``numpy.load``, ``os.system`` ..), becomes an inert placeholder class whose construction, call and ``__setstate__`` do
    nothing.  No callable outside the allow-list is ever invoked (``tests/test_host_cpu.py`` feeds it hostile pickles).
    It is used only when the caller asks for it (``load_checkpoint(.., stub_unknown_globals=True)`` /
    ``--ckpt_stub_unknown_globals``), never as a silent fallback."""
    import pickle as _pickle

    _NUMPY_OK = frozenset({("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                           ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"),
                           ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray")})
    _allowed = None

    @classmethod
    def allowed(cls) -> dict:
        if cls._allowed is None:
            from torch._weights_only_unpickler import _get_allowed_globals
            cls._allowed = dict(_get_allowed_globals())
        return cls._allowed

    class Unpickler(_pickle.Unpickler):
        _stubs: dict = {}

        def find_class(self, module, name):
            hit = _StubUnpickler.allowed().get(f"{module}.{name}")
            if hit is not None:
                return hit
            if (module, name) in _StubUnpickler._NUMPY_OK:
                return super().find_class(module, name)
            key = (module, name)
            if key not in self._stubs:
                def _new(cls, *a, **k):
                    return object.__new__(cls)
                self._stubs[key] = type(name, (), {"__module__": module, "__new__": _new, "__init__": lambda self, *a, **k: None,
                                                   "__setstate__": lambda self, st: None, "__call__": lambda self, *a, **k: None,
                                                   "append": lambda self, *a: None, "extend": lambda self, *a: None,
                                                   "__setitem__": lambda self, *a: None})
            return self._stubs[key]

    @staticmethod
    def load(f, **kw):
        return _StubUnpickler.Unpickler(f, **kw).load()

    @staticmethod
    def loads(b, **kw):
        import io
        return _StubUnpickler.Unpickler(io.BytesIO(b), **kw).load()

    __name__ = "vface_stub_pickle"


def load_checkpoint(model, path: str, with_vae: bool = False, stub_unknown_globals: bool = False) -> str:
    """Load ``last.ckpt`` of the reference (``VFace_inference_batch.py:118-135``: ``torch.load`` -> ``["state_dict"]`` ->
    ``load_state_dict(strict=False)``) into ``model``.  Only the keys of what this build runs are taken:
    ``model.diffusion_model.*`` always, ``first_stage_model.*`` when the first stage was built (``with_vae``); the CLIP /
    ArcFace / landmark conditioning weights of a full LDM checkpoint are outside the path and ignored.  Unlike the reference's
    non-strict load, a checkpoint that does not cover every parameter of the path raises instead of silently running on
    default-initialised weights.

    The load is ``weights_only=True``.  A pytorch_lightning checkpoint carries class-keyed callback state that the safe loader
    rejects: pass ``stub_unknown_globals=True`` (``--ckpt_stub_unknown_globals``) to read it through ``_StubUnpickler`` (exact
    allow-list, inert placeholders for everything else).  There is no automatic fallback: a file that fails the safe loader
    raises, naming the flag."""
    import pickle
    if stub_unknown_globals:
        ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_StubUnpickler)
    else:
        try:
            ck = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            raise RuntimeError(
                f"{path}: the weights-only loader rejected this checkpoint ({str(e).splitlines()[0][:200]}).  If it is a "
                "pytorch_lightning checkpoint (callbacks / hyper_parameters pickle class objects), pass "
                "--ckpt_stub_unknown_globals (load_checkpoint(.., stub_unknown_globals=True)): tensors load through torch's own "
                "allow-list and every other global becomes an inert placeholder") from e
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
    prefixes = ("model.diffusion_model.",) + (("first_stage_model.",) if with_vae else ())
    taken = {k: v for k, v in sd.items() if k.startswith(prefixes) and torch.is_tensor(v)}
    if not taken:
        raise RuntimeError(f"{path}: no key starts with {prefixes} -- not an LDM checkpoint of this model "
                           f"(first keys: {list(sd)[:5]})")
    missing, unexpected = model.load_state_dict(taken, strict=False)
    hot = [k for k in missing if k.startswith(prefixes)]
    if hot or unexpected:
        raise RuntimeError(f"{path}: {len(hot)} parameter(s) of the denoising path missing (first: {hot[:5]}), "
                           f"{len(unexpected)} unexpected key(s) (first: {list(unexpected)[:5]})")
    return (f"loaded {path}: {len(taken)} tensors under {', '.join(prefixes)} matched every parameter of the path "
            f"({len(sd) - len(taken)} checkpoint entries outside it ignored)")


def _quad_coeffs(size, quad):
    """Eight PIL perspective coefficients (output pixel -> crop coordinates) that land the ``size`` x ``size`` crop on ``quad``
    (four frame-space corners, clockwise from top-left) -- what ``crop_and_align_face`` (:58-73) stores per frame."""
    import numpy as np
    src = [(0, 0), (size, 0), (size, size), (0, size)]
    A, B = [], []
    for (x, y), (u, v) in zip(quad, src):
        A += [[x, y, 1, 0, 0, 0, -u * x, -u * y], [0, 0, 0, x, y, 1, -v * x, -v * y]]
        B += [u, v]
    return np.linalg.solve(np.array(A, float), np.array(B, float))


def run_synthetic(opt) -> dict:
    from ..ldm.models.diffusion.ddim_w_inv import DDIMSampler, HookPlan
    from ..ldm.models.diffusion.ddpm import LatentDiffusion
    from ..utils import synth

    dev = torch.device("cuda", 0)
    dt = torch.float16 if opt.compute_dtype == "fp16" else torch.bfloat16
    cfg = load_unet_config(opt.config)
    cfg["compute_dtype"] = dt
    if opt.with_vae:
        from ..ldm.models.autoencoder import FFHQ_VAE_CONFIG
        model = LatentDiffusion(cfg, first_stage_config=dict(FFHQ_VAE_CONFIG, compute_dtype=dt))
    else:
        model = LatentDiffusion(cfg)
    if opt.ckpt:
        print(load_checkpoint(model, opt.ckpt, with_vae=opt.with_vae, stub_unknown_globals=opt.ckpt_stub_unknown_globals))
    else:
        synth.fill_module_(model.unet, seed=0)
        if opt.with_vae:
            synth.fill_module_(model.first_stage_model, seed=0, prefix="vae.")
    model = model.to(dev).eval()
    sampler = DDIMSampler(model)
    sampler.hook_plan = HookPlan(fusion=opt.fusion, enabled=opt.fusion != "none")
    sampler.flow_gate = opt.flow_gate
    sampler.drop_dead_branches = bool(opt.drop_dead_branches)
    if opt.hip_graph:
        sampler.model.model.diffusion_model.engine.use_graph = True
    if opt.no_hip_graph:
        sampler.model.model.diffusion_model.engine.use_graph = False
    sampler.flow_resample = "area" if opt.flow_pixels else None
    h, w = opt.H // opt.f, opt.W // opt.f
    F_ = opt.n_samples
    os.makedirs(opt.Base_dir, exist_ok=True)
    results, t_all = [], time.time()
    paster, raft = None, None
    nbatches = opt.n_frames // F_       # DataLoader(batch_size=n_samples, drop_last=True) (:376-382)
    # --pipeline_inversion: the DDIM inversion of batch k + 1 runs beside the sampling of batch k (DDIMSampler.sample_while_inverting:
    # the batches are independent, :413, :529-553); batch 0's inversion runs alone, the last batch's sampling too
    pipelined = bool(getattr(opt, "pipeline_inversion", False)) and not opt.no_inversion and nbatches > 1

    def stage(stages, name, t_start):      # wall seconds of a pipeline stage, GPU drained on both sides
        torch.cuda.synchronize()
        stages[name] = stages.get(name, 0.0) + time.time() - t_start
        return time.time()

    def prepare(batch_id):
        """Everything of a batch up to (not including) its DDIM inversion: conditioning, VAE encoding, flow."""
        nonlocal raft
        tag = lambda s: f"cli.{s}.{batch_id}"
        d = lambda t: t.to(dev)
        stages = {}
        c, uc, tc = (d(synth.synth_normal(tag(k), (F_, 1, 768))) for k in ("c", "uc", "tc"))
        img = None
        if opt.with_vae:
            img = d(torch.stack([synth.synth_normal(tag(f"img{f}"), (3, opt.H, opt.W)).clamp(-1, 1) for f in range(F_)]))
            ts_ = stage(stages, "_", time.time())
            z_inp = model.get_first_stage_encoding(model.encode_first_stage(img)).detach()       # :456-457
            stage(stages, "vae_encode", ts_)
        else:
            z_inp = d(synth.synth_normal(tag("inp"), (F_, opt.C, h, w)) * 0.18215)
        mask = d(synth.synth_mask(F_, h, w))
        if opt.raft_flow:     # :550-553 flow = return_flow(target frames), pixel resolution
            from . import temporal_flow as tflow
            if raft is None:
                from ..raft import RAFT
                raft = RAFT(compute_dtype=dt)
                if opt.raft_ckpt:
                    raft.load_state_dict(torch.load(opt.raft_ckpt, map_location="cpu", weights_only=True))
                else:
                    synth.fill_module_(raft, seed=0, prefix="raft.")
                raft = raft.to(dev).eval()
            video = img if opt.with_vae else d(torch.stack([synth.synth_normal(tag(f"img{f}"), (3, opt.H, opt.W)).clamp(-1, 1)
                                                            for f in range(F_)]))
            ts_ = stage(stages, "_", time.time())
            flow = tflow.return_flow(video, raft)
            stage(stages, "flow", ts_)
            sampler.flow_resample = "area"
        elif opt.flow_pixels:   # a pixel-resolution field whose latent resample is a +-2-cell motion
            flow = [f[None] * opt.f for f in synth.synth_flow(F_ - 1, opt.H, opt.W, seed=opt.seed + batch_id)]
        else:
            flow = [f[None] for f in synth.synth_flow(F_ - 1, h, w, seed=opt.seed + batch_id)]
        inv_store = {}
        invert_kw = None
        if opt.no_inversion:
            sampler.make_schedule(opt.ddim_steps, ddim_eta=opt.ddim_eta, verbose=False)
            for s in sampler.ddim_timesteps:
                inv_store[int(s)] = d(synth.synth_normal(tag(f"inv{int(s)}"), (F_, opt.C, h, w)))
        else:
            # :531-540: invert [target ; source] (2F), hooks off; the target half is cached per timestep
            z2 = d(synth.synth_normal(tag("z2"), (2 * F_, opt.C, h, w)))
            kw2 = {"inpaint_image": torch.cat([z_inp, z_inp]), "inpaint_mask": torch.cat([mask, mask])}
            invert_kw = dict(x=z2, cond=torch.cat([tc, c]), S=opt.ddim_steps, shape=[opt.C, h, w], eta=opt.ddim_eta,
                             unconditional_guidance_scale=opt.scale, unconditional_conditioning=None,
                             inverse_dir=inv_store, batch_size=F_, test_model_kwargs=kw2, max_steps=opt.max_steps)
        return {"id": batch_id, "c": c, "uc": uc, "tc": tc, "z_inp": z_inp, "mask": mask, "flow": flow, "inv_store": inv_store,
                "invert_kw": invert_kw, "inverted": invert_kw is None, "stages": stages}

    def sample_kwargs(b):
        # :541 start code = the cached latent of the second-highest timestep ("ddim_latents_961.pt" at 50 steps)
        sampler.make_schedule(opt.ddim_steps, ddim_eta=opt.ddim_eta, verbose=False)
        ts = [int(s) for s in sampler.ddim_timesteps]
        inv_store = b["inv_store"]
        start_t = ts[-2] if ts[-2] in inv_store else max(inv_store)
        x_T = inv_store[start_t]
        if opt.max_steps is not None:  # smoke runs: the remaining cache entries are never read
            for s in ts:
                inv_store.setdefault(s, x_T)
        return dict(S=opt.ddim_steps, conditioning=b["c"], target_conditioning=b["tc"], inverse_results_dir=inv_store, batch_size=F_,
                    shape=[opt.C, h, w], verbose=False, unconditional_guidance_scale=opt.scale, unconditional_conditioning=b["uc"],
                    eta=opt.ddim_eta, x_T=x_T, flow=b["flow"] if opt.fusion == "flow_fix" else None,
                    test_model_kwargs={"inpaint_image": b["z_inp"], "inpaint_mask": b["mask"]}, max_steps=opt.max_steps)

    cur = prepare(0)
    torch.cuda.synchronize()
    t_batch = time.time()       # a batch's wall time = from the previous batch's last frame to its own (its preparation included)
    for batch_id in range(nbatches):
        stages = cur["stages"]
        if not cur["inverted"]:
            ts_ = stage(stages, "_", time.time())
            sampler.ddim_invert(**cur["invert_kw"])
            stage(stages, "inversion", ts_)
            cur["inverted"] = True
        nxt = prepare(batch_id + 1) if (pipelined and batch_id + 1 < nbatches) else None
        if nxt is not None:                      # (its encoding / flow are this batch's wall time: they run before this batch samples)
            for k, v in nxt["stages"].items():
                if k != "_":
                    stages["next_" + k] = v
            nxt["stages"] = {}
        torch.cuda.synchronize()
        t0 = time.time()
        if nxt is not None:
            (samples, _), _ = sampler.sample_while_inverting(sample_kwargs(cur), nxt["invert_kw"])
            nxt["inverted"] = True
        else:
            samples, _ = sampler.sample(**sample_kwargs(cur))
        torch.cuda.synchronize()
        dt_s = time.time() - t0
        stages["sampling" if nxt is None else "sampling_beside_next_inversion"] = dt_s
        pixels = None
        if opt.with_vae:
            ts_ = time.time()
            x_samples = model.decode_first_stage(samples)                                          # :596
            pixels = torch.clamp((x_samples + 1.0) / 2.0, min=0.0, max=1.0)                       # :597
            stage(stages, "vae_decode", ts_)
        pasted, paste_s = None, None
        if opt.paste_back:
            if not opt.with_vae:
                raise SystemExit("--paste_back pastes decoded pixels: it needs --with_vae")
            from .paste_back import PasteBack
            if paster is None:
                # (--precision autocast, the reference's default: its decoded tensor is float16 and :597-608 quantise in float16)
                paster = PasteBack(H=opt.H, W=opt.W, device=dev, encode_decode=PasteBack.vae_round_trip(model),
                                   half_arithmetic=opt.precision == "autocast")
            S_ = opt.frame_size
            gen = torch.Generator().manual_seed(opt.seed + 1000 + batch_id)
            frames = torch.randint(0, 256, (F_, S_, S_, 3), dtype=torch.uint8, generator=gen).to(dev)
            # inv_transforms_all rows (:625): here the 1024 canvas lands on the central half of the frame, slightly sheared
            q = S_ / 4.0
            co = [_quad_coeffs(paster.canvas, [(q + 3 * f, q), (3 * q, q + f), (3 * q - f, 3 * q), (q, 3 * q - 2 * f)]) for f in range(F_)]
            torch.cuda.synchronize()
            t1 = time.time()
            pasted = paster.paste(x_samples, frames, co)                                           # :603-633
            torch.cuda.synchronize()
            paste_s = time.time() - t1
            stages["paste_back"] = paste_s
        if not opt.skip_save:
            if pasted is not None:
                from PIL import Image
                for f in range(F_):                                                                # :636
                    Image.fromarray(pasted[f].cpu().numpy()).save(os.path.join(opt.Base_dir, f"pasted_b{batch_id}_f{f}.png"))
            torch.save(samples.cpu(), os.path.join(opt.Base_dir, f"samples_batch{batch_id}.pt"))
            if pixels is not None:
                torch.save(pixels.cpu(), os.path.join(opt.Base_dir, f"pixels_batch{batch_id}.pt"))
        torch.cuda.synchronize()
        results.append({"batch": batch_id, "frames": F_, "sample_seconds": dt_s, "batch_wall_seconds": time.time() - t_batch,
                        "samples": samples if getattr(opt, "return_samples", False) else None,
                        "finite": bool(torch.isfinite(samples).all()) and (pixels is None or bool(torch.isfinite(pixels).all())),
                        "pixels": None if pixels is None else list(pixels.shape),
                        "pasted": None if pasted is None else list(pasted.shape), "paste_seconds": paste_s,
                        "stage_seconds": {k: v for k, v in stages.items() if k != "_"}})
        print(f"batch {batch_id}: {F_} frames sampled in {dt_s:.2f} s")
        t_batch = time.time()
        cur = nxt if nxt is not None else (prepare(batch_id + 1) if batch_id + 1 < nbatches else None)
    return {"batches": results, "total_seconds": time.time() - t_all}


def main(argv=None):
    opt = build_parser().parse_args(argv)
    torch.manual_seed(opt.seed)  # seed_everything(42) (:862)
    if opt.plms:
        raise NotImplementedError("--plms selects PLMSSampler, which is outside the VFace hot path (SURVEY §2)")
    if not opt.synthetic:
        sys.exit("Only the denoising hot path is built here.  Video decoding, dlib/BiSeNet pre-processing, CLIP/ArcFace "
                 "conditioning and paste-back (VFace_inference_batch.py:193-528,603-670) need checkpoints "
                 "that are not available offline; run with --synthetic, or feed real latents through "
                 "vface_amd.ldm.models.diffusion.ddim_w_inv.DDIMSampler (see INTEGRATION.md).")
    if not torch.cuda.is_available():
        sys.exit("An MI355X is required: the VFace hot path has no CPU fallback.")
    return run_synthetic(opt)


if __name__ == "__main__":
    main()
