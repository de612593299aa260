"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header declares,
the ctypes table matches the header's argument counts, and the weight packing is an exact refactoring."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT
from oracle import hooks as ohooks
from vface_amd import packing

HEADER = os.path.join(ROOT, "include", "vface_hip.h")
LIB = os.path.join(ROOT, "vface_amd", "lib", "libvface_hip.so")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int64_t|int|size_t|const char\*)\s+(vface_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return decls


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(LIB)
    decls = _declared()
    assert len(decls) >= 20
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in include/vface_hip.h but not exported"
    lib.vface_abi_version.restype = ctypes.c_int
    assert lib.vface_abi_version() == 7


def test_ctypes_table_matches_header():
    from vface_amd import hip
    decls = _declared()
    assert set(decls) == set(hip.SIGNATURES)
    for name, n in decls.items():
        assert len(hip.SIGNATURES[name][1]) == n, name


def test_no_gpu_means_loud_failure():
    from vface_amd import hip
    hip.load()
    a = torch.zeros(8, 8, dtype=torch.float16)
    with pytest.raises(hip.VFaceHipError):
        hip.gemm(a, a, a, M=8, N=8, K=8, lda=8, ldc=8)


@pytest.mark.parametrize("d,ratio", [(64, 0.8), (320, 0.8), (320, 0.5)])
def test_fold_fsai_is_exact(d, ratio):
    g = torch.Generator().manual_seed(0)
    wq, wk = torch.randn(d, d, generator=g) / d ** 0.5, torch.randn(d, d, generator=g) / d ** 0.5
    x_own, x_st = torch.randn(5, d, generator=g), torch.randn(5, d, generator=g)
    wl = packing.fold_fsai(wq, wk, ratio).double()
    got = torch.cat([x_own, x_st], 1).double() @ wl.t()
    for i, w in enumerate((wq, wk)):
        ref = ohooks.combine_fft_high_low(x_st @ w.t(), x_own @ w.t(), ratio)
        assert (got[:, i * d:(i + 1) * d].float() - ref).abs().max() < 5e-6


def test_pack_geglu_and_conv():
    g = torch.Generator().manual_seed(1)
    d = 32
    w, b = torch.randn(8 * d, d, generator=g), torch.randn(8 * d, generator=g)
    wp, bp = packing.pack_geglu(w, b)
    x = torch.randn(3, d, generator=g)
    y = x @ wp.t() + bp
    y = y.reshape(3, -1, 2, 16)
    val, gate = y[:, :, 0].reshape(3, -1), y[:, :, 1].reshape(3, -1)
    ref_v, ref_g = (x @ w.t() + b).chunk(2, -1)
    assert torch.allclose(val, ref_v, atol=1e-5) and torch.allclose(gate, ref_g, atol=1e-5)
    cw = torch.randn(6, 9, 3, 3, generator=g)
    pw = packing.pack_conv3x3(cw)
    assert pw.shape == (6, 9 * 16)
    xi = torch.randn(1, 9, 5, 5, generator=g)
    ref = F.conv2d(xi, cw, padding=1)
    cols = F.unfold(F.pad(xi, (0, 0, 0, 0, 0, 7)), 3, padding=1)  # [1, 16*9, 25], (c, ky, kx) order
    cols = cols.reshape(1, 16, 9, 25).permute(0, 2, 1, 3).reshape(1, 144, 25)  # -> (ky, kx, c)
    assert torch.allclose((pw @ cols[0]).reshape(1, 6, 5, 5), ref, atol=1e-5)


def test_every_entry_point_rejects_null_arguments_without_a_gpu():
    """Error behaviour of the C ABI: invalid arguments give a negative VFACE_ERR_* and nothing is launched
    -- checked here with all-zero arguments, which every launcher refuses before its first HIP call."""
    from vface_amd import hip
    lib = hip.load()
    import ctypes as C
    checked = 0
    for name, (restype, argtypes) in hip.SIGNATURES.items():
        if restype is not C.c_int or not any(a is C.c_void_p for a in argtypes):
            continue
        if name in ("vface_attention_shared_scores_supported",):
            continue
        args = [None if (a is C.c_void_p or a is hip._s32p) else (0.0 if a is C.c_float else 0) for a in argtypes]
        rc = getattr(lib, name)(*args)
        assert rc < 0, (name, rc)
        msg = lib.vface_error_string(rc)
        assert msg and b"unknown" not in msg.lower(), (name, rc, msg)
        checked += 1
    assert checked >= 15
    assert lib.vface_splitk_workspace_bytes(0, 0, 0, 0, 0) == 0
    assert lib.vface_attn1_workspace_bytes(0, 0, 0, 0) == 0
    assert lib.vface_attention_shared_scores_supported(40, 3) == 1 and lib.vface_attention_shared_scores_supported(80, 3) == 0


@pytest.mark.parametrize("cin,cout,H,W", [(5, 7, 4, 6), (64, 8, 3, 3)])
def test_upsample_phase_weights_are_exact(cin, cout, H, W):
    """conv3x3(nearest x2 upsample) == the four parity-phase 2x2 convolutions with the pre-summed taps of
    packing.pack_upsample_phases (checked here on CPU with the packed K order undone)."""
    import torch.nn.functional as F
    from vface_amd.packing import pack_upsample_phases
    g = torch.Generator().manual_seed(3)
    w = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64)
    x = torch.randn(2, cin, H, W, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    packed = pack_upsample_phases(w.float()).double()           # [4, cout, 4 * cin_pad]
    cp = (cin + 7) // 8 * 8
    out = torch.zeros_like(ref)
    for py in (0, 1):
        for px in (0, 1):
            k = packed[2 * py + px]
            if cp % 64 == 0:                                     # (chunk, tap, channel) -> (tap, channel)
                k = k.reshape(cout, cp // 64, 4, 64).permute(0, 2, 1, 3).reshape(cout, 4, cp)
            else:
                k = k.reshape(cout, 4, cp)
            k = k[:, :, :cin].reshape(cout, 2, 2, cin).permute(0, 3, 1, 2)
            out[:, :, py::2, px::2] = F.conv2d(F.pad(x, (1 - px, px, 1 - py, py)), k)
    assert torch.allclose(out, ref, atol=1e-5)                   # weights went through fp32 once


def test_bench_self_launches_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` (the driver's form, no rendezvous in the environment) must start N ranks itself, as a
    child process, before anything touches the GPU; a rank started by torch.distributed.run must not re-launch."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("parent touched the GPU")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_lightning_style_checkpoint_loads_with_and_without_vae(tmp_path):
    """``--ckpt`` must take the reference's real ``last.ckpt`` layout (VFace_inference_batch.py:118-135): a
    pytorch_lightning 1.4 checkpoint = state_dict of the WHOLE LatentDiffusion (UNet + first stage + conditioning encoders)
    plus ``callbacks`` keyed by a callback CLASS and other non-tensor entries.  Loads with and without the first stage built;
    a checkpoint that misses a UNet parameter raises."""
    import sys
    import types
    from vface_amd.ldm.models.autoencoder import FFHQ_VAE_CONFIG
    from vface_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from vface_amd.scripts.VFace_inference_batch import load_checkpoint
    from vface_amd.utils import synth
    ucfg = dict(image_size=32, in_channels=9, out_channels=4, model_channels=32, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                transformer_depth=1, context_dim=768, legacy=False)
    vcfg = dict(FFHQ_VAE_CONFIG, ddconfig=dict(FFHQ_VAE_CONFIG["ddconfig"], ch=32, resolution=32))
    src = LatentDiffusion(ucfg, first_stage_config=vcfg)
    synth.fill_module_(src.unet, seed=3)
    synth.fill_module_(src.first_stage_model, seed=3, prefix="vae.")
    sd = dict(src.state_dict())
    assert any(k.startswith("model.diffusion_model.") for k in sd) and any(k.startswith("first_stage_model.") for k in sd)
    sd["cond_stage_model.mapper.weight"] = torch.zeros(4, 4)          # conditioning encoders: outside the path
    sd["learnable_vector"] = torch.zeros(1, 1, 768)
    sd["face_ID_model.facenet.input_layer.0.weight"] = torch.zeros(2, 2)
    # a module that exists only while the checkpoint is WRITTEN (as pytorch_lightning does on the training machine)
    fake = types.ModuleType("pytorch_lightning_fake_callbacks")

    class ModelCheckpoint:
        pass
    ModelCheckpoint.__module__ = fake.__name__
    ModelCheckpoint.__qualname__ = "ModelCheckpoint"
    fake.ModelCheckpoint = ModelCheckpoint
    sys.modules[fake.__name__] = fake
    path = str(tmp_path / "last.ckpt")
    try:
        torch.save({"epoch": 3, "global_step": 1234, "pytorch-lightning_version": "1.4.2", "state_dict": sd,
                    "callbacks": {ModelCheckpoint: {"best_model_score": torch.tensor(0.5), "best_model_path": "x"}},
                    "hyper_parameters": ModelCheckpoint()}, path)
    finally:
        del sys.modules[fake.__name__]
    # no silent fallback from the weights-only loader: the Lightning layout needs the explicit flag
    with pytest.raises(RuntimeError, match="ckpt_stub_unknown_globals"):
        load_checkpoint(LatentDiffusion(ucfg), path)
    for with_vae in (False, True):
        dst = LatentDiffusion(ucfg, first_stage_config=vcfg if with_vae else None)
        msg = load_checkpoint(dst, path, with_vae=with_vae, stub_unknown_globals=True)
        assert "matched every parameter" in msg
        for k, v in dst.state_dict().items():
            assert torch.equal(v, sd[k]), k
    # a checkpoint without one UNet tensor must not run on default-initialised weights
    bad = {k: v for k, v in sd.items() if k != "model.diffusion_model.out.2.weight"}
    torch.save({"state_dict": bad}, path)
    with pytest.raises(RuntimeError, match="missing"):
        load_checkpoint(LatentDiffusion(ucfg), path)
    torch.save({"state_dict": {"something.else": torch.zeros(1)}}, path)
    with pytest.raises(RuntimeError, match="not an LDM checkpoint"):
        load_checkpoint(LatentDiffusion(ucfg), path)


def test_stub_unpickler_never_calls_a_global_outside_the_allow_list(tmp_path, monkeypatch):
    """the placeholder unpickler used for Lightning checkpoints must resolve EXACT (module, name) pairs of
    torch's weights-only allow-list and nothing else -- a pickle that names ``torch.utils.collect_env.run``, ``os.system``,
    ``builtins.eval``, ``torch.hub.load`` or ``numpy.load`` gets an inert placeholder, and nothing runs."""
    import os as _os
    import pickle
    import torch.utils.collect_env as ce
    from vface_amd.scripts.VFace_inference_batch import _StubUnpickler
    called = []
    monkeypatch.setattr(ce, "run", lambda *a, **k: called.append(("collect_env.run", a)) or (0, "", ""))
    monkeypatch.setattr(_os, "system", lambda *a, **k: called.append(("os.system", a)) or 0)
    marker = tmp_path / "pwned"
    hostile = [
        b"ctorch.utils.collect_env\nrun\n(Vtouch " + str(marker).encode() + b"\ntR.",      # the advisor's demonstration
        b"cos\nsystem\n(Vtouch " + str(marker).encode() + b"\ntR.",
        b"cposix\nsystem\n(Vtouch " + str(marker).encode() + b"\ntR.",
        b"cbuiltins\neval\n(V__import__('os').system('touch " + str(marker).encode() + b"')\ntR.",
        b"cbuiltins\nexec\n(Vimport os\ntR.",
        b"ctorch.hub\nload\n(Vx\nVy\ntR.",
        b"cnumpy\nload\n(V/etc/passwd\ntR.",
        b"ctorch.utils.cpp_extension\nload\n(Vx\n(lp0\ntR.",
        b"csubprocess\ncheck_output\n((Vtouch\nV" + str(marker).encode() + b"\nltR.",
    ]
    for payload in hostile:
        obj = _StubUnpickler.loads(payload)
        assert type(obj) in _StubUnpickler.Unpickler._stubs.values(), payload       # an inert placeholder instance
    assert not called and not marker.exists()
    # what a checkpoint legitimately holds still loads: tensors, OrderedDict, numpy scalars, torch.Size, dtypes
    import collections
    import numpy as np
    good = {"sd": collections.OrderedDict(w=torch.arange(6.).reshape(2, 3)), "step": np.int64(7), "shape": torch.Size([2, 3]),
            "dt": torch.float16}
    path = str(tmp_path / "ok.ckpt")
    torch.save(good, path)
    back = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_StubUnpickler)
    assert torch.equal(back["sd"]["w"], good["sd"]["w"]) and back["shape"] == good["shape"] and back["dt"] is torch.float16
    assert int(back["step"]) == 7
    # every name the unpickler resolves for real is on torch's list or the six numpy reconstructors: no package-wide rule
    assert "torch.utils.collect_env.run" not in _StubUnpickler.allowed() and "builtins.eval" not in _StubUnpickler.allowed()
    assert pickle  # (imported for the payload format above)


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_multi_gpu_command_line_defaults_to_the_shipped_schedule():
    """the driver's `bench.py --gpus N --steps K --warmup W` with N > 1 must time the workload that HAS
    the halo exchange -- the shipped schedule (flow_fix, ddim_w_inv.py:303-305) at config 4's per-GPU share (16 frames) --
    while N = 1 is BASELINE configs[2] (32 frames, fft: the largest single-GPU configuration); explicit flags win."""
    bench = _bench_module()
    for n in (2, 4, 8):
        a = bench.parse(["--gpus", str(n), "--steps", "20", "--warmup", "5"])
        assert (a.fusion, a.frames, a.exchange) == ("flow_fix", 16, "p2p")
    a = bench.parse(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert (a.fusion, a.frames) == ("fft", 32)
    a = bench.parse(["--gpus", "8", "--fusion", "replace", "--frames", "8", "--exchange", "allgather"])
    assert (a.fusion, a.frames, a.exchange) == ("replace", 8, "allgather")


def test_bench_refuses_a_traffic_summary_from_other_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic is quoted from profiles/*_hbm_traffic.json only when that summary records the kernel sources this
    tree is built from (it used to go stale silently)."""
    import json
    from vface_amd.utils.buildinfo import source_sha16
    bench = _bench_module()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    body = {"gemm_kernel<F16, 0, 5, true, false, 2>": {"launches": 10, "hbm_bytes_per_launch": 100.0},
            "gemm_kernel<F16, 0, 4, true, false, 0>": {"launches": 30, "hbm_bytes_per_launch": 300.0},
            "conv_patch_kernel<F16, 5, 3, 3, false, false, 0, false>": {"launches": 5, "hbm_bytes_per_launch": 7.0}}
    (prof / "r99_hbm_traffic.json").write_text(json.dumps(dict(body, _build={"source_sha16": "0" * 16})))
    t, why = bench.traffic_from_profiles(("gemm_kernel<F16, 0,",), True)
    assert t is None and "not quoted" in why
    (prof / "r99_hbm_traffic.json").write_text(json.dumps(body))          # no build record at all
    assert bench.traffic_from_profiles(("gemm_kernel<F16, 0,",), True)[0] is None
    (prof / "r99_hbm_traffic.json").write_text(json.dumps(dict(body, _build={"source_sha16": source_sha16()})))
    t, why = bench.traffic_from_profiles(("gemm_kernel<F16, 0,",), True)
    assert t == (10 * 100.0 + 30 * 300.0) / 40 and "QUOTED" in why
    assert bench.traffic_from_profiles(("gemm_kernel<F16, 0,",), False)[0] is None     # another workload: never quoted


def test_workspace_domain_nests_and_restores():
    """hip.workspace_domain: the split-K scratch key of launches issued inside (two launch sequences on two streams must not share
    one scratch); nesting restores the outer domain, also on an exception."""
    from vface_amd import hip
    assert hip._ws_domain == 0
    with hip.workspace_domain(1):
        assert hip._ws_domain == 1
        with hip.workspace_domain(2):
            assert hip._ws_domain == 2
        assert hip._ws_domain == 1
        try:
            with hip.workspace_domain(3):
                raise RuntimeError("x")
        except RuntimeError:
            pass
        assert hip._ws_domain == 1
    assert hip._ws_domain == 0


def test_split_plan_halves_frames_across_chunks_and_refuses_coupled_modes():
    """UNetEngine._split_plan (the two launch streams): frames [0, F/2) and [F/2, F) of EVERY chunk; hook modes that read more than one
    other frame (temporal, adaIn), odd frame counts, batches under 12 samples and sharded engines keep one launch sequence;
    a batch without its last chunk (live_chunks) is halved over the chunks it has."""
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler, HookPlan
    from vface_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    cfg = dict(image_size=32, in_channels=9, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
               channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768, legacy=False)
    ldm = LatentDiffusion(cfg)
    sampler = DDIMSampler(ldm)
    eng = ldm.unet.engine
    assert eng.split_streams == 2
    for fusion in ("replace", "fft", "mix"):
        sampler.hook_plan = HookPlan(fusion=fusion)
        sampler._register_step_hooks(None)
        a, b = (t.tolist() for t in eng._split_plan(24))
        assert a == [0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19] and b == [4, 5, 6, 7, 12, 13, 14, 15, 20, 21, 22, 23]
        assert eng._split_plan(9) is None and eng._split_plan(15) is None      # too small / odd frame count
    for fusion in ("temporal", "adaIn"):
        sampler.hook_plan = HookPlan(fusion=fusion)
        sampler._register_step_hooks(None)
        assert eng._split_plan(24) is None, fusion
    # flow_fix reads exactly one neighbour frame: its halves run as two coupled in-process shards (parallel.StreamShard, round 5);
    # without a flow field it is "fft" (no coupling); a field count that does not match the frames keeps the batch whole
    import torch
    sampler.hook_plan = HookPlan(fusion="flow_fix")
    sampler._register_step_hooks(None)
    assert eng._split_plan(24) is not None and not eng._split_coupled
    sampler._register_step_hooks(torch.zeros(7, 2, 4, 4))
    a, b = (t.tolist() for t in eng._split_plan(24))
    assert a == [0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19] and len(eng._split_coupled) == 1
    sampler._register_step_hooks(torch.zeros(5, 2, 4, 4))
    assert eng._split_plan(24) is None
    sampler.hook_plan = HookPlan(fusion="replace", enabled=False)
    sampler._register_step_hooks(None)
    a, b = (t.tolist() for t in eng._split_plan(16))                            # unhooked (inversion): plain halves
    assert a == list(range(8)) and b == list(range(8, 16))
    sampler.hook_plan = HookPlan(fusion="fft")
    sampler._register_step_hooks(None)
    eng.live_chunks = 2                                                         # [uncond ; cond] of 8 frames
    a, b = (t.tolist() for t in eng._split_plan(16))
    assert a == [0, 1, 2, 3, 8, 9, 10, 11] and b == [4, 5, 6, 7, 12, 13, 14, 15]
    eng.live_chunks = None
    eng.halo_exchange = object()                                                # frames sharded over ranks
    assert eng._split_plan(24) is None
    eng.halo_exchange = None
    eng.split_streams = 1
    assert eng._split_plan(24) is None


def test_every_built_kernel_keeps_its_values_in_registers():
    """no instantiation that ships in libvface_hip.so carries scratch (a spilled kernel is a kernel whose schedule
    nobody chose) -- read from the code-object notes of csrc/build/*.o (tools/codeobj_audit.py; no GPU needed).  Round 5 still had
    eight: an attention A/B form, the 160-wide fused-GroupNorm convolution and the 160-wide 2 x 2 window with a residual operand;
    they are no longer built (conv.hip launch_patch / vf_conv_patch_tile hand those launches the 128-wide tile or refuse them)."""
    import glob
    import sys
    import tempfile
    objs = sorted(glob.glob(os.path.join(ROOT, "vface_amd", "csrc", "build", "*.o")))
    if not objs or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no build objects / no llvm-readelf here (the driver's build() leaves both in this container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import codeobj_audit as audit
    spilled, n = [], 0
    with tempfile.TemporaryDirectory() as td:
        for obj in objs:
            co = audit.extract(obj, td)
            if co is None:
                continue
            for sym, k in audit.notes(co).items():
                n += 1
                # (SGPR spills go to lanes of a VGPR, not to memory: they are not scratch)
                if int(k[".private_segment_fixed_size"]) or int(k[".vgpr_spill_count"]):
                    spilled.append((sym, int(k[".private_segment_fixed_size"])))
    assert n > 200, n
    assert not spilled, spilled


def test_shipped_flow_warp_has_no_select_behind_a_packed_product():
    """Round 5 / 6: the one instruction schedule that ever mis-executed beside another queue's attention waves had a
    ``v_cndmask_b32`` reading, within three issue slots, the result of a packed-fp32 instruction (``v_pk_mul_f32 v[40:41]`` ->
    ``v_cndmask_b32_e32 v40, 0, v40, vcc``).  The shipped warp forms its weights without selects; this reads its listing from the build
    (tools/codeobj_audit.py, no GPU) and fails if hipcc ever brings that pair back."""
    import sys
    import tempfile
    obj = os.path.join(ROOT, "vface_amd", "csrc", "build", "pointwise.hip.o")
    if not os.path.exists(obj) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no build objects / no llvm-objdump here")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import codeobj_audit as audit
    sites = []
    with tempfile.TemporaryDirectory() as td:
        co = audit.extract(obj, td)
        assert co is not None
        stats = audit.scan(co, sites)
    warp = [k for k in stats if "flow_warp_kernel" in k]
    assert warp, "flow_warp_kernel not found in pointwise.hip.o"
    bad = [(k, d, p, c) for k, d, p, c in sites if "flow_warp_kernel" in k and c.startswith("v_cndmask")]
    assert not bad, bad
