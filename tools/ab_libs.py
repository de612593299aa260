#!/usr/bin/env python3
"""A/B of TWO BUILDS of libvface_hip.so in one process, interleaved rounds (guide rule 24): the same C-ABI call timed through
both libraries alternately.  usage: python tools/ab_libs.py <base.so> <new.so> [what ...]
what: attn40 (dh=40, n=4096, 24 samples), attn40s3 (shared-score form, 8 frames x 3 value sets), attn80, attn160,
      gemm:<M>x<N>x<K>[:geglu|:res32] ...   Prints median / min microseconds per build and the ratio; for attention also the
      rel-L2 of each build against an fp64 reference (2 samples)."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip

DEV = "cuda:0"


def bind(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in hip.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


LIBS = {}


def use(name):
    hip._lib = LIBS[name]


def time_us(fn, iters=8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    fn(); ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3


def ab(label, fn, rounds=7, flops=None):
    res = {k: [] for k in LIBS}
    for _ in range(rounds):
        for k in LIBS:
            use(k)
            res[k].append(time_us(fn))
    med = {k: statistics.median(v) for k, v in res.items()}
    names = list(LIBS)
    extra = "" if flops is None else "  " + " / ".join(f"{flops / med[k] / 1e6:7.1f}" for k in names) + " TFLOP/s"
    print(f"{label:42s} " + "  ".join(f"{k}: med {med[k]:8.1f} min {min(res[k]):8.1f} us" for k in names) +
          f"   {names[1]}/{names[0]} = {med[names[1]] / med[names[0]]:.3f}" + extra, flush=True)


def attn_case(dh, n, N, sets=1, variant=0):
    d = 8 * dh
    g = torch.Generator(device=DEV).manual_seed(0)
    qkv = torch.randn(N * sets, n, 3 * d, device=DEV, generator=g).half()
    out = torch.empty(N * sets, n, d, dtype=torch.float16, device=DEV)
    kw = dict(heads=8, n=n, nk=n, dh=dh, ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d,
              bso=n * d, scale=dh ** -0.5, variant=variant)
    if sets > 1:
        run = lambda: hip.attention(qkv, qkv[:, :, d:], qkv[:, :, 2 * d:], out, B=N, v_sets=sets, set_stride=N, **kw)
    else:
        run = lambda: hip.attention(qkv, qkv[:, :, d:], qkv[:, :, 2 * d:], out, B=N, **kw)
    ab(f"attention dh={dh} n={n} B={N} sets={sets} variant={variant}", run, flops=4.0 * n * n * dh * 8 * N * sets)
    sp = lambda t: t.reshape(2, n, 8, dh).permute(0, 2, 1, 3).double()
    r = torch.softmax(sp(qkv[:2, :, :d]) @ sp(qkv[:2, :, d:2 * d]).transpose(-1, -2) * dh ** -0.5, -1) @ sp(qkv[:2, :, 2 * d:])
    r = r.permute(0, 2, 1, 3).reshape(2, n, d)
    outs = {}
    for k in LIBS:
        use(k); out.zero_(); run(); torch.cuda.synchronize()
        outs[k] = out.clone()
        print(f"    {k}: rel-L2 vs fp64 (samples 0-1) {((out[:2].double() - r).norm() / r.norm()).item():.3e}")
    ks = list(LIBS)
    print(f"    outputs bit-identical across builds: {torch.equal(outs[ks[0]], outs[ks[1]])}")


def gemm_case(spec):
    parts = spec.split(":")
    M, N, K = (int(v) for v in parts[0].split("x"))
    mode = parts[1] if len(parts) > 1 else ""
    g = torch.Generator(device=DEV).manual_seed(1)
    a = (torch.randn(M, K, device=DEV, generator=g) * 0.5).half()
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
    bias = torch.randn(N, device=DEV, generator=g)
    if mode == "geglu":
        out = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
        run = lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N // 2, bias=bias, flags=hip.EPI_GEGLU)
    elif mode == "res32":
        r32 = torch.randn(M, N, device=DEV, generator=g)
        o32 = torch.empty(M, N, device=DEV)
        run = lambda: hip.gemm(a, w, None, M=M, N=N, K=K, lda=K, ldc=0, bias=bias, residual32=r32, out32=o32)
    else:
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        run = lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias)
    ab(f"gemm {spec}", run, flops=2.0 * M * N * K)


def conv_case(spec):
    """conv:<cin>x<cout>x<H>[:res32] -- 3x3 stride-1 convolution of 24 H x H images (bias; optional fp32 residual stream)."""
    from vface_amd import packing
    parts = spec.split(":")
    cin, cout, H = (int(v) for v in parts[0].split("x"))
    res32 = "res32" in parts[1:]
    N = 24
    g = torch.Generator(device=DEV).manual_seed(3)
    x = (torch.randn(N * H * H, cin, device=DEV, generator=g) * 0.5).half()
    w = packing.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=torch.Generator().manual_seed(4)) * (9 * cin) ** -0.5).half().to(DEV)
    bias = torch.randn(cout, device=DEV, generator=g)
    out = torch.empty(N * H * H, cout, dtype=torch.float16, device=DEV)
    kw = dict(nimg=N, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=bias)
    if res32:
        r32 = torch.randn(N * H * H, cout, device=DEV, generator=g)
        o32 = torch.empty(N * H * H, cout, device=DEV)
        kw.update(residual32=r32, out32=o32)
    run = lambda: hip.conv3x3(x, w, out, **kw)
    ab(f"conv3x3 {spec}", run, flops=2.0 * N * H * H * cout * 9 * cin)
    outs = {}
    for k in LIBS:
        use(k); out.zero_(); run(); torch.cuda.synchronize()
        outs[k] = out.clone()
    ks = list(LIBS)
    d = (outs[ks[0]].float() - outs[ks[1]].float())
    print(f"    outputs bit-identical across builds: {torch.equal(outs[ks[0]], outs[ks[1]])}; rel-L2 of the difference {(d.norm() / outs[ks[0]].float().norm()).item():.2e}; "
          f"elements that differ {(d != 0).float().mean().item():.2e}")


if __name__ == "__main__":
    LIBS["base"], LIBS["new"] = bind(sys.argv[1]), bind(sys.argv[2])
    for what in sys.argv[3:] or ["attn40", "attn40s3", "attn80", "attn160"]:
        var = int(what.split(":v")[1]) if ":v" in what else 0
        what = what.split(":v")[0]
        if what == "attn40": attn_case(40, 4096, 24, variant=var)
        elif what == "attn40s3": attn_case(40, 4096, 8, sets=3, variant=var)
        elif what == "attn80": attn_case(80, 1024, 24, variant=var)
        elif what == "attn160": attn_case(160, 256, 24)
        elif what.startswith("gemm:"): gemm_case(what[5:])
        elif what.startswith("conv:"): conv_case(what[5:])
        else: raise SystemExit(f"unknown case {what}")
