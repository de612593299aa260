#!/usr/bin/env python3
"""A/B: persistent vs one-tile-per-workgroup GEMM / conv on the UNet's shapes, interleaved rounds, median of medians."""
import sys, os, math, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3, pack_geglu
DEV = "cuda:0"
N_ = 24
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).half()

def timeit(fn, iters=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3  # us

def ab(name, mk, flops, extra_flags=()):
    variants = [("persist", 0), ("plain", hip.TUNE_NO_PERSISTENT)] + [(n, f) for n, f in extra_flags]
    res = {n: [] for n, _ in variants}
    for _ in range(4):
        for n, f in variants:
            fn = mk(f); fn(); res[n].append(timeit(fn))
    row = f"{name:34s}"
    for n, _ in variants:
        t = statistics.median(res[n]); row += f" {n} {t:7.1f} us {flops / t / 1e6:6.0f} TF |"
    print(row, flush=True)

def gemm_case(M, N, K, geglu=False, res=True, cs=True):
    a, w = rnd(M, K), rnd(N, K) / math.sqrt(K)
    b = torch.randn(N, device=DEV)
    nout = N // 2 if geglu else N
    out = torch.empty(M, nout, dtype=torch.float16, device=DEV)
    r = rnd(M, nout) if res and not geglu else None
    c = torch.zeros((M + 63) // 64, nout, 2, device=DEV) if cs and not geglu else None
    return lambda f: (lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=nout, bias=b, residual=r, ldr=nout,
                                       flags=f | (hip.EPI_GEGLU if geglu else 0), colstats=c))

def conv_case(H, cin, cout):
    x = rnd(N_, H, H, cin); w = rnd(cout, 9 * cin) / math.sqrt(9 * cin)
    b = torch.randn(cout, device=DEV); rb = torch.randn(N_, cout, device=DEV)
    out = torch.empty(N_, H, H, cout, dtype=torch.float16, device=DEV)
    c = torch.zeros(N_ * H * H // 64, cout, 2, device=DEV)
    return lambda f: (lambda: hip.conv3x3(x, w, out, nimg=N_, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b,
                                          rowbias=rb, colstats=c, flags=f))

M0, M1, M2 = N_ * 4096, N_ * 1024, N_ * 256
for name, M, N, K, gg in [("proj L0 320x320", M0, 320, 320, False), ("qkv L0 960x320", M0, 960, 320, False),
                          ("ff1 L0 geglu 2560x320", M0, 2560, 320, True), ("ff2 L0 320x1280", M0, 320, 1280, False),
                          ("proj L1 640x640", M1, 640, 640, False), ("ff1 L1 geglu 5120x640", M1, 5120, 640, True),
                          ("ff2 L1 640x2560", M1, 640, 2560, False), ("ff1 L2 geglu 10240x1280", M2, 10240, 1280, True)]:
    ab(name, gemm_case(M, N, K, gg), 2.0 * M * N * K)
for H, cin, cout in [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 640, 640), (32, 1280, 640)]:
    ab(f"conv H{H} {cin}->{cout}", conv_case(H, cin, cout), 2.0 * N_ * H * H * cout * 9 * cin)
