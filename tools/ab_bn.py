#!/usr/bin/env python3
"""A/B: BN = 128 vs 160 tiles for the UNet's convs / GEMMs with N a multiple of both (640, 1280), interleaved rounds."""
import sys, os, math, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
DEV = "cuda:0"
N_ = 24
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).half()

def timeit(fn, iters=12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3

def ab(name, mk, flops, variants):
    res = {n: [] for n, _ in variants}
    for _ in range(5):
        for n, f in variants:
            fn = mk(f); fn(); res[n].append(timeit(fn))
    row = f"{name:30s}"
    for n, _ in variants:
        t = statistics.median(res[n]); row += f" {n} {t:7.1f} us {flops / t / 1e6:6.0f} TF |"
    print(row, flush=True)

def conv_case(H, cin, cout, up=False):
    x = rnd(N_, H, H, cin); w = rnd(cout, 9 * cin) / math.sqrt(9 * cin)
    b = torch.randn(cout, device=DEV); rb = torch.randn(N_, cout, device=DEV)
    OH = 2 * H if up else H
    out = torch.empty(N_, OH, OH, cout, dtype=torch.float16, device=DEV)
    c = torch.zeros(N_ * OH * OH // 64, cout, 2, device=DEV)
    return lambda f: (lambda: hip.conv3x3(x, w, out, nimg=N_, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b,
                                          rowbias=rb, colstats=c, flags=f, upsample=up))

def gemm_case(M, N, K, res=True):
    a, w = rnd(M, K), rnd(N, K) / math.sqrt(K)
    b = torch.randn(N, device=DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    r = rnd(M, N) if res else None
    return lambda f: (lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=b, residual=r, ldr=N, flags=f))

V = [("bn128", 0x500), ("bn160", 0x600)]
for H, cin, cout, up in [(32, 320, 640, False), (32, 640, 640, False), (32, 1280, 640, False), (32, 960, 640, False),
                         (16, 640, 1280, False), (16, 1280, 1280, False), (16, 2560, 1280, False), (16, 1920, 1280, False),
                         (32, 640, 640, True), (16, 1280, 1280, True)]:
    OH = 2 * H if up else H
    ab(f"conv H{H}{'up' if up else ''} {cin}->{cout}", conv_case(H, cin, cout, up), 2.0 * N_ * OH * OH * cout * 9 * cin, V)
M1, M2 = N_ * 1024, N_ * 256
for name, M, N, K in [("L1 640x640", M1, 640, 640), ("L1 ff2 640x2560", M1, 640, 2560), ("L1 qkv 1920x640", M1, 1920, 640),
                      ("L2 1280x1280", M2, 1280, 1280), ("L2 ff2 1280x5120", M2, 1280, 5120), ("L2 qkv 3840x1280", M2, 3840, 1280)]:
    ab(name, gemm_case(M, N, K, "qkv" not in name), 2.0 * M * N * K, V)
