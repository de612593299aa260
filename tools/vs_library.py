#!/usr/bin/env python3
"""Where the hand-written GEMM stands next to the vendor library (torch.mm -> hipBLASLt / rocBLAS) on the UNet's plain
GEMM shapes (no epilogue on either side)."""
import sys, os, math, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
DEV = "cuda:0"
def timeit(fn, iters=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for _ in range(3): fn()
    ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3
g = torch.Generator(device=DEV).manual_seed(0)
for name, M, N, K in [("qkv L0", 98304, 960, 320), ("proj L0", 98304, 320, 320), ("ff2 L0", 98304, 320, 1280), ("qkv L1", 24576, 1920, 640),
                      ("ff2 L1", 24576, 640, 2560), ("qkv L2", 6144, 3840, 1280), ("ff2 L2", 6144, 1280, 5120), ("big", 16384, 8192, 8192)]:
    a = torch.randn(M, K, device=DEV, generator=g).half(); w = torch.randn(N, K, device=DEV, generator=g).half()
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    res = {"ours": [], "lib": []}
    for _ in range(3):
        res["ours"].append(timeit(lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N)))
        res["lib"].append(timeit(lambda: torch.mm(a, w.t(), out=out)))
    fl = 2.0 * M * N * K
    to, tl = statistics.median(res["ours"]), statistics.median(res["lib"])
    print(f"{name:8s} M{M:6d} N{N:5d} K{K:5d}: ours {to:7.1f} us {fl / to / 1e6:6.0f} TF | library {tl:7.1f} us {fl / tl / 1e6:6.0f} TF", flush=True)
