#!/usr/bin/env python3
"""tools/vs_library.py on the plain-GEMM shapes of the 32-frame batch (BASELINE configs[2]: 96 samples per step; 48 per launch
stream): the hand-written 128-row kernel, its 256-row patch form, and torch.mm (hipBLASLt)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
DEV = "cuda:0"
def timeit(fn, iters=10):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for _ in range(3): fn()
    ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3
g = torch.Generator(device=DEV).manual_seed(0)
shapes = [("ff1 L1", 98304, 5120, 640), ("ff2 L1", 98304, 640, 2560), ("qkv L1", 98304, 1920, 640), ("ff1 L2", 24576, 10240, 1280),
          ("ff2 L2", 24576, 1280, 5120), ("qkv L2", 24576, 3840, 1280),
          ("ff1 L1 h", 49152, 5120, 640), ("ff2 L1 h", 49152, 640, 2560), ("ff1 L2 h", 12288, 10240, 1280), ("ff2 L2 h", 12288, 1280, 5120)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=DEV, generator=g).half(); w = torch.randn(N, K, device=DEV, generator=g).half()
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    to = timeit(lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N))
    try:
        tp = timeit(lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, flags=0x100000))
    except Exception as e:
        tp = float("nan")
    tl = timeit(lambda: torch.mm(a, w.t(), out=out))
    fl = 2.0 * M * N * K
    print(f"{name:9s} M{M:6d} N{N:5d} K{K:5d}: ours {to:7.1f} us {fl / to / 1e6:6.0f} TF | 256-row {tp:7.1f} us {fl / tp / 1e6:6.0f} TF | library {tl:7.1f} us {fl / tl / 1e6:6.0f} TF", flush=True)
