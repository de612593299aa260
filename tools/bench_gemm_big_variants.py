#!/usr/bin/env python3
"""A/B of the DMA-issue placements of csrc/gemm_big.hip (VFACE_BIG_VARIANT, experiment builds only)."""
import os, statistics, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
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
    out = []
    for name, M, N, K in [("qkv L1", 98304, 1920, 640), ("ff2 L1", 98304, 640, 2560), ("qkv L2", 24576, 3840, 1280), ("ff2 L2x", 24576, 2560, 5120), ("sq", 16384, 8320, 8192)]:
        a = torch.randn(M, K, device=DEV, generator=g).half(); w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).half()
        o = torch.empty(M, N, dtype=torch.float16, device=DEV)
        t = timeit(lambda: hip.gemm(a, w, o, M=M, N=N, K=K, lda=K, ldc=N, flags=hip.TUNE_BIG_TILE))
        out.append(f"{name} {t:7.1f} us {2.0 * M * N * K / t / 1e6:5.0f} TF")
        del a, w, o
    print(" | ".join(out), flush=True)
else:
    for v in sys.argv[1:] or ["0", "1", "2", "3", "4"]:
        env = dict(os.environ, VFACE_BIG_VARIANT=v)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(f"variant {v}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
