#!/usr/bin/env python3
"""The 256 x 320 tile (csrc/gemm_big.hip) on the epilogue forms the 640- / 1280-channel transformer blocks run it with -- plain (qkv,
proj_in), GEGLU (ff1), + 16-bit residual rows (to_out, ff2 with the block's interior sums in 16 bits), + fp32 residual rows -- at the
48- and 96-sample batches.  For same-box A/B runs of two builds: VFACE_HIP_LIB=<other .so> python tools/bench_gemm_big_epilogue.py

usage (GPU box): python tools/bench_gemm_big_epilogue.py > gpurun_out/epi.txt"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vface_amd import hip  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for _ in range(3):
        fn()
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    print(f"lib: {hip.LIB_PATH}")
    for samples in (48, 96):
        for lvl, (n, c) in (("L1", (1024, 640)), ("L2", (256, 1280))):
            M = samples * n
            for name, N, K, kind in ((f"proj_in {lvl}", c, c, "plain"), (f"qkv {lvl}", 3 * c, c, "plain"), (f"ff1 {lvl}", 8 * c, c, "geglu"),
                                     (f"to_out {lvl}", c, c, "res16"), (f"ff2 {lvl}", c, 4 * c, "res16"), (f"ff2/32 {lvl}", c, 4 * c, "res32")):
                a = torch.randn(M, K, device=DEV, generator=g).half()
                w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).half()
                bias = torch.randn(N, device=DEV, generator=g)
                out = torch.empty(M, N // 2 if kind == "geglu" else N, dtype=torch.float16, device=DEV)
                kw = {"plain": dict(bias=bias), "geglu": dict(bias=bias, flags=hip.EPI_GEGLU),
                      "res16": dict(bias=bias, residual=torch.randn(M, N, device=DEV, generator=g).half(), ldr=N),
                      "res32": dict(bias=bias, residual32=torch.randn(M, N, device=DEV, generator=g))}[kind]
                fl0 = kw.pop("flags", 0)
                t = {}
                for tag, fl in (("old", hip.TUNE_NO_BIG_TILE), ("big", hip.TUNE_BIG_TILE)):
                    t[tag] = timeit(lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=out.shape[1], flags=fl | fl0, **kw))
                fl = 2.0 * M * N * K
                print(f"{name:12s} {kind:6s} M {M:6d} N {N:6d} K {K:5d} tiles {((M + 255) // 256) * (N // 320):5d} | 128-row {t['old']:7.1f} us {fl / t['old'] / 1e6:5.0f} TF | "
                      f"256x320 {t['big']:7.1f} us {fl / t['big'] / 1e6:5.0f} TF", flush=True)
                del a, w, out, kw


if __name__ == "__main__":
    main()
