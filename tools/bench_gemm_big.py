#!/usr/bin/env python3
"""The 256 x 320 tile (csrc/gemm_big.hip) next to the 128-row kernel and to torch.mm (hipBLASLt) on the plain-GEMM launches of the
640- / 1280-channel transformer blocks, WITH the epilogues the UNet runs them with (qkv: none; ff1: bias + GEGLU; ff2: bias + fp32
residual rows), at the 24-, 48- and 96-sample batches (8 frames; one launch stream's half of 32 frames; 32 frames).

usage (GPU box): python tools/bench_gemm_big.py > gpurun_out/gemm_big.txt"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vface_amd import hip  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=10):
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
    print(f"{'launch':10s} {'M':>6s} {'N':>6s} {'K':>5s} {'tiles':>6s} | {'128-row us':>10s} {'TF':>5s} | {'256x320 us':>10s} {'TF':>5s} | {'library us':>10s} {'TF':>5s}  (library: no epilogue)")
    o32 = cs = rbv = None
    for samples in (24, 48, 96):
        for lvl, (n, c) in (("L1", (1024, 640)), ("L2", (256, 1280)), ("mid", (64, 1280))):
            M = samples * n
            for name, N, K, kind in ((f"qkv {lvl}", 3 * c, c, "plain"), (f"ff1 {lvl}", 8 * c, c, "geglu"), (f"ff2 {lvl}", c, 4 * c, "res32"),
                                     (f"fsai {lvl}", 2 * c, 2 * c, "a2"), (f"proj_in {lvl}", c, c, "o32"), (f"to_out {lvl}", c, c, "rb_r32_o32"),
                                     (f"proj_out {lvl}", c, c, "r32_o32_o16_cs")):
                Mx = M // 3 if kind == "a2" else M
                a = torch.randn(Mx, K if kind != "a2" else K // 2, device=DEV, generator=g).half()
                a2 = torch.randn(Mx, K // 2, device=DEV, generator=g).half() if kind == "a2" else None
                w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).half()
                bias = torch.randn(N, device=DEV, generator=g)
                out = torch.empty(Mx, N // 2 if kind == "geglu" else N, dtype=torch.float16, device=DEV)
                res = torch.randn(Mx, N, device=DEV, generator=g) if kind == "res32" else None
                if kind in ("o32", "rb_r32_o32", "r32_o32_o16_cs"):
                    if n % 256:
                        continue
                    o32 = torch.empty(Mx, N, dtype=torch.float32, device=DEV)
                    res = torch.randn(Mx, N, device=DEV, generator=g) if kind != "o32" else None
                    rbv = torch.randn(samples, N, device=DEV, generator=g)
                    cs = torch.empty(Mx // 64, N, 2, dtype=torch.float32, device=DEV)
                    if kind != "r32_o32_o16_cs":
                        out = None
                kw = {"plain": dict(), "geglu": dict(bias=bias, flags=hip.EPI_GEGLU), "res32": dict(bias=bias, residual32=res),
                      "a2": dict(a2=a2, lda2=K // 2, k1=K // 2),
                      "o32": dict(bias=bias, out32=o32 if kind == "o32" else None, rows_per_sample=n),
                      "rb_r32_o32": dict(bias=bias, rowbias=rbv if kind == "rb_r32_o32" else None, rows_per_sample=n, residual32=res, out32=o32 if kind == "rb_r32_o32" else None, split_k=False),
                      "r32_o32_o16_cs": dict(bias=bias, residual32=res, out32=o32 if kind == "r32_o32_o16_cs" else None, colstats=cs if kind == "r32_o32_o16_cs" else None, rows_per_sample=n)}[kind]
                fl0 = kw.pop("flags", 0)
                t = {}
                for tag, fl in (("old", hip.TUNE_NO_BIG_TILE), ("big", hip.TUNE_BIG_TILE)):
                    t[tag] = timeit(lambda: hip.gemm(a, w, out, M=Mx, N=N, K=K, lda=a.shape[1], ldc=out.shape[1] if out is not None else 0, flags=fl | fl0, **kw))
                if kind == "a2":
                    acat = torch.cat([a, a2], 1)
                    lo = torch.empty(Mx, N, dtype=torch.float16, device=DEV)
                    tl = timeit(lambda: torch.mm(acat, w.t(), out=lo))
                else:
                    lo = torch.empty(Mx, N, dtype=torch.float16, device=DEV)
                    tl = timeit(lambda: torch.mm(a, w.t(), out=lo))
                fl = 2.0 * Mx * N * K
                tiles = ((Mx + 255) // 256) * (N // 320)
                print(f"{name:10s} {Mx:6d} {N:6d} {K:5d} {tiles:6d} | {t['old']:10.1f} {fl / t['old'] / 1e6:5.0f} | {t['big']:10.1f} {fl / t['big'] / 1e6:5.0f} | "
                      f"{tl:10.1f} {fl / tl / 1e6:5.0f}", flush=True)
                del a, a2, w, out, res, lo
                o32 = cs = rbv = None


if __name__ == "__main__":
    main()
