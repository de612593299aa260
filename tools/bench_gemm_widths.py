#!/usr/bin/env python3
"""Round 6: the big GEMM tile 320 and 256 channels wide (csrc/gemm_big.hip, NJ = 10 | 8) on the launches of the 640- / 1280-channel
transformer blocks whose N allows both, at 48 samples (one launch stream's half of the headline batch) and 96 (the whole batch): forced
widths and the library's own rule, with a bit-equality check between the widths.
usage (GPU box): python tools/bench_gemm_widths.py"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vface_amd import hip  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=12):
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
    hip.load()
    g = torch.Generator(device=DEV).manual_seed(0)
    print(f"{'launch':12s} {'M':>6s} {'N':>6s} {'K':>5s} | {'tiles 320':>9s} {'us':>7s} {'TF':>5s} | {'tiles 256':>9s} {'us':>7s} {'TF':>5s} | {'rule us':>7s} bits")
    for samples in (48, 96):
        for name, tok, N, K, kind in (("ff2 L2", 256, 1280, 5120, "res16"), ("qkv L2", 256, 3840, 1280, "plain"), ("proj L2", 256, 1280, 1280, "plain"),
                                      ("ff1 L2", 256, 10240, 1280, "geglu"), ("fsai L2", 256, 2560, 2560, "a2"), ("ff1 L1", 1024, 5120, 640, "geglu"),
                                      ("ff2 mid", 64, 1280, 5120, "res16")):
            M = samples * tok // (3 if kind == "a2" else 1)
            a = torch.randn(M, K if kind != "a2" else K // 2, device=DEV, generator=g).half()
            a2 = torch.randn(M, K // 2, device=DEV, generator=g).half() if kind == "a2" else None
            w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).half()
            bias = torch.randn(N, device=DEV, generator=g)
            res = torch.randn(M, N, device=DEV, generator=g).half() if kind == "res16" else None
            kw = {"plain": dict(), "geglu": dict(bias=bias, flags=hip.EPI_GEGLU), "res16": dict(bias=bias, residual=res, ldr=N),
                  "a2": dict(a2=a2, lda2=K // 2, k1=K // 2)}[kind]
            base = kw.pop("flags", 0)
            outs, us = {}, {}
            for tag, fl in (("320", hip.TUNE_BIG_TILE | hip.TUNE_BIG_W320), ("256", hip.TUNE_BIG_TILE | hip.TUNE_BIG_W256), ("rule", 0)):
                o = torch.empty(M, N // 2 if kind == "geglu" else N, dtype=torch.float16, device=DEV)
                call = lambda o=o, fl=fl: hip.gemm(a, w, o, M=M, N=N, K=K, lda=a.stride(0), ldc=o.stride(0), flags=base | fl, split_k=False, **kw)
                us[tag] = timeit(call)
                outs[tag] = o
            fl = 2.0 * M * N * K
            mt = (M + 255) // 256
            same = torch.equal(outs["320"], outs["256"]) and torch.equal(outs["320"], outs["rule"])
            print(f"{name:12s} {M:6d} {N:6d} {K:5d} | {mt * (N // 320):9d} {us['320']:7.1f} {fl / us['320'] / 1e6:5.0f} | {mt * (N // 256):9d} {us['256']:7.1f} "
                  f"{fl / us['256'] / 1e6:5.0f} | {us['rule']:7.1f} {'equal' if same else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    main()
