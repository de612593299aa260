#!/usr/bin/env python3
"""The fused SpatialTransformer front (csrc/stfront.hip) against the launches it replaces, at the level-0 shape of the F-frame
batch: gn_finalize + gn_apply + GEMM(proj_in) + layernorm + GEMM(projection).  usage: python tools/bench_front.py [--frames 8]"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from vface_amd import hip as h
    from vface_amd.packing import pack_st_front
    h.load()
    dev, dt = "cuda", torch.float16
    C, hw = 320, 4096
    N = 3 * a.frames
    M = N * hw
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, C, generator=g).to(dev)
    w_in = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dt)
    w_p = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).to(dt)
    b_in = torch.randn(C, generator=g).to(dev) * 0.1
    one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    sl = x.reshape(M // 64, 64, C)
    cs = torch.stack([sl.sum(1), (sl * sl).sum(1)], -1).contiguous()
    wcat = pack_st_front(w_in.float(), w_p.float()).to(dt).to(dev)
    w_in_d, w_p_d = w_in.to(dev), w_p.to(dev)
    t0 = torch.empty(M, C, dtype=torch.float32, device=dev)
    qkv = torch.empty(M, 3 * C, dtype=dt, device=dev)
    ln = torch.empty(M, C, dtype=dt, device=dev)
    g16 = torch.empty(M, C, dtype=dt, device=dev)
    Fn = a.frames * hw

    def fused(rows_full, nq_lo, want_ln):
        ab = h.groupnorm_coeffs_from_cols(cs, one, zero, nimg=N, hw=hw, C_=C, eps=1e-6)
        h.st_front(x, ab, wcat, b_in, one, zero, t0, qkv, M=M, C_=C, hw=hw, NQ=3 * C, rows_full=rows_full, nq_lo=nq_lo,
                   ln=ln if want_ln else None)

    def chain(rows_full, nq_lo, want_ln):
        st = h.groupnorm_stats_from_cols(cs, nimg=N, hw=hw, C_=C, eps=1e-6)
        h.groupnorm_apply(x, st, one, zero, g16, nimg=N, hw=hw, C_=C, ldx=C, ldy=C, silu=False)
        h.gemm(g16, w_in_d, None, M=M, N=C, K=C, lda=C, ldc=0, bias=b_in, out32=t0, rows_per_sample=hw)
        h.layernorm(t0, one, zero, ln, M=M, C_=C, ldx=C, ldy=C)
        h.gemm(ln, w_p_d, qkv, M=rows_full, N=3 * C, K=C, lda=C, ldc=3 * C, split_k=False)
        if rows_full < M:
            h.gemm(ln[rows_full:], w_p_d[nq_lo:], qkv[rows_full:, nq_lo:], M=M - rows_full, N=3 * C - nq_lo, K=C, lda=C, ldc=3 * C,
                   split_k=False)

    def timeit(fn, *args):
        for _ in range(3):
            fn(*args)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(*args); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        return best

    for name, rf, lo, wl in (("unhooked (every column, all rows)", M, 0, False), ("replace (chunk 0: q,k,v; chunks 1,2: v)", Fn, 2 * C, False),
                             ("linear fusions (+ LayerNorm output)", Fn, 2 * C, True)):
        tf, tc = timeit(fused, rf, lo, wl), timeit(chain, rf, lo, wl)
        fl = 2.0 * M * C * C + 2.0 * C * (rf * 3 * C + (M - rf) * (3 * C - lo))
        by = M * C * 8 + (rf * 3 * C + (M - rf) * (3 * C - lo)) * 2 + (M * C * 2 if wl else 0)
        print(f"{name:48s} fused {tf:7.1f} us ({fl / tf / 1e6:5.0f} TFLOP/s, {by / tf / 1e3:5.0f} GB/s)   separate launches {tc:7.1f} us   x{tc / tf:.2f}")


if __name__ == "__main__":
    main()
