#!/usr/bin/env python3
"""The FeedForward third of a level-0 transformer block (norm3 -> ff.net[0] GEGLU -> ff.net[2] -> + x) as ONE launch
(csrc/ffn.hip) against the three-kernel path, interleaved rounds in one process.  usage: python tools/bench_ffn.py [M] [C]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip, packing
DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
C = int(sys.argv[2]) if len(sys.argv) > 2 else 320
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
x = r(M, C, sc=1.5)
gamma, beta = 1 + r(C, sc=0.2), r(C, sc=0.2)
w1, b1 = r(8 * C, C, sc=C ** -0.5).half(), r(8 * C, sc=0.3)
w2, b2 = r(C, 4 * C, sc=(4 * C) ** -0.5).half(), r(C, sc=0.3)
w1p, b1p = packing.pack_geglu(w1.cpu(), b1.cpu())
w1p, b1p, w2p = w1p.to(DEV), b1p.to(DEV), packing.pack_ffn_w2(w2.cpu()).to(DEV)
o16 = torch.empty(M, C, dtype=torch.float16, device=DEV)
ln = torch.empty(M, C, dtype=torch.float16, device=DEV)
ff = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)


def fused():
    hip.ffn_fused(x, gamma, beta, w1p, b1p, w2p, b2, o16, M=M, C_=C)


def three():
    hip.layernorm(x, gamma, beta, ln, M=M, C_=C, ldx=C, ldy=C)
    hip.gemm(ln, w1p, ff, M=M, N=8 * C, K=C, lda=C, ldc=4 * C, bias=b1p, flags=hip.EPI_GEGLU)
    hip.gemm(ff, w2, o16, M=M, N=C, K=4 * C, lda=4 * C, ldc=C, bias=b2, residual32=x, rows_per_sample=4096)


def time_us(fn, iters=6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    fn(); ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3


res = {"fused": [], "three-kernel": []}
for _ in range(7):
    res["fused"].append(time_us(fused))
    res["three-kernel"].append(time_us(three))
fl = 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C
for k, v in res.items():
    m = statistics.median(v)
    print(f"{k:14s} M={M} C={C}: median {m:8.1f} us  min {min(v):8.1f}  ({fl / m / 1e6:7.1f} TFLOP/s of the two GEMMs)")
print(f"fused / three-kernel = {statistics.median(res['fused']) / statistics.median(res['three-kernel']):.3f}")
