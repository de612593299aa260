#!/usr/bin/env python3
"""A/B: attention dh=40 n=4096 schedule variants (QT = 4 vs 2 queries tiles per wave), interleaved rounds."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
DEV = "cuda:0"
N, n, dh = 24, 4096, 40
d = 8 * dh
g = torch.Generator(device=DEV).manual_seed(0)
qkv = torch.randn(N, n, 3 * d, device=DEV, generator=g).half()
out = torch.empty(N, n, d, dtype=torch.float16, device=DEV)
def run(var):
    hip.attention(qkv, qkv[:, :, d:], qkv[:, :, 2 * d:], out, B=N, heads=8, n=n, nk=n, dh=dh, ldq=3 * d, ldk=3 * d, ldv=3 * d,
                  bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=dh ** -0.5, variant=var)
def timeit(var, iters=8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    run(var); ev[0].record()
    for i in range(iters):
        run(var); ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3
res = {0: [], 1: []}
for _ in range(6):
    for v in (0, 1):
        res[v].append(timeit(v))
for v in (0, 1):
    print(f"variant {v} (QT={'4' if v == 0 else '2'}): median {statistics.median(res[v]):7.1f} us  all {[round(x) for x in res[v]]}")
