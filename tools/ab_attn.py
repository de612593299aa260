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
res = {0: [], 1: [], 2: []}
for _ in range(6):
    for v in (0, 1, 2):
        res[v].append(timeit(v))
for v in (0, 1, 2):
    print(f"variant {v} (QT={'2' if v == 1 else '4'}{' exact-scale' if v == 2 else ' lazy'}): median {statistics.median(res[v]):7.1f} us  all {[round(x) for x in res[v]]}")

ref = None
def ref_attn():
    sp = lambda t: t.reshape(2, n, 8, dh).permute(0, 2, 1, 3).double()
    a = torch.softmax(sp(qkv[:2, :, :d]) @ sp(qkv[:2, :, d:2 * d]).transpose(-1, -2) * dh ** -0.5, -1) @ sp(qkv[:2, :, 2 * d:])
    return a.permute(0, 2, 1, 3).reshape(2, n, d)
r = ref_attn()
for v in (0, 2):
    run(v); e = ((out[:2].double() - r).norm() / r.norm()).item()
    print(f"variant {v}: rel-L2 vs fp64 reference {e:.3e}")
