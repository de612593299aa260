#!/usr/bin/env python3
"""Diagnostic (never on the product path): ablations of the patch-staged convolution's K loop in its DIAG build -- the same
launch with (a) no LDS-DMA / DMA wait inside the loop, (b) no workgroup barrier, (c) no fragment reads after the first K
tile, and combinations.  Results are garbage; the times say which part of a K-tile period the matrix pipe waits for."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3
hip.load()
N = 24
ABL = [("full (diag build)", 0), ("no DMA", 0x200000), ("no barrier", 0x400000), ("no LDS reads", 0x800000),
       ("no DMA, no barrier", 0x600000), ("no DMA, no reads", 0xA00000), ("no DMA, no barrier, no reads (MFMA only)", 0xE00000)]
for cin, cout, Hh in [(960, 320, 64), (1280, 1280, 16)]:
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(N * Hh * Hh, cin, generator=g) * 0.5).half().cuda()
    w = pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().cuda()
    out = torch.empty(N * Hh * Hh, cout, dtype=torch.float16, device="cuda")
    b = torch.zeros(cout, device="cuda")
    nblk = N * (Hh // 16) ** 2 * (cout // 160)
    dbg = torch.zeros(nblk * 64 // (2 * cout) + 2, cout, 2, dtype=torch.float32, device="cuda")
    fl = 2.0 * N * Hh * Hh * cout * 9 * cin
    def run(flags, cs):
        hip.conv3x3(x, w, out, nimg=N, H=Hh, W=Hh, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b, flags=flags, colstats=cs)
    def timeit(flags, cs):
        for _ in range(3): run(flags, cs)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(flags, cs); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return sorted(ts)[3]
    t = timeit(hip.TUNE_PATCH, None)
    print(f"conv {cin}->{cout} @{Hh}: product build {t * 1e3:7.1f} us = {fl / t / 1e9:6.0f} TFLOP/s")
    for name, f in ABL:
        t = timeit(hip.TUNE_PATCH | 0x4000 | f, dbg)
        print(f"    {name:44s} {t * 1e3:7.1f} us = {fl / t / 1e9:6.0f} TFLOP/s (as if)")
