#!/usr/bin/env python3
"""Does an under-filled last round of the one-workgroup-per-CU patch-kernel grid cost what its CU count says?  Times one
convolution shape at several image counts (= workgroup counts) in interleaved rounds."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3
from tools.bench_kernels import timeit
hip.load()
g = torch.Generator().manual_seed(0)
for (H, cin, cout) in [(16, 1280, 1280), (32, 640, 640)]:
    w = pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().cuda()
    b = torch.zeros(cout, device="cuda")
    counts = [16, 24, 32, 48, 64] if H == 16 else [8, 12, 16, 24, 32]
    bufs = {n: ((torch.randn(n * H * H, cin, generator=g) * 0.5).half().cuda(), torch.empty(n * H * H, cout, dtype=torch.float16, device="cuda")) for n in counts}
    res = {(n, bn): [] for n in counts for bn in (160, 128)}
    for _ in range(3):
        for n in counts:
            for bn in (160, 128):
                x, o = bufs[n]
                fl_ = hip.TUNE_PATCH | (hip.TUNE_PATCH_BN160 if bn == 160 else 0)
                med, best = timeit(lambda: hip.conv3x3(x, w, o, nimg=n, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b, flags=fl_), iters=6, warm=2)
                res[(n, bn)].append(med)
    for n in counts:
        row = f"H{H} {cin}->{cout} images {n:3d}: "
        for bn in (160, 128):
            wgs = n * (H // 16) ** 2 * (cout // bn)
            fl = 2.0 * n * H * H * cout * 9 * cin
            v = sorted(res[(n, bn)])[1]
            row += f" BN{bn}: {wgs:4d} workgroups = {wgs / 256:.2f} rounds {v * 1e3:7.1f} us {fl / v / 1e9:6.0f} TFLOP/s |"
        print(row, flush=True)
