#!/usr/bin/env python3
"""Diagnostic: where the convolution kernels spend their cycles (s_memtime stamps, flag 0x4000; never on the product path).
`patch`: conv.hip (per wave: prologue / K loop / epilogue, and inside the loop: wait+barrier / LDS-DMA issue (+ the late
waves' half tile) / fragment reads + MFMAs); `im2col`: gemm.hip's implicit GEMM."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3
hip.load()
N = 24
for cin, cout, Hh in [(320, 320, 64), (960, 320, 64), (640, 640, 32), (1280, 1280, 16)]:
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(N * Hh * Hh, cin, generator=g) * 0.5).half().cuda()
    w = pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().cuda()
    out = torch.empty(N * Hh * Hh, cout, dtype=torch.float16, device="cuda")
    b = torch.zeros(cout, device="cuda")
    nt = 9 * cin // 64
    for stag, fl in (("same", 0), ("split", 0x2000)):
        nblk = N * (Hh // 16) ** 2 * (cout // 160)
        dbg = torch.zeros(nblk * 64 // (2 * cout) + 2, cout, 2, dtype=torch.float32, device="cuda")   # flat float scratch
        for _ in range(3):
            hip.conv3x3(x, w, out, nimg=N, H=Hh, W=Hh, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b,
                        flags=hip.TUNE_PATCH | 0x4000 | fl, colstats=dbg)
        torch.cuda.synchronize()
        d = dbg.flatten()[: nblk * 64].reshape(nblk, 8, 8).cpu()
        for name, sel in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
            m = d[:, sel].mean(dim=(0, 1))
            print(f"patch {stag:8s} conv {cin}->{cout} @{Hh} ({nt} K tiles, {nblk} workgroups) {name}: prologue {m[0]:7.0f}  K loop "
                  f"{m[1]:8.0f} ({m[1] / nt:5.0f}/tile: DMA wait {m[3] / nt:5.0f}, barrier wait {m[4] / nt:5.0f}; "
                  f"MFMA floor 640/wave, 1280/SIMD)  epilogue {m[2]:7.0f} cycles")
