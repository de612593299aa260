#!/usr/bin/env python3
"""Diagnostic: where a K tile of the implicit-GEMM conv spends its cycles (s_memtime stamps, flag 0x4000)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3
hip.load()
N, H = 24, 64
for cin, cout, var in [(640, 320, 6), (320, 320, 6), (1280, 640, 5)]:
    Hh = H if cout == 320 else 32
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(N * Hh * Hh, cin, generator=g) * 0.5).half().cuda()
    w = pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().cuda()
    out = torch.empty(N * Hh * Hh, cout, dtype=torch.float16, device="cuda")
    b = torch.zeros(cout, device="cuda")
    bn = 160 if var == 6 else 128
    nblk = (N * Hh * Hh // 128) * ((cout + bn - 1) // bn)
    dbg = torch.zeros(nblk * 4 * 4 // 2 + 64, 2, 2, dtype=torch.float32, device="cuda")  # reinterpret as flat floats
    for _ in range(3):
        hip.conv3x3(x, w, out, nimg=N, H=Hh, W=Hh, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b, flags=(var << 8) | 0x4000,
                    colstats=dbg)
    torch.cuda.synchronize()
    d = dbg.flatten()[: nblk * 16].reshape(nblk, 4, 4).cpu()
    nt = 9 * cin // 64
    m = d.mean(dim=(0, 1))
    print(f"conv {cin}->{cout} BN={bn} ({nt} K tiles, {nblk} workgroups): per wave: prologue {m[0]:8.0f}  K loop {m[1]:8.0f} "
          f"({m[1] / nt:5.0f}/tile, MFMA floor {16 * 4 * (bn // 32) * 2})  epilogue {m[2]:8.0f} cycles")
