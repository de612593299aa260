#!/usr/bin/env python3
"""Diagnostic: prologue / K loop / epilogue cycles per wave of the plain GEMM (gemm.hip flag 0x4000) on the short-K level-0 shapes."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
hip.load()
for name, M, N, K, var in [("qkv L0", 98304, 960, 320, 6), ("proj L0", 98304, 320, 320, 6), ("ff2 L0", 98304, 320, 1280, 6),
                           ("qkv L1", 24576, 1920, 640, 6), ("ff2 L2", 6144, 1280, 5120, 5)]:
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    b = torch.zeros(N, device="cuda")
    bn = 160 if var == 6 else 128
    nblk = ((M + 127) // 128) * ((N + bn - 1) // bn)
    dbg = torch.zeros(nblk * 32 // (2 * N) + 2, N, 2, dtype=torch.float32, device="cuda")
    for abl, fl in (("16-bit out", 0),):   # (the diagnostic flag is not accepted together with the fp32 stream operands)
        kw = {}
        for _ in range(3):
            hip.gemm(x, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=b, flags=(var << 8) | 0x4000 | fl, colstats=dbg, split_k=False, **kw)
        torch.cuda.synchronize()
        d = dbg.flatten()[: nblk * 16].reshape(nblk, 4, 4).cpu()
        m = d.mean(dim=(0, 1))
        nt = K // 64
        bar = (d[..., 3] % 65536).mean()
        drain = (d[..., 3] // 65536).mean() * 16
        print(f"{name:8s} M{M} N{N} K{K} ({nt} K tiles, {nblk} tiles) [{abl:14s}]: prologue {m[0]:7.0f}  K loop {m[1]:7.0f} ({m[1] / nt:5.0f}/tile; "
              f"MFMA floor {16 * 4 * (bn // 32) * 2}/wave)  epilogue {m[2]:7.0f} = barrier wait {bar:6.0f} + body {m[2] - bar - drain:6.0f} + store drain {drain:6.0f}")
        e = dbg.flatten()[nblk * 16: nblk * 32].reshape(nblk, 4, 4).cpu().mean(dim=(0, 1))
        print(f"         epilogue body: setup (bias / residual loads, fold, wait) {e[0]:6.0f} | LDS writes {e[1]:6.0f} | LDS read wait {e[2]:6.0f} | convert + store {e[3]:6.0f}   (4 passes summed)")
