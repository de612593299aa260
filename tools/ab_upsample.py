#!/usr/bin/env python3
"""A/B: conv3x3 after nearest x2 upsample -- fused-upsample implicit GEMM (9 taps) vs four parity-phase 2x2 convs."""
import sys, os, math, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3, pack_upsample_phases
DEV = "cuda:0"
N_ = 24
g = torch.Generator().manual_seed(0)
def timeit(fn, iters=10):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    fn(); ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3
for (H, c) in [(32, 640), (16, 1280), (8, 1280)]:
    x = torch.randn(N_, H, H, c, generator=g).half().to(DEV)
    w = torch.randn(c, c, 3, 3, generator=g) / math.sqrt(9 * c)
    b = torch.randn(c, generator=g).to(DEV)
    w9, w4 = pack_conv3x3(w).half().to(DEV), pack_upsample_phases(w).half().to(DEV)
    out = torch.empty(N_, 2 * H, 2 * H, c, dtype=torch.float16, device=DEV)
    cs = torch.zeros(N_ * 4 * H * H // 64, c, 2, device=DEV)
    res = {"direct": [], "phases": []}
    for _ in range(4):
        res["direct"].append(timeit(lambda: hip.conv3x3(x, w9, out, nimg=N_, H=H, W=H, cin=c, cout=c, ldx=c, ldy=c, upsample=True, bias=b, colstats=cs)))
        res["phases"].append(timeit(lambda: hip.upsample2x_conv3x3(x, w4, out, nimg=N_, H=H, W=H, cin=c, cout=c, ldx=c, ldy=c, bias=b)))
    print(f"H{H}->{2 * H} C{c}: direct {statistics.median(res['direct']):7.1f} us   phases {statistics.median(res['phases']):7.1f} us", flush=True)
