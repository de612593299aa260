#!/usr/bin/env python3
"""What the fp32 residual + carrier epilogue costs a 3x3 convolution on the patch-staged kernel (one workgroup per CU) and on the im2col kernel
(two per CU): the same launch with a 16-bit epilogue, with the fp32 residual read + carrier write, and with the 16-bit copy as well.
usage (GPU box): python tools/epilogue_probe.py"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3
DEV = "cuda:0"
def time_us(fn, iters=10):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    fn(); fn(); ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3
for (nimg, H, cin, cout) in [(24, 64, 320, 320), (24, 32, 640, 640), (24, 16, 1280, 1280), (48, 64, 320, 320)]:
    M = nimg * H * H
    x = torch.randn(M, cin, device=DEV).half()
    w = pack_conv3x3(torch.randn(cout, cin, 3, 3) * 0.02).half().to(DEV)
    b = torch.randn(cout, device=DEV)
    o16 = torch.empty(M, cout, dtype=torch.float16, device=DEV)
    o32 = torch.empty(M, cout, dtype=torch.float32, device=DEV)
    r32 = torch.randn(M, cout, device=DEV)
    cs = torch.empty(M // 64, cout, 2, dtype=torch.float32, device=DEV)
    for name, fl in (("patch", 0), ("im2col", hip.TUNE_NO_PATCH)):
        kw = dict(nimg=nimg, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b, colstats=cs, flags=fl)
        t0 = time_us(lambda: hip.conv3x3(x, w, o16, **kw))
        t1 = time_us(lambda: hip.conv3x3(x, w, None, residual32=r32, out32=o32, **{**kw, "ldy": 0}))
        t2 = time_us(lambda: hip.conv3x3(x, w, o16, residual32=r32, out32=o32, **kw))
        fl_ = 2.0 * M * cout * 9 * cin
        print(f"{nimg}x{H}x{H} {cin}->{cout} {name:7s}: 16-bit out {t0:7.1f} us ({fl_/t0/1e6:6.0f} TF)   fp32 residual+carrier {t1:7.1f} us (+{t1-t0:5.1f})   + 16-bit copy {t2:7.1f} us (+{t2-t0:5.1f})", flush=True)
