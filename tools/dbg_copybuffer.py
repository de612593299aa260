#!/usr/bin/env python3
"""Which host call emits the `__amd_rocclr_copyBuffer` dispatches seen in the step profile?  (run under rocprofv3 --kernel-trace)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3
hip.load()
which = sys.argv[1]
N, H, cin, cout = 2, 32, 64, 160
g = torch.Generator().manual_seed(0)
x = (torch.randn(N * H * H, cin, generator=g) * 0.5).half().cuda()
w = pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().cuda()
out = torch.empty(N * H * H, cout, dtype=torch.float16, device="cuda")
b = torch.zeros(cout, device="cuda")
wl = torch.randn(cout, cin).half().cuda()
torch.cuda.synchronize()
for _ in range(10):
    if which == "patch":
        hip.conv3x3(x, w, out, nimg=N, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b, flags=hip.TUNE_PATCH)
    elif which == "im2col":
        hip.conv3x3(x, w, out, nimg=N, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b, flags=hip.TUNE_NO_PATCH)
    elif which == "gemm":
        hip.gemm(x, wl, out, M=x.shape[0], N=cout, K=cin, lda=cin, ldc=cout, bias=b)
    elif which == "events":
        e = torch.cuda.Event(enable_timing=True); e.record()
    elif which == "full":
        t = torch.full((8,), 5, device="cuda", dtype=torch.long); t2 = torch.cat([t] * 3)
torch.cuda.synchronize()
print("done", which)
