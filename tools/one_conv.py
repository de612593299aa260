#!/usr/bin/env python3
"""Run one conv / gemm shape a few times (target for rocprofv3 --pmc passes)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3
hip.load()
N, H, cin, cout = 24, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 640, int(sys.argv[2]) if len(sys.argv) > 2 else 320
flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
g = torch.Generator().manual_seed(0)
x = (torch.randn(N * H * H, cin, generator=g) * 0.5).half().cuda()
w = pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().cuda()
out = torch.empty(N * H * H, cout, dtype=torch.float16, device="cuda")
b = torch.zeros(cout, device="cuda")
for _ in range(5):
    hip.conv3x3(x, w, out, nimg=N, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b, flags=flags)
torch.cuda.synchronize()
print("done")
