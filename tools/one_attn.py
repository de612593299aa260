#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
hip.load()
N, n, dh = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 40
d = 8 * dh
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(N, n, 3 * d, generator=g) * 0.5).half().cuda()
out = torch.empty(N, n, d, dtype=torch.float16, device="cuda")
for _ in range(4):
    hip.attention(qkv, qkv[:, :, d:], qkv[:, :, 2 * d:], out, B=N, heads=8, n=n, nk=n, dh=dh, ldq=3 * d, ldk=3 * d, ldv=3 * d,
                  bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=dh ** -0.5)
torch.cuda.synchronize()
print("done")
