#!/usr/bin/env python3
"""A few launches of the level-0 fused tail (vface_attn_out_ffn_proj_fused, csrc/ffn.hip PRE + POST form) at 24 samples, for tools/pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip, packing
hip.load()
dev = "cuda"
C, n, N = 320, 4096, 24
M = N * n
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
w_o, w1, w2, w_po = r(C, C, sc=C ** -0.5), r(8 * C, C, sc=C ** -0.5), r(C, 4 * C, sc=(4 * C) ** -0.5), r(C, C, sc=C ** -0.5)
ffw, ffb = packing.pack_geglu(w1, r(8 * C, sc=0.1))
tail = torch.cat([packing.pack_attn_out_ffn(w_o, ffw), w_po[:, packing.ffn_w2_perm(C)]], 0).half().to(dev).contiguous()
w2p = packing.pack_ffn_w2(w2).half().to(dev)
att = r(M, C).half().to(dev)
t0, x_in = r(M, C).to(dev), r(M, C).to(dev)
a2 = r(N, C, sc=0.1).to(dev)
vec = lambda: r(C, sc=0.1).to(dev)
bo, gm, bt, b2, bpo = vec(), (1 + vec()), vec(), vec(), vec()
b1 = ffb.float().to(dev)
out32 = torch.empty(M, C, device=dev)
cs = torch.empty(M // 64, C, 2, device=dev)
for _ in range(4):
    hip.attn_out_ffn_proj_fused(att, t0, a2, tail, bo, gm, bt, b1, w2p, b2, bpo, x_in, None, out32, cs, M=M, C_=C, rows_per_sample=n)
torch.cuda.synchronize()
print("done")
