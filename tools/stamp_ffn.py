#!/usr/bin/env python3
"""In-kernel phase stamps of the fused FeedForward kernel (DIAGNOSTIC build of csrc/ffn.hip only: a library built with the
stamp edits of this tool's docstring).  Prints median cycles per wave: prologue | main loop | epilogue, the GEGLU
share of the loop, and the in-kernel clock (s_memtime / s_memrealtime)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip, packing
DEV = "cuda:0"
M, C = 98304, 320
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=g) * sc
x = r(M, C, sc=1.5)
gamma, beta = 1 + r(C, sc=0.2), r(C, sc=0.2)
w1, b1 = r(8 * C, C, sc=C ** -0.5).half(), r(8 * C, sc=0.3)
w2, b2 = r(C, 4 * C, sc=(4 * C) ** -0.5).half(), r(C, sc=0.3)
w1p, b1p = packing.pack_geglu(w1.cpu(), b1.cpu())
w1p, b1p, w2p = w1p.to(DEV), b1p.to(DEV), packing.pack_ffn_w2(w2.cpu()).to(DEV)
o16 = torch.empty(M, C, dtype=torch.float16, device=DEV)
nw = (M // 128) * 4
buf = torch.zeros(M * C + nw * 8, device=DEV)
o32 = buf[:M * C].view(M, C)
for _ in range(30):
    hip.ffn_fused(x, gamma, beta, w1p, b1p, w2p, b2, o16, M=M, C_=C, out32=o32)
torch.cuda.synchronize()
d = buf[M * C:].view(nw, 8).cpu()
med = d.median(0).values
print(f"cycles per wave (median over {nw} waves): prologue {med[0]:.0f} | main loop {med[1]:.0f} (of which: vmcnt waits {med[3]:.0f}, barrier waits {med[6]:.0f}) | epilogue {med[2]:.0f} | total {med[5]:.0f}")
print(f"in-kernel clock: {(d[:, 5] / d[:, 4]).median().item() * 100:.0f} MHz;  per 64 hidden units: {med[1] / 20:.0f} cycles for 120 MFMAs (3840 matrix cycles)")
