#!/usr/bin/env python3
"""Diagnostic: per-phase cycle stamps of the attention kernel (variant bit 8 = stamp build path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip

DEV = "cuda:0"
N, n, dh = 24, 4096, 40
d = 8 * dh
g = torch.Generator(device=DEV).manual_seed(0)
qkv = torch.randn(N, n, 3 * d, device=DEV, generator=g).half()
out = torch.zeros(N, n, d, dtype=torch.float16, device=DEV)
names = ["load issue", "S = K Q^T", "softmax", "P V", "store LDS", "barrier"]
for label, var, kw, B in [("QT4", 0, {}, N), ("QT2", 1, {}, N), ("shared3 QT2", 0, dict(v_sets=3, set_stride=N // 3), N // 3)]:
    for _ in range(3):
        hip.attention(qkv, qkv[:, :, d:], qkv[:, :, 2 * d:], out, B=B, heads=8, n=n, nk=n, dh=dh, ldq=3 * d, ldk=3 * d,
                      ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=dh ** -0.5,
                      variant=var | 0x100, **kw)
    torch.cuda.synchronize()
    qt = 2 if (var == 1 or kw) else 4
    nwg = (n // (64 * qt)) * 8 * B
    st = out.view(-1).view(torch.float32)[: nwg * 4 * 8].view(nwg * 4, 8).double().cpu()
    per_block = st[:, :6].mean(0) / (n // 64)
    print(f"{label:12s} cycles per key block per wave: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, per_block.tolist()))
          + f"  total {per_block.sum():6.0f}")
