#!/usr/bin/env python3
"""A/B of the speculative softmax reference of the attention kernel (variant bit 4 = checked loop only) at the UNet's shapes:
interleaved launches in one process, min over reps; outputs compared (rel-L2; both are exact softmaxes with different references)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vface_amd import hip  # noqa: E402

DEV = "cuda:0"
hip.load()


def run(N, n, dh, heads=8, reps=12, scale_in=1.0, **kw):
    d = heads * dh
    g = torch.Generator(device=DEV).manual_seed(0)
    qkv = (torch.randn(N, n, 3 * d, device=DEV, generator=g) * scale_in).half()
    outs, best = {}, {0: 1e9, 16: 1e9}
    B = kw.pop("B", N)
    for var in (0, 16):
        outs[var] = torch.zeros(N, n, d, dtype=torch.float16, device=DEV)
    args = dict(B=B, heads=heads, n=n, nk=n, dh=dh, ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d,
                ldo=d, bso=n * d, scale=dh ** -0.5, **kw)
    for rep in range(reps):
        for var in (0, 16):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.attention(qkv, qkv[:, :, d:], qkv[:, :, 2 * d:], outs[var], variant=var, **args)
            e1.record()
            torch.cuda.synchronize()
            if rep >= 2:
                best[var] = min(best[var], e0.elapsed_time(e1) * 1e3)
    a, b = outs[0].float(), outs[16].float()
    rel = float((a - b).norm() / b.norm())
    return best[0], best[16], rel


for name, N, n, dh, kw in (("level 0, plain (24 samples, n 4096, dh 40)", 24, 4096, 40, {}),
                           ("level 0, shared scores (8 frames x 3 sets)", 24, 4096, 40, dict(B=8, v_sets=3, set_stride=8)),
                           ("level 1 (n 1024, dh 80)", 24, 1024, 80, {}), ("level 2 (n 256, dh 160)", 24, 256, 160, {}),
                           ("level 0, plain, logits x4 (sharper maps)", 24, 4096, 40, dict(_scale=2.0))):
    sc = kw.pop("_scale", 1.0)
    s, c, rel = run(N, n, dh, scale_in=sc, **kw)
    print(f"{name:48s} speculative {s:7.1f} us   checked {c:7.1f} us   x{c / s:.3f}   rel-L2 between them {rel:.2e}")
