#!/usr/bin/env python3
"""Attention kernel forms side by side on the UNet's levels (vface_attention's `variant` bits: 1 = more query tiles per wave, 8 = eight
waves per workgroup), HIP events, medians.  usage (GPU box): python tools/bench_attn_variants.py > gpurun_out/attn_variants.txt"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vface_amd import hip  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for _ in range(3):
        fn()
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) * 1e3


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    for B in (24, 48, 96):
        for n, d in ((1024, 640), (256, 1280), (4096, 320)):
            heads, dh = 8, d // 8
            qkv = torch.randn(B * n, 3 * d, device=DEV, generator=g).half()
            ref = None
            for variant in (0, 1, 8, 9):
                if dh == 160 and variant:
                    continue
                out = torch.empty(B * n, d, dtype=torch.float16, device=DEV)
                kw = dict(B=B, heads=heads, n=n, nk=n, dh=dh, ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d,
                          ldo=d, bso=n * d, scale=float(np.float32(1.0) / np.sqrt(np.float32(dh))), variant=variant)
                t = timeit(lambda: hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], out, **kw))
                if ref is None:
                    ref = out.clone()
                same = bool(torch.equal(ref, out))
                fl = 4.0 * n * n * dh * heads * B
                print(f"B {B:3d} n {n:5d} dh {dh:4d} variant {variant}: {t:8.1f} us {fl / t / 1e6:6.0f} TFLOP/s  bits == variant 0: {same}", flush=True)


if __name__ == "__main__":
    main()
