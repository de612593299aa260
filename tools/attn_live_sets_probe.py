#!/usr/bin/env python3
"""Round 6: the shared-score attention with TWO of its three value sets live (`v_sets=3, v_sets_live=2`: the batch came
without the recon third, or -- since round 6 -- it is the [A ; C] prefix batch of the shared uncond/cond block) must give the full call's bits
for the live sets.  The four-wave instantiation does; the eight-wave one (`attn_kernel<.., 40, 2, 3, .., 8, 2>`, variant bit 4 here) was
reported not to.  This prints WHERE the two differ: per (sample, head) and per query row / value column, for several batch sizes.
usage (GPU box): python tools/attn_live_sets_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vface_amd import hip  # noqa: E402

dev = torch.device("cuda", 0)


def main():
    hip.load()
    dh, heads, n = 40, 8, 4096
    D = heads * dh
    g = torch.Generator(device=dev).manual_seed(1)
    for F_ in (2, 8):
        qkv = torch.randn(3 * F_ * n, 3 * D, device=dev, generator=g).half()
        kw = dict(heads=heads, n=n, nk=n, dh=dh, ldq=3 * D, ldk=3 * D, ldv=3 * D, bsq=n * 3 * D, bsk=n * 3 * D, bsv=n * 3 * D, ldo=D, bso=n * D,
                  scale=dh ** -0.5, B=F_, v_sets=3, set_stride=F_)
        full = torch.full((3 * F_ * n, D), 7.0, dtype=torch.float16, device=dev)
        hip.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], full, **kw)
        outs = {}
        for name, variant in (("four waves, two live sets", 0), ("eight waves, two live sets", 16)):
            o = torch.full((3 * F_ * n, D), 7.0, dtype=torch.float16, device=dev)
            hip.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], o, v_sets_live=2, variant=variant, **kw)
            torch.cuda.synchronize()
            outs[name] = o
            live, dead = o[:2 * F_ * n], o[2 * F_ * n:]
            neq = live != full[:2 * F_ * n]
            print(f"F = {F_}: {name}: {int(neq.sum())} of {neq.numel()} live elements differ from the full call; dead set untouched: "
                  f"{bool((dead == 7.0).all())}", flush=True)
            if bool(neq.any()):
                v = neq.reshape(2, F_, n, heads, dh)
                print("   by set:", v.sum((1, 2, 3, 4)).tolist(), " by head:", v.sum((0, 1, 2, 4)).tolist())
                print("   by value column:", v.sum((0, 1, 2, 3)).tolist())
                rows = v.sum((0, 1, 3, 4))
                nz = torch.nonzero(rows).flatten()
                print(f"   query rows affected: {nz.numel()} of {n}; first {nz[:12].tolist()} last {nz[-4:].tolist()}; rows mod 512 histogram of 64-row "
                      f"groups: {torch.bincount((nz % 512) // 64, minlength=8).tolist()}")
                d = (live.float() - full[:2 * F_ * n].float()).abs()
                print(f"   max abs difference {float(d.max()):.3e} (values ~ {float(full.float().abs().mean()):.3e})")


if __name__ == "__main__":
    main()
