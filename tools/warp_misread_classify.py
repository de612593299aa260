#!/usr/bin/env python3
"""What ARE the wrong values the SELECT form of the flow warp (tools/probe/probe_kernels.hip) writes (lanes 48-63) while the dh = 40 attention kernel runs on another stream?  Re-computes
the warp's terms in torch (fp32) for the wrong 16-byte chunks of one bad round and tests candidates: a tap missing (load returned
zeros), the un-warped row, the blend of a neighbouring pixel's coordinates, a neighbouring chunk's data.
usage (GPU box): python tools/warp_misread_classify.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vface_amd import hip  # noqa: E402
from vface_amd.utils import synth  # noqa: E402

dev = torch.device("cuda", 0)


def main():
    hip.load()
    F_, h, w, d = 4, 64, 64, 320
    n, C = h * w, 2 * d
    g = torch.Generator(device=dev).manual_seed(0)
    src = torch.randn(F_ * n, C, device=dev, generator=g).half()
    flow = synth.synth_flow(F_ - 1, h, w).to(dev)
    dst = torch.empty(F_ * n, 3 * d, dtype=torch.float16, device=dev)

    import ctypes
    probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libprobe.so"))
    vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
    probe.launch_warp_select_form.restype = i32
    probe.launch_warp_select_form.argtypes = [vp, i64, i64, vp, vp, i64, i64, i32, i32, i32, i32, f32, f32, vp]
    oma_ = float(torch.tensor(1.0 - 0.8, dtype=torch.float32))

    def warp():      # the pre-round-5 SELECT form (tools/probe/probe_kernels.hip): the shipped kernel no longer shows the effect
        dst.fill_(7.0)
        assert probe.launch_warp_select_form(src.data_ptr(), C, n * C, flow.data_ptr(), dst.data_ptr(), 3 * d, n * 3 * d, F_, h, w, C, 0.8, oma_,
                                             torch.cuda.current_stream().cuda_stream) == 0
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sA):
        warp()
    sA.synchronize()
    ref = dst[:, :C].clone()
    q = torch.randn(24 * 4096, 960, device=dev, generator=g).half()
    att = torch.empty(24 * 4096, 320, dtype=torch.float16, device=dev)
    # torch model of the terms (fp32, the kernel's operation order)
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")

    def unnorm(pos, dlt, size):
        v = pos + dlt
        qn = (2.0 * v) / float(size - 1)
        c = ((qn - 1.0) + 1.0) * 0.5 * float(size - 1)
        return c.clamp(0.0, float(size - 1))
    terms = {}
    for f in range(1, F_):
        ix, iy = unnorm(xs, flow[f - 1, 0], w).flatten(), unnorm(ys, flow[f - 1, 1], h).flatten()
        x0, y0 = ix.floor(), iy.floor()
        wx1, wy1 = ix - x0, iy - y0
        x0, y0 = x0.long(), y0.long()
        vx, vy = x0 + 1 <= w - 1, y0 + 1 <= h - 1
        x1, y1 = torch.where(vx, x0 + 1, x0), torch.where(vy, y0 + 1, y0)
        fr = src[(f - 1) * n:f * n].float()
        w00, w01 = (1 - wx1) * (1 - wy1), torch.where(vx, wx1 * (1 - wy1), torch.zeros_like(wx1))
        w10, w11 = torch.where(vy, (1 - wx1) * wy1, torch.zeros_like(wx1)), torch.where(vx & vy, wx1 * wy1, torch.zeros_like(wx1))
        A, B = fr[y0 * w + x0] * w00[:, None], fr[y0 * w + x1] * w01[:, None]
        Cc, D = fr[y1 * w + x0] * w10[:, None], fr[y1 * w + x1] * w11[:, None]
        cur = (0.8 * src[f * n:(f + 1) * n].float()).half().float()
        terms[f] = (cur, A, B, Cc, D)
    oma = float(torch.tensor(1.0 - 0.8, dtype=torch.float32))
    model = torch.cat([src[:n].float()] + [terms[f][0] + oma * (((terms[f][1] + terms[f][2]) + terms[f][3]) + terms[f][4]) for f in range(1, F_)]).half()
    print("torch model == kernel alone:", bool(torch.equal(model, ref)), "max abs", float((model.float() - ref.float()).abs().max()))
    found = None
    for it in range(200):
        with torch.cuda.stream(sB):
            for _ in range(3):
                hip.attention(q, q[:, 320:], q[:, 640:], att, B=24, heads=8, n=4096, nk=4096, dh=40, ldq=960, ldk=960, ldv=960, bsq=4096 * 960,
                              bsk=4096 * 960, bsv=4096 * 960, ldo=320, bso=4096 * 320, scale=40 ** -0.5)
        with torch.cuda.stream(sA):
            warp()
        sA.synchronize()
        got = dst[:, :C].clone()
        neq = (got != ref).reshape(F_ * n, C // 8, 8).any(-1)
        if bool(neq.any()):
            found = (got, neq)
            break
    torch.cuda.synchronize()
    if found is None:
        print("no bad round in 200")
        return
    got, neq = found
    idx = torch.nonzero(neq)
    print(f"bad round: {idx.shape[0]} wrong chunks; frames {sorted(set((idx[:, 0] // n).tolist()))}")
    tally = {}
    shown = 0
    for r, ch in idx[:4000].tolist():
        f, pix = r // n, r % n
        sl = slice(ch * 8, ch * 8 + 8)
        gv = got[r, sl].float()
        if f == 0:
            kind = "frame 0 (copy path)"
        else:
            cur, A, B, Cc, D = (t[pix, sl] for t in terms[f])
            cands = {"all four taps (correct)": ((A + B) + Cc) + D, "no taps (alpha*cur only)": torch.zeros_like(A), "A": A, "A+B": A + B, "A+B+C": (A + B) + Cc,
                     "B+C+D": (B + Cc) + D, "A+C+D": (A + Cc) + D, "A+B+D": (A + B) + D, "C+D": Cc + D, "B": B, "C": Cc, "D": D, "A+C": A + Cc, "A+D": A + D, "B+D": B + D, "B+C": B + Cc}
            kind = "none of the candidates"
            for name, tsum in cands.items():
                if bool(torch.equal((cur + oma * tsum).half().float(), gv)):
                    kind = name
                    break
            if kind == "none of the candidates":
                # the same pixel's OTHER chunks (a lane reading a neighbour lane's data), or the unblended source row
                if bool(torch.equal(src[r, sl].float(), gv)):
                    kind = "unblended src row"
                else:
                    full = (cur * 0 + 0)
                    for dc in (-16, -8, -4, -2, -1, 1, 2, 4, 8, 16):
                        c2 = ch + dc
                        if 0 <= c2 < C // 8 and bool(torch.equal(ref[r, c2 * 8:c2 * 8 + 8].float(), gv)):
                            kind = f"the correct value of chunk {dc:+d}"
                            break
                    for dp in (-65, -64, -63, -1, 1, 63, 64, 65):
                        p2 = pix + dp
                        if kind == "none of the candidates" and 0 <= p2 < n and bool(torch.equal(ref[f * n + p2, sl].float(), gv)):
                            kind = f"the correct value of pixel {dp:+d}"
            if kind == "none of the candidates" and shown < 6:
                shown += 1
                print(f"   unexplained: frame {f} pixel {pix} chunk {ch}: got {gv.tolist()} want {ref[r, sl].float().tolist()} cur {cur.tolist()} taps {(((A + B) + Cc) + D).tolist()}")
        tally[kind] = tally.get(kind, 0) + 1
    for k, v in sorted(tally.items(), key=lambda kv: -kv[1]):
        print(f"   {v:6d} wrong chunks = alpha*cur + (1-alpha)*[{k}]" if not k.startswith(("frame", "unbl", "the ", "none")) else f"   {v:6d} wrong chunks: {k}")


if __name__ == "__main__":
    main()
