#!/usr/bin/env python3
"""Timings of the widened rows at the sizes the 8-frame / 16-frame workloads use, HIP events on the launch
stream: the RAFT-shaped flow producer (F - 1 pairs of 512 x 512 frames, 20 updates) and the paste-back (F decoded 512 x 512
crops -> 1024 canvas -> 1024 x 1024 frames), with the per-step DDIM time of bench.py beside them for scale.
usage: python tools/bench_widening.py [--frames 8]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vface_amd.utils import synth

DEV = "cuda:0"


def time_ms(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return statistics.median(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--frame_size", type=int, default=1024)
    a = ap.parse_args()
    F_, R, S = a.frames, a.res, a.frame_size
    from vface_amd.raft import RAFT
    from vface_amd.scripts import temporal_flow as tflow
    from vface_amd.scripts.paste_back import PasteBack
    from vface_amd.scripts.VFace_inference_batch import _quad_coeffs
    raft = RAFT()
    synth.fill_module_(raft, seed=0, prefix="raft.")
    raft = raft.to(DEV).eval()
    video = torch.stack([synth.synth_normal(f"w.img{f}", (3, R, R)).clamp(-1, 1) for f in range(F_)]).to(DEV)
    t = time_ms(lambda: tflow.return_flow(video, raft), iters=3)
    print(f"return_flow: {F_ - 1} pairs of {R}x{R}, 20 updates: {t:8.1f} ms  ({t / (F_ - 1):6.1f} ms per pair)", flush=True)
    for it in (1, 12):
        tt = time_ms(lambda: raft(video[1:], video[:-1], num_flow_updates=it), iters=3)
        print(f"   the same with {it:2d} update(s): {tt:8.1f} ms", flush=True)
    dec = torch.stack([synth.synth_normal(f"w.dec{f}", (3, R, R)) * 0.6 for f in range(F_)]).to(DEV)
    frames = torch.randint(0, 256, (F_, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(DEV)
    q = S / 4.0
    co = np.stack([_quad_coeffs(1024, [(q + 3 * f, q), (3 * q, q + f), (3 * q - f, 3 * q), (q, 3 * q - 2 * f)]) for f in range(F_)])
    pb = PasteBack(H=R, W=R, device=DEV, encode_decode=None)
    t = time_ms(lambda: pb.paste(dec, frames, co))
    print(f"paste-back without the background round trip: {F_} frames {S}x{S}: {t:8.2f} ms  ({t / F_ * 1e3:6.0f} us per frame)", flush=True)
    from vface_amd.ldm.models.autoencoder import FFHQ_VAE_CONFIG, AutoencoderKL
    vae = AutoencoderKL(**FFHQ_VAE_CONFIG, compute_dtype=torch.float16)
    synth.fill_module_(vae, seed=0, prefix="vae.")
    vae = vae.to(DEV)
    rt = lambda x: vae.decode(vae.encode(x).mode(1.0))
    pb2 = PasteBack(H=R, W=R, device=DEV, encode_decode=rt)
    t2 = time_ms(lambda: pb2.paste(dec, frames, co), iters=3)
    print(f"paste-back with the VAE encode + decode of the background (:610-623): {t2:8.1f} ms  ({t2 / F_:6.1f} ms per frame)", flush=True)
    # the host route of the reference for scale: one frame through Pillow (resize, transform, composite), decoded crop already on the host
    from PIL import Image
    import time
    x = torch.clamp((dec[0] + 1.0) / 2.0, 0, 1).permute(1, 2, 0).cpu().numpy()
    fr = frames[0].cpu().numpy()
    t0 = time.time()
    for _ in range(3):
        img = Image.fromarray((255. * x).astype(np.uint8)).resize((1024, 1024), Image.BILINEAR)
        sw = img.convert("RGBA")
        sw.putalpha(255)
        bg = Image.fromarray(fr).convert("RGBA")
        bg.alpha_composite(sw.transform((S, S), Image.PERSPECTIVE, co[0], Image.BILINEAR))
    print(f"the same three Pillow calls on the host (no VAE, no PCIe): {(time.time() - t0) / 3 * 1e3:8.1f} ms per frame", flush=True)


if __name__ == "__main__":
    main()
