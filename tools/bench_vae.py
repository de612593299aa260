#!/usr/bin/env python3
"""First-stage KL-VAE throughput: encode and decode of F 512x512 frames, synthetic weights."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd.ldm.models.autoencoder import FFHQ_VAE_CONFIG, AutoencoderKL
from vface_amd.utils import synth

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=8); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
DEV = "cuda:0"
m = AutoencoderKL(**FFHQ_VAE_CONFIG); synth.fill_module_(m, seed=0, prefix="vae."); m = m.to(DEV)
x = torch.stack([synth.synth_normal(f"bv.x.{f}", (3, 512, 512)).clamp(-1, 1) for f in range(a.frames)]).to(DEV)
z = m.encode(x).mode(); m.decode(z); torch.cuda.synchronize()
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / a.iters * 1e3
te, td = t(lambda: m.encode(x).mode()), t(lambda: m.decode(z))
# algorithmic FLOPs per frame at 512x512 (conv + 1x1 + attention), counted from the layer table
def flops(enc):
    ch, mult, nrb = 128, (1, 2, 4, 4), 2
    f, res = 0, 512
    conv = lambda cin, cout, r, k=3: 2.0 * r * r * cout * cin * k * k
    if enc:
        f += conv(3, ch, res); cin = ch
        for l, mu in enumerate(mult):
            for b in range(nrb):
                co = ch * mu; f += conv(cin, co, res) + conv(co, co, res) + (conv(cin, co, res, 1) if cin != co else 0); cin = co
            if l != 3: res //= 2; f += conv(cin, cin, res)
        mid = 2 * (conv(cin, cin, res) * 2) + 4 * conv(cin, cin, res, 1) + 4.0 * (res * res) ** 2 * cin
        return f + mid + conv(cin, 8, res)
    res = 64; cin = 512; f += conv(4, cin, res)
    f += 2 * (conv(cin, cin, res) * 2) + 4 * conv(cin, cin, res, 1) + 4.0 * (res * res) ** 2 * cin
    for l in (3, 2, 1, 0):
        for b in range(nrb + 1):
            co = ch * mult[l]; f += conv(cin, co, res) + conv(co, co, res) + (conv(cin, co, res, 1) if cin != co else 0); cin = co
        if l != 0: res *= 2; f += conv(cin, cin, res)
    return f + conv(cin, 3, res)
fe, fd = flops(True) * a.frames, flops(False) * a.frames
print(f"VAE F={a.frames} 512x512: encode {te:.2f} ms ({fe / te / 1e9:.0f} TFLOP/s, {fe / a.frames / 1e9:.0f} GFLOP/frame)   "
      f"decode {td:.2f} ms ({fd / td / 1e9:.0f} TFLOP/s, {fd / a.frames / 1e9:.0f} GFLOP/frame)")
