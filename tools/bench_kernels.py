#!/usr/bin/env python3
"""Per-kernel throughput on the MI355X for the shapes the F=8, 512x512 UNet step issues (A/B tuning aid).
Interleaved rounds in one process (guide rule 24); random data (rule 25)."""
import argparse
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3

DEV = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--what", default="conv,gemm,attn,norm")
    a = ap.parse_args()
    hip.load()
    N = 3 * a.frames
    dt = torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dt).to(DEV)
    variants = {"auto": 0, "gn8": hip.TUNE_GN8, "persist": hip.TUNE_PERSISTENT, "db128": 0x500, "db160": 0x600}
    cvariants = {"patch": hip.TUNE_PATCH, "patch_bn160": hip.TUNE_PATCH | hip.TUNE_PATCH_BN160 | hip.TUNE_NO_Q8, "im2col": hip.TUNE_NO_PATCH}
    if "conv" in a.what:
        print("== conv3x3 implicit GEMM (TFLOP/s median | best), variants:", list(cvariants))
        for (H, cin, cout, stride) in [(64, 320, 320, 1), (64, 640, 320, 1), (64, 960, 320, 1), (32, 640, 640, 1),
                                       (32, 1280, 640, 1), (16, 1280, 1280, 1), (16, 2560, 1280, 1), (8, 1280, 1280, 1), (8, 2560, 1280, 1),
                                       (64, 320, 320, 2)]:
            x = rnd(N * H * H, cin)
            w = pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))).to(dt).to(DEV)
            OH = (H - 1) // stride + 1
            out = torch.empty(N * OH * OH, cout, dtype=dt, device=DEV)
            b = torch.zeros(cout, device=DEV)
            fl = 2.0 * N * OH * OH * cout * 9 * cin
            row = f"H{H:3d} {cin:4d}->{cout:4d} s{stride}: "
            res = {n: [] for n in cvariants}
            for rnd_ in range(3):      # interleaved rounds in one process (guide rule 24)
                for name, f in cvariants.items():
                    med, best = timeit(lambda: hip.conv3x3(x, w, out, nimg=N, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout,
                                                           stride=stride, bias=b, flags=f), iters=6, warm=2)
                    res[name].append(med)
            for name in cvariants:
                v = sorted(res[name])
                row += f"{name} {fl / v[1] / 1e9:6.0f}|{fl / v[0] / 1e9:6.0f}  "
            print(row, flush=True)
    if "gemm" in a.what:
        print("== plain GEMM")
        for (M, Nn, K, name0) in [(N * 4096, 960, 320, "qkv L0"), (N * 4096, 320, 320, "proj L0"), (N * 4096, 2560, 320, "ff1 L0"),
                                  (N * 4096, 320, 1280, "ff2 L0"), (N * 1024, 1920, 640, "qkv L1"), (N * 1024, 5120, 640, "ff1 L1"),
                                  (N * 1024, 640, 2560, "ff2 L1"), (N * 256, 3840, 1280, "qkv L2"), (N * 256, 10240, 1280, "ff1 L2"),
                                  (N * 256, 1280, 5120, "ff2 L2")]:
            x, w = rnd(M, K), rnd(Nn, K)
            geglu = name0.startswith("ff1")
            out = torch.empty(M, Nn // 2 if geglu else Nn, dtype=dt, device=DEV)
            bias = torch.zeros(Nn, device=DEV)
            fl = 2.0 * M * Nn * K
            row = f"{name0:8s} M{M:6d} N{Nn:5d} K{K:5d}: "
            res = {n: [] for n in variants}
            for rnd_ in range(3):
                for name, f in variants.items():
                    if geglu and name in ("db160", "db160nox", "pp160"):
                        continue
                    ff = f | (hip.EPI_GEGLU if geglu else 0)
                    med, best = timeit(lambda: hip.gemm(x, w, out, M=M, N=Nn, K=K, lda=K, ldc=out.shape[1], bias=bias, flags=ff), iters=6, warm=2)
                    res[name].append(med)
            for name in variants:
                if res[name]:
                    v = sorted(res[name])
                    row += f"{name} {fl / v[1] / 1e9:6.0f}|{fl / v[0] / 1e9:6.0f} ({v[1] * 1e3:.0f} us)  "
            print(row, flush=True)
    if "stream" in a.what:
        print("== plain GEMM with the fp32 residual stream (residual32 in, out32 and / or 16-bit out)")
        for (M, Nn, K, name0, want16, want32) in [(N * 4096, 320, 320, "out-proj L0", False, True), (N * 4096, 320, 1280, "ff2 L0", True, False),
                                                  (N * 4096, 320, 320, "proj_out L0", True, True), (N * 1024, 640, 640, "out-proj L1", False, True),
                                                  (N * 1024, 640, 2560, "ff2 L1", True, False), (N * 256, 1280, 1280, "out-proj L2", False, True)]:
            x, w = rnd(M, K), rnd(Nn, K)
            r32 = torch.randn(M, Nn, device=DEV)
            o16 = torch.empty(M, Nn, dtype=dt, device=DEV) if want16 else None
            o32 = torch.empty(M, Nn, device=DEV) if want32 else None
            bias = torch.zeros(Nn, device=DEV)
            fl = 2.0 * M * Nn * K
            byt = M * K * 2 + M * Nn * (4 + (2 if want16 else 0) + (4 if want32 else 0))
            row = f"{name0:12s} M{M:6d} N{Nn:5d} K{K:5d}: "
            vs = {"auto": 0}
            res = {n: [] for n in vs}
            for rnd_ in range(4):
                for name, f in vs.items():
                    med, best = timeit(lambda: hip.gemm(x, w, o16, M=M, N=Nn, K=K, lda=K, ldc=Nn, bias=bias, flags=f, residual32=r32, out32=o32), iters=6, warm=2)
                    res[name].append(med)
            for name in vs:
                v = sorted(res[name])
                row += f"{name} {v[1] * 1e3:6.1f} us ({byt / v[1] / 1e9:5.2f} TB/s, {fl / v[1] / 1e9:4.0f} TF)  "
            print(row, flush=True)
    if "attn" in a.what:
        print("== attention")
        for (n, dh) in [(4096, 40), (1024, 80), (256, 160), (64, 160)]:
            d = 8 * dh
            qkv = rnd(N, n, 3 * d)
            out = torch.empty(N, n, d, dtype=dt, device=DEV)
            fl = 4.0 * N * 8 * n * n * dh
            row = f"n{n:5d} dh{dh:4d}: "
            for var in (0, 1, 2):
                med, best = timeit(lambda: hip.attention(qkv, qkv[:, :, d:], qkv[:, :, 2 * d:], out, B=N, heads=8, n=n, nk=n, dh=dh,
                                                         ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d,
                                                         ldo=d, bso=n * d, scale=dh ** -0.5, variant=var))
                row += f"v{var} {fl / med / 1e9:6.0f}|{fl / best / 1e9:6.0f} ({med * 1e3:.0f} us)  "
            if dh == 40:
                med, best = timeit(lambda: hip.attention(qkv, qkv[:, :, d:], qkv[:, :, 2 * d:], out, B=N // 3, heads=8, n=n, nk=n, dh=dh,
                                                         ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d,
                                                         ldo=d, bso=n * d, scale=dh ** -0.5, v_sets=3, set_stride=N // 3))
                row += f"shared3 ({med * 1e3:.0f} us)"
            print(row, flush=True)
    if "norm" in a.what:
        print("== GroupNorm / LayerNorm (GB/s of algorithmic bytes)")
        for (hw, C) in [(4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1920), (256, 1280), (256, 2560), (64, 1280)]:
            x = rnd(N * hw, C)
            y = torch.empty_like(x)
            gm, bt = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
            st = hip.groupnorm_stats(x, nimg=N, hw=hw, C_=C, ldx=C)
            m1, _ = timeit(lambda: hip.groupnorm_stats(x, nimg=N, hw=hw, C_=C, ldx=C))
            m2, _ = timeit(lambda: hip.groupnorm_apply(x, st, gm, bt, y, nimg=N, hw=hw, C_=C, ldx=C, ldy=C, silu=True))
            by = N * hw * C * 2
            print(f"GN hw{hw:5d} C{C:5d}: stats {by / m1 / 1e6:6.0f} GB/s ({m1 * 1e3:.0f} us)  apply {2 * by / m2 / 1e6:6.0f} GB/s ({m2 * 1e3:.0f} us)", flush=True)
        for (M, C) in [(N * 4096, 320), (N * 1024, 640), (N * 256, 1280)]:
            x = rnd(M, C); y = torch.empty_like(x)
            gm, bt = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
            m, _ = timeit(lambda: hip.layernorm(x, gm, bt, y, M=M, C_=C, ldx=C, ldy=C))
            print(f"LN M{M:6d} C{C:5d}: {2 * M * C * 2 / m / 1e6:6.0f} GB/s ({m * 1e3:.0f} us)", flush=True)


if __name__ == "__main__":
    main()
