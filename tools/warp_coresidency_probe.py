#!/usr/bin/env python3
"""Does the flow warp give the same bits while another stream keeps the GPU busy?  (HISTORY R5: the cause of the round-1 "zero-row
misread" and of the round-4 flow_fix two-stream mismatch.)

Stream A runs a VICTIM on fixed inputs into a destination pre-filled with a sentinel and compares it, 16-byte chunk by chunk (one
lane's store), with its own output computed alone; stream B runs an AGGRESSOR at the same time.
  victims     `product`  vface_flow_warp as shipped (neighbour indices clamped with min, weights without selects)
              `selects`  the pre-round-5 form (tools/probe/probe_kernels.hip: boolean validity, `vx ? .. : 0` selects on the VCC lane mask)
  aggressors  none | torch matmul | this library's GEMM / convolution | its attention kernel at dh 160 / 80 / 40 / 32 and the
              shared-score form | LDS-DMA micro-kernels (full EXEC, partial EXEC, out-of-range lanes)
Build the probe library first (here, it travels with the snapshot):
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared tools/probe/probe_kernels.hip -o tools/probe/libprobe.so
usage (GPU box): python tools/warp_coresidency_probe.py [--rounds 100]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vface_amd import hip  # noqa: E402
from vface_amd.utils import synth  # noqa: E402

dev = torch.device("cuda", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=100)
    a_ = ap.parse_args()
    hip.load()
    probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libprobe.so"))
    vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
    probe.launch_dma_partial.restype = i32
    probe.launch_dma_partial.argtypes = [vp, ctypes.c_uint, i32, ctypes.c_ulonglong, ctypes.c_ulonglong, i32, vp, vp]
    probe.launch_warp_select_form.restype = i32
    probe.launch_warp_select_form.argtypes = [vp, i64, i64, vp, vp, i64, i64, i32, i32, i32, i32, f32, f32, vp]
    F_, h, w, d = 4, 64, 64, 320
    n, C = h * w, 2 * d
    g = torch.Generator(device=dev).manual_seed(0)
    src = torch.randn(F_ * n, C, device=dev, generator=g).half()
    flow = synth.synth_flow(F_ - 1, h, w).to(dev)
    dst = torch.empty(F_ * n, 3 * d, dtype=torch.float16, device=dev)
    SENT = 7.0
    oma = float(torch.tensor(1.0 - 0.8, dtype=torch.float32))

    def victim(kind):
        dst.fill_(SENT)
        if kind == "product":
            hip.flow_warp(src, dst[:, :C], flow, F=F_, h=h, w=w, C_=C, ld_src=C, fs_src=n * C, ld_dst=3 * d, fs_dst=n * 3 * d, alpha=0.8)
        else:
            rc = probe.launch_warp_select_form(src.data_ptr(), C, n * C, flow.data_ptr(), dst.data_ptr(), 3 * d, n * 3 * d, F_, h, w, C, 0.8, oma,
                                               torch.cuda.current_stream().cuda_stream)
            assert rc == 0
    a = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    qkv = {dh: torch.randn(24 * nn, 24 * dh, device=dev, generator=g).half() for dh, nn in ((40, 4096), (32, 4096), (80, 1024), (160, 256))}
    att = {dh: torch.empty(24 * nn, 8 * dh, dtype=torch.float16, device=dev) for dh, nn in ((40, 4096), (32, 4096), (80, 1024), (160, 256))}
    ga, gw = torch.randn(24576, 640, device=dev, generator=g).half(), torch.randn(5120, 640, device=dev, generator=g).half()
    go = torch.empty(24576, 5120, dtype=torch.float16, device=dev)
    cx = torch.randn(24 * 4096, 320, device=dev, generator=g).half()
    cw = (torch.randn(320, 9 * 320, device=dev, generator=g) * 0.02).half()
    cy = torch.empty(24 * 4096, 320, dtype=torch.float16, device=dev)
    dsrc = torch.randn(1 << 24, device=dev, dtype=torch.float16)
    sink = torch.zeros(4096, device=dev)

    def attention(dh, nn, reps=1, **kw):
        q, o, D = qkv[dh], att[dh], 8 * dh
        for _ in range(reps):
            hip.attention(q, q[:, D:], q[:, 2 * D:], o, B=kw.pop("B", 24), heads=8, n=nn, nk=nn, dh=dh, ldq=3 * D, ldk=3 * D, ldv=3 * D, bsq=nn * 3 * D,
                          bsk=nn * 3 * D, bsv=nn * 3 * D, ldo=D, bso=nn * D, scale=dh ** -0.5, **kw)

    def dma(mask, oob=0):
        assert probe.launch_dma_partial(dsrc.data_ptr(), dsrc.numel() * 2, 2000, mask, oob, 2048, sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    aggressors = {
        "none": None,
        "torch matmul": lambda: torch.mm(a, a),
        "gemm (LDS-DMA, MFMA 16x16x32)": lambda: hip.gemm(ga, gw, go, M=24576, N=5120, K=640, lda=640, ldc=5120),
        "conv3x3 (LDS-DMA with out-of-range halo lanes)": lambda: hip.conv3x3(cx, cw, cy, nimg=24, H=64, W=64, cin=320, cout=320, ldx=320, ldy=320),
        "LDS-DMA micro-kernel, full EXEC": lambda: dma(0xFFFFFFFFFFFFFFFF),
        "LDS-DMA micro-kernel, 5 of 8 lanes active": lambda: dma(0x1F1F1F1F1F1F1F1F),
        "LDS-DMA micro-kernel, 3 of 8 offsets out of range": lambda: dma(0xFFFFFFFFFFFFFFFF, 0xE0E0E0E0E0E0E0E0),
        "attention dh=160": lambda: attention(160, 256, reps=10),
        "attention dh=80": lambda: attention(80, 1024, reps=6),
        "attention dh=40": lambda: attention(40, 4096),
        "attention dh=40, eight waves per workgroup": lambda: attention(40, 4096, variant=8),
        "attention dh=40, shared scores (3 value sets)": lambda: attention(40, 4096, B=8, v_sets=3, set_stride=8),
        "attention dh=32": lambda: attention(32, 4096),
    }
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    for vk in ("selects", "product"):
        with torch.cuda.stream(sA):
            victim(vk)
        sA.synchronize()
        ref = dst.clone()
        print(f"victim: flow warp, `{vk}` form", flush=True)
        for name, busy in aggressors.items():
            bad_rounds = bad_chunks = sent_chunks = 0
            quarters = {}
            torch.cuda.synchronize()
            for it in range(a_.rounds):
                if busy is not None:
                    with torch.cuda.stream(sB):
                        for _ in range(3):
                            busy()
                with torch.cuda.stream(sA):
                    victim(vk)
                sA.synchronize()
                neq = (dst[:, :C] != ref[:, :C]).reshape(F_ * n, C // 8, 8).any(-1)
                if bool(neq.any()):
                    bad_rounds += 1
                    idx = torch.nonzero(neq)
                    bad_chunks += idx.shape[0]
                    sent_chunks += int((dst[:, :C].reshape(F_ * n, C // 8, 8)[neq] == SENT).all(-1).sum())
                    # lane of the thread that owns (pixel, chunk): i = pixel * 80 + chunk, lane = i % 64
                    for q4 in ((((idx[:, 0] % n) * (C // 8) + idx[:, 1]) % 64) // 16).tolist():
                        quarters[q4] = quarters.get(q4, 0) + 1
            torch.cuda.synchronize()
            print(f"   beside {name:52s}: {bad_rounds:3d}/{a_.rounds} rounds wrong; {bad_chunks} wrong 16-B chunks ({sent_chunks} never stored); "
                  f"owning lane's quarter-wave: {dict(sorted(quarters.items()))}", flush=True)


if __name__ == "__main__":
    main()
