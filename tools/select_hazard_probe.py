#!/usr/bin/env python3
"""Round 6: what in the OLD flow warp's select made it the victim of round 5's co-residency misread (HISTORY.md R5)?

The select that returned its zero branch in lanes 48..63 -- ``v_cndmask_b32_e32 v40, 0, v40, vcc`` -- reads, ONE issue slot after
it was written, the low half of a PACKED fp32 product (``v_pk_mul_f32 v[40:41], ..``); the two selects a few slots further on (a VOP2 on
VCC reading a plain ``v_mul_f32`` result, a VOP3 on an SGPR pair reading the high half of the same packed product seven slots later) were
right.  This probe runs three BUILDS of that reproducer (tools/probe/probe_kernels.hip, `warp_select_form_kernel`) beside the attention
kernel of another stream:

  base   as round 5 measured it (hipcc forms the two weights with one v_pk_mul_f32)
  nopk   the same source compiled without packed fp32 (-Xclang -target-feature -Xclang -packed-fp32-ops): the product is a v_mul_f32
  nop    -DPROBE_SELECT_NOP: the product pinned in a register and eight idle issue slots in front of the select (hipcc then also forms it
         with a plain v_mul_f32)

Build (here; the libraries travel with the snapshot):
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Ivface_amd/csrc -shared tools/probe/probe_kernels.hip"
  hipcc $F -o tools/probe/libprobe.so
  hipcc $F -Xclang -target-feature -Xclang -packed-fp32-ops -o tools/probe/libprobe_nopk.so
  hipcc $F -DPROBE_SELECT_NOP -o tools/probe/libprobe_nop.so
usage (GPU box): python tools/select_hazard_probe.py [--rounds 40]"""
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
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--diag", action="store_true", help="the -DPROBE_SELECT_DIAG build (libprobe_diag.so) beside the four-wave dh = 40 attention: "
                                                         "in the wrong chunks, is it the selected weight or the loaded tap that differs?")
    ap.add_argument("--war", action="store_true", help="pinned micro-victims of the second hypothesis: a vector write of a buffer_load's ADDRESS register "
                                                        "two slots after the load (probe_kernels.hip, vmem_war_victim_kernel)")
    ap.add_argument("--micro", action="store_true", help="the pinned producer / consumer micro-victims (probe_kernels.hip, pk_victim_kernel) "
                                                          "instead of the three builds of the warp")
    a_ = ap.parse_args()
    hip.load()
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")
    vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
    libs = {}
    for tag, name in (("base", "libprobe.so"), ("nopk", "libprobe_nopk.so"), ("nop", "libprobe_nop.so")) + ((("diag", "libprobe_diag.so"),) if a_.diag else ()):
        lib = ctypes.CDLL(os.path.join(here, name))
        lib.launch_warp_select_form.restype = i32
        lib.launch_warp_select_form.argtypes = [vp, i64, i64, vp, vp, i64, i64, i32, i32, i32, i32, f32, f32, vp]
        libs[tag] = lib
    F_, h, w, d = 4, 64, 64, 320
    n, C = h * w, 2 * d
    g = torch.Generator(device=dev).manual_seed(0)
    src = torch.randn(F_ * n, C, device=dev, generator=g).half()
    flow = synth.synth_flow(F_ - 1, h, w).to(dev)
    dst = torch.empty(F_ * n, 3 * d, dtype=torch.float16, device=dev)
    SENT = 7.0
    oma = float(torch.tensor(1.0 - 0.8, dtype=torch.float32))

    def victim(tag):
        dst.fill_(SENT)
        rc = libs[tag].launch_warp_select_form(src.data_ptr(), C, n * C, flow.data_ptr(), dst.data_ptr(), 3 * d, n * 3 * d, F_, h, w, C, 0.8, oma,
                                               torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    qkv = {dh: torch.randn(24 * nn, 24 * dh, device=dev, generator=g).half() for dh, nn in ((40, 4096), (80, 1024))}
    att = {dh: torch.empty(24 * nn, 8 * dh, dtype=torch.float16, device=dev) for dh, nn in ((40, 4096), (80, 1024))}

    def attention(dh, nn, reps=1, **kw):
        q, o, D = qkv[dh], att[dh], 8 * dh
        for _ in range(reps):
            hip.attention(q, q[:, D:], q[:, 2 * D:], o, B=24, heads=8, n=nn, nk=nn, dh=dh, ldq=3 * D, ldk=3 * D, ldv=3 * D, bsq=nn * 3 * D,
                          bsk=nn * 3 * D, bsv=nn * 3 * D, ldo=D, bso=nn * D, scale=dh ** -0.5, **kw)
    # (attention.hip's dispatch since the end of round 5: dh = 40 runs eight waves per workgroup by default, variant bit 3 = four;
    #  dh = 80 runs four by default, variant bit 3 = eight)
    aggressors = {
        "none": None,
        "attention dh=40, four waves per workgroup (round 5: 100/100)": lambda: attention(40, 4096, variant=8),
        "attention dh=40, eight waves per workgroup (round 5: 0/100)": lambda: attention(40, 4096),
        "attention dh=80, four waves per workgroup (round 5: 75/100)": lambda: attention(80, 1024, reps=6),
        "attention dh=80, eight waves per workgroup": lambda: attention(80, 1024, reps=6, variant=8),
    }
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    if a_.diag:
        with torch.cuda.stream(sA):
            victim("diag")
        sA.synchronize()
        ref = dst.clone()
        busy = aggressors["attention dh=40, four waves per workgroup (round 5: 100/100)"]
        tot = {"rounds": 0, "chunks": 0, "weight": 0, "tap": 0, "both": 0, "neither": 0}
        shown = 0
        for it in range(a_.rounds):
            with torch.cuda.stream(sB):
                for _ in range(3):
                    busy()
            with torch.cuda.stream(sA):
                victim("diag")
            sA.synchronize()
            neq = (dst[:, :C] != ref[:, :C]).reshape(F_ * n, C // 8, 8).any(-1)
            if not bool(neq.any()):
                continue
            tot["rounds"] += 1
            dg, dr = dst[:, C:C + C // 2].contiguous().view(torch.int32).reshape(F_ * n, C // 8, 2), \
                ref[:, C:C + C // 2].contiguous().view(torch.int32).reshape(F_ * n, C // 8, 2)
            wdiff, tdiff = (dg[..., 0] != dr[..., 0])[neq], (dg[..., 1] != dr[..., 1])[neq]
            tot["chunks"] += int(neq.sum()); tot["weight"] += int((wdiff & ~tdiff).sum()); tot["tap"] += int((tdiff & ~wdiff).sum())
            tot["both"] += int((wdiff & tdiff).sum()); tot["neither"] += int((~wdiff & ~tdiff).sum())
            if shown < 6:
                idx = torch.nonzero(neq)[:3]
                for r_, c_ in idx.tolist():
                    print(f"   row {r_} chunk {c_}: weight got {dg[r_, c_, 0].view(torch.float32).item():.6f} want {dr[r_, c_, 0].view(torch.float32).item():.6f}; "
                          f"tap dword got {dg[r_, c_, 1].item() & 0xFFFFFFFF:08x} want {dr[r_, c_, 1].item() & 0xFFFFFFFF:08x}", flush=True)
                    shown += 1
        print(f"diag build beside four-wave dh = 40 attention, {a_.rounds} rounds: {tot}", flush=True)
        return
    if a_.war:
        lib = libs["base"]
        lib.launch_vmem_war_victim.restype = i32
        lib.launch_vmem_war_victim.argtypes = [i32, vp, ctypes.c_uint, i32, i32, vp, vp]
        table = torch.randint(1, 2 ** 31 - 1, (1 << 22,), dtype=torch.int32, device=dev)
        for mode, text in ((0, "buffer_load offen -> v_pk_mul_f32 -> vector write of the load's address register (the failing build's order)"),
                           (1, "buffer_load offen -> s_nop 7 -> the same write"),
                           (2, "four loads in flight -> v_pk_mul_f32 -> address write -> select (the failing build's window)")):
            print(f"VMEM address WAR micro-victim {mode}: {text}", flush=True)
            for name, busy in aggressors.items():
                err = torch.zeros(8, dtype=torch.int32, device=dev)
                bad_rounds = 0
                torch.cuda.synchronize()
                for it in range(a_.rounds):
                    before = int(err[0])
                    if busy is not None:
                        with torch.cuda.stream(sB):
                            for _ in range(3):
                                busy()
                    with torch.cuda.stream(sA):
                        assert lib.launch_vmem_war_victim(mode, table.data_ptr(), table.numel(), 100, 2048, err.data_ptr(), sA.cuda_stream) == 0
                    sA.synchronize()
                    bad_rounds += int(err[0]) != before
                torch.cuda.synchronize()
                e = err.tolist()
                print(f"   beside {name:62s}: {bad_rounds:3d}/{a_.rounds} rounds wrong; {e[0]} wrong loads of {a_.rounds * 2048 * 256 * 100:.2e}; "
                      f"by quarter-wave {e[1:5]}", flush=True)
        return
    if a_.micro:
        lib = libs["base"]
        lib.launch_pk_victim.restype = i32
        lib.launch_pk_victim.argtypes = [i32, vp, ctypes.c_uint, i32, i32, vp, vp]
        xin = torch.rand(1 << 20, device=dev) + 0.5
        modes = {0: "v_pk_mul_f32 -> 1 filler -> v_cndmask_b32_e32 (vcc)   [the old warp's pair]", 4: "v_pk_mul_f32 -> 0 filler -> v_cndmask_b32_e32 (vcc)",
                 5: "v_pk_mul_f32 -> s_nop 7  -> v_cndmask_b32_e32 (vcc)", 1: "v_pk_mul_f32 -> 1 filler -> v_cndmask_b32_e64 (SGPR pair)",
                 3: "v_mul_f32    -> 1 filler -> v_cndmask_b32_e32 (vcc)", 2: "v_pk_mul_f32 -> 1 filler -> v_add_f32_e32",
                 8: "v_pk_mul_f32 -> 1 filler -> v_add_f32_e64", 6: "v_pk_mul_f32 -> 1 filler -> v_mul_f32_e32", 7: "v_pk_mul_f32 -> 1 filler -> v_mov_b32_e32"}
        for mode, text in modes.items():
            print(f"micro-victim {mode}: {text}", flush=True)
            for name, busy in aggressors.items():
                err = torch.zeros(8, dtype=torch.int32, device=dev)
                bad_rounds = 0
                torch.cuda.synchronize()
                for it in range(a_.rounds):
                    before = int(err[0])
                    if busy is not None:
                        with torch.cuda.stream(sB):
                            for _ in range(3):
                                busy()
                    with torch.cuda.stream(sA):
                        # 2048 workgroups x 256 lanes x 200 sequences: ~as long as one attention launch
                        assert lib.launch_pk_victim(mode, xin.data_ptr(), xin.numel(), 200, 2048, err.data_ptr(), sA.cuda_stream) == 0
                    sA.synchronize()
                    bad_rounds += int(err[0]) != before
                torch.cuda.synchronize()
                e = err.tolist()
                print(f"   beside {name:62s}: {bad_rounds:3d}/{a_.rounds} rounds wrong; {e[0]} wrong results of {a_.rounds * 2048 * 256 * 200:.2e}; "
                      f"by quarter-wave {e[1:5]}", flush=True)
        return
    with torch.cuda.stream(sA):
        victim("base")
    sA.synchronize()
    ref0 = dst.clone()
    for tag in libs:
        with torch.cuda.stream(sA):
            victim(tag)
        sA.synchronize()
        ref = dst.clone()          # every build is compared with ITS OWN output computed alone (the builds differ in fp32 contraction)
        with torch.cuda.stream(sA):
            victim(tag)
        sA.synchronize()
        assert torch.equal(dst, ref), "a build alone must reproduce itself"
        print(f"victim build `{tag}`: alone it {'==' if torch.equal(ref[:, :C], ref0[:, :C]) else '!='} the base build alone "
              f"({int((ref[:, :C] != ref0[:, :C]).sum())} elements differ)", flush=True)
        for name, busy in aggressors.items():
            bad_rounds = bad_chunks = 0
            quarters = {}
            torch.cuda.synchronize()
            for it in range(a_.rounds):
                if busy is not None:
                    with torch.cuda.stream(sB):
                        for _ in range(3):
                            busy()
                with torch.cuda.stream(sA):
                    victim(tag)
                sA.synchronize()
                neq = (dst[:, :C] != ref[:, :C]).reshape(F_ * n, C // 8, 8).any(-1)
                if bool(neq.any()):
                    bad_rounds += 1
                    idx = torch.nonzero(neq)
                    bad_chunks += idx.shape[0]
                    for q4 in ((((idx[:, 0] % n) * (C // 8) + idx[:, 1]) % 64) // 16).tolist():
                        quarters[q4] = quarters.get(q4, 0) + 1
            torch.cuda.synchronize()
            print(f"   beside {name:62s}: {bad_rounds:3d}/{a_.rounds} rounds wrong; {bad_chunks} wrong 16-B chunks; owning lane's quarter-wave: "
                  f"{dict(sorted(quarters.items()))}", flush=True)


if __name__ == "__main__":
    main()
