#!/usr/bin/env python3
"""A minimal victim beside the dh = 40 attention kernel: NL back-to-back 16-byte loads per lane from a table of known contents
(tools/probe/probe_kernels.hip victim_kernel) -- which load ordinal, which quarter-wave, zeros or not.
usage (GPU box): python tools/load_victim_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vface_amd import hip  # noqa: E402

dev = torch.device("cuda", 0)


def main():
    hip.load()
    probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libprobe.so"))
    probe.launch_victim.restype = ctypes.c_int
    probe.launch_victim.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    nelem = 1 << 20
    table = torch.arange(nelem, dtype=torch.int32, device=dev).repeat_interleave(4).contiguous()      # element i = {i, i, i, i}
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(24 * 4096, 960, device=dev, generator=g).half()
    att = torch.empty(24 * 4096, 320, dtype=torch.float16, device=dev)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    for busy in (False, True):
        for mubuf in (1, 0):
            for nl in (1, 2, 4, 8):
                err = torch.zeros(8 * 4 * 2, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                for it in range(30):
                    if busy:
                        with torch.cuda.stream(sB):
                            for _ in range(2):
                                hip.attention(q, q[:, 320:], q[:, 640:], att, B=24, heads=8, n=4096, nk=4096, dh=40, ldq=960, ldk=960, ldv=960,
                                              bsq=4096 * 960, bsk=4096 * 960, bsv=4096 * 960, ldo=320, bso=4096 * 320, scale=40 ** -0.5)
                    with torch.cuda.stream(sA):
                        rc = probe.launch_victim(table.data_ptr(), nelem, 64, nl, mubuf, 1024, err.data_ptr(), torch.cuda.current_stream().cuda_stream)
                        assert rc == 0
                    sA.synchronize()
                torch.cuda.synchronize()
                e = err.cpu().reshape(8, 4, 2)
                tot = int(e[..., 0].sum())
                desc = "; ".join(f"load {l} quarter {qq}: {int(e[l, qq, 0])} wrong ({int(e[l, qq, 1])} zeros)" for l in range(nl) for qq in range(4) if int(e[l, qq, 0]))
                print(f"attention beside: {busy!s:5s}  {'buffer_load' if mubuf else 'global_load'}_dwordx4 x {nl}: {tot} wrong loads of {30 * 1024 * 256 * 64 * nl}" + (f"  [{desc}]" if desc else ""), flush=True)


if __name__ == "__main__":
    main()
