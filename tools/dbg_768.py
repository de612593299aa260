import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_unet_gpu as T
from vface_amd.utils import synth
for h in (96, 64, 48):
    flow_all = synth.synth_flow(2, h, h)
    for fusion in ("fft", "replace", "flow_fix"):
        whole = T._full_run(fusion, 3, 0, 3, h, flow_all)
        pair = T._full_run(fusion, 3, 1, 2, h, flow_all)
        d1 = (pair[:, 1] - whole[:, 2]).abs().max().item(); d0 = (pair[:, 0] - whole[:, 1]).abs().max().item()
        print(f"h={h} {fusion:9s} frame2 diff {d1:.3e}   frame1 diff {d0:.3e}", flush=True)
