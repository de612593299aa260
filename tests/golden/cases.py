"""Input cases shared by the golden generator and the tests (no reference code, no reference access)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from vface_amd.utils import synth  # noqa: E402


def make_flows(h=64, w=64):
    """Flow fields ``[2,h,w]`` fp32 for the warp fixtures: zero, sub-pixel, +-3 px, integer shifts (floor
    sensitivity of the normalise/un-normalise round trip), out-of-border, and the bench's smooth field."""
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    return {
        "zero": np.zeros((2, h, w), np.float32),
        "subpixel": np.stack([0.37 * np.ones((h, w)), -0.61 * np.ones((h, w))]).astype(np.float32),
        "pm3": np.stack([3.0 * np.sin(ys / 5.0), -3.0 * np.cos(xs / 7.0)]).astype(np.float32),
        "integer": np.stack([np.round(2 * np.sin(xs)), np.round(2 * np.cos(ys))]).astype(np.float32),
        "oob": np.stack([xs - 80.0 + 0.25, 90.0 - ys]).astype(np.float32),
        "smooth": synth.synth_flow(1, h, w, seed=7)[0].numpy(),
    }
