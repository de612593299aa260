#!/usr/bin/env python3
"""The synthetic end-to-end pipeline with --drop_dead_branches --pipeline_inversion, three batches of 8 frames: stage seconds per batch (the
middle batch is the steady state: its sampling runs beside the next batch's inversion on two HIP streams).  Round 6: 1.41 s where the two
streams had landed on one hardware queue, 1.12 s with a pair verified by the engine's spin-kernel probe.
usage (GPU box): python tools/e2e_pipeline_stages.py"""
import sys, os, json
sys.path.insert(0, os.getcwd())
from vface_amd.scripts import VFace_inference_batch as cli
import contextlib
opt = cli.build_parser().parse_args(["--synthetic", "--with_vae", "--raft_flow", "--paste_back", "--skip_save", "--n_frames", "24", "--n_samples", "8", "--fusion", "flow_fix",
                                     "--ddim_steps", "50", "--Base_dir", "/tmp/vface_e2e_x", "--drop_dead_branches", "--pipeline_inversion"])
with contextlib.redirect_stdout(sys.stderr):
    res = cli.run_synthetic(opt)
for i, b in enumerate(res["batches"]):
    print(i, {k: round(v, 3) for k, v in b["stage_seconds"].items()}, round(b.get("batch_wall_seconds", 0), 3))
