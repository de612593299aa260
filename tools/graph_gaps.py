#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of `bench.py` (graph-replayed timed region): busy fraction of the GPU inside the replayed
steps = sum of kernel durations / (last end - first start), per step, and the distribution of the gaps between kernels."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# a step starts at pack_input_kernel
starts = [i for i, e in enumerate(ev) if "pack_input_kernel" in e[2]]
out = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = ev[a:b]
    busy = sum(e[1] - e[0] for e in seg)
    span = seg[-1][1] - seg[0][0]
    gaps = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
    out.append((len(seg), busy / 1e6, span / 1e6, sum(gaps) / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3, max(gaps) / 1e3))
for i, o in enumerate(out):
    print(f"step {i:2d}: {o[0]:4d} kernels, busy {o[1]:6.2f} ms of {o[2]:6.2f} ms span ({100 * o[1] / o[2]:5.1f} %), gaps {o[3]:5.2f} ms (median {o[4]:4.1f} us, max {o[5]:6.1f} us)")
