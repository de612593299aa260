#!/usr/bin/env python3
"""Count instructions per basic block of one kernel in a hipcc -S listing (which blocks carry the exps / MFMAs)."""
import collections, re, sys
path, start = sys.argv[1], int(sys.argv[2])
lines = open(path).read().split("\n")[start:]
end = next(i for i, l in enumerate(lines) if "s_endpgm" in l)
lines = lines[:end]
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in lines:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        cur = m.group(1); blocks[cur] = []; continue
    if not t or t.startswith((";", ".", "//")):
        continue
    blocks[cur].append(t.split()[0])
for lab, ins in blocks.items():
    c = collections.Counter(ins)
    if len(ins) < int(sys.argv[3]) if len(sys.argv) > 3 else len(ins) < 40:
        continue
    cat = collections.Counter()
    for k, v in c.items():
        if k.startswith("v_mfma"): cat["MFMA"] += v
        elif k.startswith("s_"): cat["SALU"] += v
        else: cat[k] += v
    print(lab, len(ins), dict(sorted(cat.items(), key=lambda kv: -kv[1])[:24]))
