#!/usr/bin/env python3
"""Instruction histogram of the largest loop (longest backward branch) of one kernel's assembly (scripts/kasm.sh -> k.s): loop_hist.py k.s [unroll]"""
import re, sys
from collections import Counter
L = open(sys.argv[1]).read().split('\n'); unroll = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
labels = {m.group(1): i for i, l in enumerate(L) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
best = None
for i, l in enumerate(L):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and (best is None or i - labels[m.group(1)] > best[1] - best[0]):
        best = (labels[m.group(1)], i)
c = Counter(t[0] for l in L[best[0]:best[1] + 1] for t in [l.strip().split()] if t and re.match(r'^(v_|s_|ds_|global_|buffer_)', t[0]))
tot = sum(c.values())
print(f"loop at lines {best}: {tot} instructions, {tot / unroll:.1f} per iteration; VALU {sum(v for k, v in c.items() if k.startswith('v_')) / unroll:.1f} SALU {sum(v for k, v in c.items() if k.startswith('s_')) / unroll:.1f} LDS {sum(v for k, v in c.items() if k.startswith('ds_')) / unroll:.1f} VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_'))) / unroll:.1f}")
for k, v in c.most_common(30): print(f"{v:6d} {k}")
