#!/usr/bin/env python3
"""One iteration of a loop out of a rocprofv3 kernel trace: every launch between two consecutive launches of an anchor kernel
(default K_preprocess), with its start offset, duration and the idle gap before it.  usage: trace_timeline.py DIR [anchor] [which]
(which: index of the anchor launch counted from the END, default 3)."""
import csv, glob, sys
d = sys.argv[1]; anchor = sys.argv[2] if len(sys.argv) > 2 else "K_preprocess"; which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(({"n": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(f))), key=lambda r: r["s"])
idx = [i for i, r in enumerate(rows) if anchor in r["n"]]
a, b = idx[-which - 1], idx[-which]
t0, prev = rows[a]["s"], None
tot = 0
for r in rows[a:b]:
    gap = 0 if prev is None else r["s"] - prev
    name = r["n"].split("(")[0][-60:]
    print(f"{(r['s'] - t0) / 1e3:9.1f} us  dur {(r['e'] - r['s']) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  {name}")
    prev = r["e"]; tot += r["e"] - r["s"]
print(f"iteration {(rows[b]['s'] - t0) / 1e3:.1f} us, kernels busy {tot / 1e3:.1f} us, {b - a} launches")
