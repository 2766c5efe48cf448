#!/bin/bash
# per-kernel time of the C++ (libtorch) tracking / mapping loop at 1 M Gaussians: rocprofv3 --kernel-trace --stats around the loop test
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cl && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cl -- python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_cpp_loop.py -x -q -m gpu -k 1000000 > /tmp/cl.log 2>&1
tail -3 /tmp/cl.log
python - <<PY
import csv,glob
for f in sorted(glob.glob("/tmp/cl/**/*kernel_stats.csv",recursive=True)):
    rows=list(csv.DictReader(open(f)))
    names=" ".join(r["Name"] for r in rows)
    if "K_adam" not in names: continue
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("==",f, "total ms %.1f" % (tot/1e6))
    for r in rows[:28]: print("%-60s %5s calls %8.1f us avg %6.2f %%" % (r["Name"].replace("void ","").replace("gsr::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
