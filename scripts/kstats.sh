#!/bin/bash
# per-kernel average durations of one bench run (timeout 300 rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kst && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --mode rasterize "$@" > /tmp/kst.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/kst/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:11]: print("%-44s %4s %8.1f us" % (r["Name"].replace("void ","").replace("gsr::","")[:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
