#!/bin/bash
# per-kernel time of the C++ tracking / mapping loop (tests/cpp/slam_loop_main.bin) at 1 M Gaussians, 1200x680:
# writes the scene file through bench.py's cpp_loop_ms, then rocprofv3 --kernel-trace --stats around the binary alone
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-loopprof}; mkdir -p $out
export GSR_LOOP_SCENE_OUT=/tmp/loop_scene.bin
python - <<PY
import sys, types
sys.argv=["bench.py"]
import bench, torch
a=types.SimpleNamespace(camera="replica", other_steps=5)
gsr=bench.entry.load_package(); gsr.lib()
print(bench.cpp_loop_ms(a, gsr, torch.device("cuda",0), track_iters=${TRACK:-20}, map_iters=${MAP:-20}))
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lp && GSR_LOOP_NORMAL_EXIT=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- $GRAFT_REPO_ROOT/tests/cpp/slam_loop_main.bin /tmp/loop_scene.bin > $GRAFT_REPO_ROOT/$out/run.log 2>&1
tail -3 $GRAFT_REPO_ROOT/$out/run.log
python - <<PY
import csv,glob
f=glob.glob("/tmp/lp/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU time %.1f ms over %d kernels" % (tot/1e6, len(rows)))
for r in rows[:40]: print("%-70s %5s %8.1f us  %5.1f%%" % (r["Name"].replace("void ","").replace("gsr::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
