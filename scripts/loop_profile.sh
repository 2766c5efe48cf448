#!/bin/bash
# per-kernel time of the C++ tracking / mapping loop (tests/cpp/slam_loop_main.bin) at 1 M Gaussians, 1200x680:
# writes the scene file through bench.py's cpp_loop_ms, then rocprofv3 --kernel-trace --stats around the binary alone
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-loopprof}; mkdir -p $out
export GSR_LOOP_SCENE_OUT=/tmp/loop_scene.bin
python - <<PY
import sys, types
sys.argv=["bench.py"]
import bench, torch
a=types.SimpleNamespace(camera="replica", other_steps=5, loop_warmup=0)
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
# one iteration of each kind, kernel by kernel, with the gaps between the launches (start of this one - end of the previous)
t=sorted(csv.DictReader(open(glob.glob("/tmp/lp/**/*kernel_trace.csv",recursive=True)[0])), key=lambda r:int(r["Start_Timestamp"]))
nm=lambda r: r["Kernel_Name"].split("(")[0].replace("void ","").replace("gsr::","")[:44]
def dump(anchor, what):
    idx=[i for i,r in enumerate(t) if nm(r).startswith(anchor)]
    if len(idx) < 6: return
    a,b=idx[len(idx)//2], idx[len(idx)//2+1]
    print("--- one %s iteration (%.1f us from %s to the next)" % (what, (int(t[b]["Start_Timestamp"])-int(t[a]["Start_Timestamp"]))/1e3, anchor))
    for i in range(a,b):
        print("   gap %5.1f  run %6.1f  %s" % ((int(t[i]["Start_Timestamp"])-int(t[i-1]["End_Timestamp"]))/1e3, (int(t[i]["End_Timestamp"])-int(t[i]["Start_Timestamp"]))/1e3, nm(t[i])))
dump("K_track_loss", "tracking")
dump("K_map_finish", "mapping")
PY
