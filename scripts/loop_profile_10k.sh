cd $GRAFT_REPO_ROOT
export GSR_LOOP_SCENE_OUT=/tmp/loop_scene.bin
python - <<PY
import sys, types
sys.argv=["bench.py"]
import bench, torch
a=types.SimpleNamespace(camera="tum", other_steps=5)
gsr=bench.entry.load_package(); gsr.lib()
r=bench.cpp_loop_ms(a, gsr, torch.device("cuda",0), P=10000, track_iters=40, map_iters=40)
print({k:v for k,v in r.items() if k!="what"})
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lp && GSR_LOOP_NORMAL_EXIT=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- $GRAFT_REPO_ROOT/tests/cpp/slam_loop_main.bin /tmp/loop_scene.bin > /tmp/run.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/lp/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("total GPU kernel time %.2f ms over %d launches" % (tot/1e6, calls))
for r in rows[:14]: print("%-60s %5s %8.1f us" % (r["Name"].replace("void ","").replace("gsr::","")[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
