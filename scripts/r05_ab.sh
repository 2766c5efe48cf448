#!/bin/bash
# A/B of library variants over the WHOLE bench line (headline, other workloads, C++ loops): LIBS="default build/a.so ..." scripts/r05_ab.sh <tag>
# Each variant is copied over csrc/libgsr_hip.so of the GPU box's scratch copy (the C++ loop binaries find the library through their rpath).
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-ab5}; mkdir -p $out
cp gsorb-slam_amd/csrc/libgsr_hip.so /tmp/libgsr_hip.default.so
for lib in $LIBS; do
  if [ "$lib" = default ]; then cp /tmp/libgsr_hip.default.so gsorb-slam_amd/csrc/libgsr_hip.so; else cp $lib gsorb-slam_amd/csrc/libgsr_hip.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu ${BENCH_ARGS} 2>>$out/bench.err | tail -1 > $out/line_$(basename $lib .so).json
  python - $out/line_$(basename $lib .so).json $lib <<'PY'
import sys, json
d = json.load(open(sys.argv[1])); r = d["roofline"]
s = "%-24s step %.4f  bwd %.4f fwd %.4f" % (sys.argv[2], d["ms_per_step"], r["avg_launch_ms"], r["fwd_blend_avg_launch_ms"])
l = d.get("loop_ms") or {}
if "mapping" in l: s += " | loop map %.4f track %.4f pair %.4f (bwd %.4f)" % (l["mapping"], l["tracking"], l["raster_pair"], l.get("raster_pair_bwd_blend_ms", 0))
o = d.get("other_workloads") or {}
s += " | " + " ".join("%s %.4f" % (k, v["ms_per_step"]) for k, v in o.items() if isinstance(v, dict) and "ms_per_step" in v)
print(s)
PY
done | tee $out/ab.txt
cp /tmp/libgsr_hip.default.so gsorb-slam_amd/csrc/libgsr_hip.so
