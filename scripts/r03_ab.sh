#!/bin/bash
# A/B of library variants: LIBS="build/a.so build/b.so" scripts/r03_ab.sh <tag>   (two bench lines each, interleaved)
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-ab}; mkdir -p $out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.4f ms/step  bwd %.4f  fwd %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['fwd_blend_avg_launch_ms']))"; }
for i in 1 2; do for lib in $LIBS; do GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --no-cpu --mode rasterize --steps 20 --warmup 5 $BENCH_ARGS 2>>$out/bench.err | line $lib; done; done | tee $out/bench.txt
