#!/bin/bash
# quick iteration on the GPU box: parity subset, two driver-style bench lines, per-kernel averages
# usage: scripts/r03_try.sh <tag> [pytest -k expression] ; env LIBS="a.so b.so" benches more library variants (GSR_LIB_OVERRIDE)
tag=${1:-try}; kexpr=${2:-"small-rgb or tum-10k or odd-sh3 or fat-clamped or dense-long or replica-300k or deep-stack"}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$kexpr" > $out/parity.txt 2>&1
tail -4 $out/parity.txt
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.4f ms/step  bwd %.4f  fwd %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['fwd_blend_avg_launch_ms']))"; }
for i in 1 2; do timeout 300 python bench.py --no-cpu --mode rasterize --steps 20 --warmup 5 2>>$out/bench.err | tee $out/bench_$i.json | line default; done
for lib in $LIBS; do for i in 1 2; do GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --no-cpu --mode rasterize --steps 20 --warmup 5 2>>$out/bench.err | line $lib; done; done
bash scripts/kstats.sh 2>&1 | tee $out/kstats.txt
