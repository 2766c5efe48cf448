#!/bin/bash
# LDS trim of the backward blend: allocation granule probe, parity of the trimmed build, bench lines of the variants
cd $GRAFT_REPO_ROOT; out=gpurun_out/lds; mkdir -p $out
timeout 120 build/lds_granule > $out/granule.txt 2>&1; cat $out/granule.txt
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.4f ms/step  bwd %.4f  fwd %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['fwd_blend_avg_launch_ms']))"; }
for i in 1 2; do for lib in $LIBS; do GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --no-cpu --mode rasterize --steps 20 --warmup 5 2>>$out/bench.err | line $lib; done; done | tee $out/bench.txt
GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$PARLIB timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dual.py -x -q -m gpu -k "small-rgb or tum-10k or odd-sh3 or fat-clamped or dense-long or replica-300k or deep-stack or dual or pair" > $out/parity.txt 2>&1
tail -4 $out/parity.txt
