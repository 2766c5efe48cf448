#!/bin/bash
# A/B of library variants over the bench workloads: LIBS="default build/a.so ..." scripts/r04_ab.sh <tag>
# ("default" = the in-tree csrc/libgsr_hip.so). One bench line per (workload, library): ms/step, backward / forward blend ms.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-ab}; mkdir -p $out
WL=${WL:-"headline|  fat4|--scale-mult 4  x2|--scale-mult 2  scan2M|--splats 2000000 --camera scannet  walls|--depth-layout two-walls  10M|--splats 10000000"}
echo "$WL" | sed 's/  /\n/g' | while IFS='|' read name args; do
  [ -z "$name" ] && continue
  for lib in $LIBS; do
    if [ "$lib" = default ]; then unset GSR_LIB_OVERRIDE; else export GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$lib; fi
    timeout 300 python bench.py --no-cpu --no-other --mode rasterize --steps ${STEPS:-30} --warmup 5 --prewarm ${PREWARM:-100} $args 2>>$out/bench.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-9s %-22s %.4f ms/step  bwd %.4f  fwd %.4f  R=%d' % ('$name', '$lib', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['fwd_blend_avg_launch_ms'], d['config']['tile_instances']))"
  done
done | tee $out/ab.txt
