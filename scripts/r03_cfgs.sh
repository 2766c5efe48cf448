#!/bin/bash
# step time of the experiment workloads next to the headline: fat splats, 2 M ScanNet frame, 10 M splats, two-walls
cd $GRAFT_REPO_ROOT
for args in "--scale-mult 4" "--scale-mult 2" "--splats 2000000 --camera scannet" "--splats 10000000" "--depth-layout two-walls" ""; do
  python bench.py --no-cpu --mode rasterize --steps 20 --warmup 5 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$args] %.4f ms/step  bwd %.4f  fwd %.4f  R=%d' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['fwd_blend_avg_launch_ms'], d['config']['tile_instances']))"
done
