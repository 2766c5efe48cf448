#!/bin/bash
# three bench lines: ms/step, backward blend ms, forward blend ms
for i in 1 2 3; do timeout 300 python bench.py --no-cpu --mode rasterize --steps 50 --warmup 10 "$@" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  bwd %.4f  fwd %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['fwd_blend_avg_launch_ms']))"; done
