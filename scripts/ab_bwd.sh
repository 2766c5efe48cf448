#!/bin/bash
# quick A/B of the backward blend variants: ms/step, bwd blend ms, fwd blend ms
for v in "$@"; do
  GSR_BWD_VARIANT=$v python bench.py --no-cpu --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v: step %.4f ms  bwd %.4f  fwd %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['fwd_blend_avg_launch_ms']))"
done
