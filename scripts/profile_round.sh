#!/bin/bash
# Collects everything the profiles/ summaries are built from (run on the GPU box through gpurun):
#   1. un-profiled bench line            -> gpurun_out/prof_<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats  -> gpurun_out/prof_<tag>/stats/
#   3. PMC passes (kernel-trace only, one counter group per run: SQ groups, FETCH_SIZE, WRITE_SIZE, TCC hit/miss)
# then scripts/profile_report.py (run in the build container) turns them into profiles/<tag>_*.md/json.
tag=${1:-r01x}
repo=$GRAFT_REPO_ROOT
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $repo/bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err   # the driver's command line
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $repo/bench.py --mode rasterize --steps 20 --warmup 5 --no-cpu --no-other > $out/stats.log 2>&1
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python $repo/bench.py --no-cpu --no-other --mode rasterize --steps 3 --warmup 1 --prewarm 20 > $out/pmc$i.log 2>&1
  python $repo/scripts/pmc_condense.py $out/pmc$i
done
python $repo/scripts/pmc_condense.py $out/stats
# keep only the csv files (the merge-back limit is 64 MiB)
find $out -type f ! -name '*.csv' ! -name '*.json' ! -name '*.log' ! -name '*.err' -delete
du -sh $out
