#!/bin/bash
# round 3, first call: microbench, baseline bench of the round-2 build, new full-size parity cases, LDS counters
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_call1; mkdir -p $out
./build/dpp_bcast_bench > $out/dpp_bench.txt 2>&1
python bench.py --steps 20 --warmup 5 --mode rasterize > $out/bench_20.json 2> $out/bench_20.err
python bench.py --steps 200 --warmup 20 --mode rasterize --no-cpu > $out/bench_200.json 2>> $out/bench_20.err
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "replica-1M or scannet-2M" -s > $out/parity_big.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CU_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_lds -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --mode rasterize --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$out/pmc_lds.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("$out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()): print("   %-24s %.4g  (per launch, %d launches)"%(c,x/cnt[(k,c)],cnt[(k,c)]))
PY
find $out -type f ! -name '*.csv' ! -name '*.json' ! -name '*.log' ! -name '*.err' ! -name '*.txt' -delete
cat $out/dpp_bench.txt; tail -3 $out/parity_big.txt; python -c "
import json
for f in ('bench_20','bench_200'):
    d=json.loads(open('$out/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['fwd_blend_avg_launch_ms'])
"
