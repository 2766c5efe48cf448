#!/bin/bash
# SQ counters of the blend kernels (separate passes, kernel-trace only). Usage: scripts/pmc_blend.sh <tag> [env...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  env "$@" rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$n -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --mode rasterize --steps 3 --warmup 1 > $out/$n.log 2>&1
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]
        import os
        if not any(f in k for f in os.environ.get("KFILTER", "blend").split(",")): continue
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()): print("   %-24s %.4g  (per launch, %d launches)"%(c,x/cnt[(k,c)],cnt[(k,c)]))
PY
