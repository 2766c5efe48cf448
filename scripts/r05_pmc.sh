#!/bin/bash
# SQ counters of the headline frame's kernels for one library variant (separate passes, kernel-trace only, --no-other: the headline frame alone):
#   scripts/r05_pmc.sh <tag> [lib.so]   -> gpurun_out/pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1.txt; : > $out
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS"; do
  rm -rf /tmp/pmcx
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcx -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-other --mode rasterize --steps 3 --warmup 1 --prewarm 20 > /tmp/pmcx.log 2>&1
  python - >> $out <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("/tmp/pmcx/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0].replace("void ","").replace("gsr::","")
        if not k.startswith("K_blend"): continue
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
for k,v in agg.items():
    for c,x in sorted(v.items()): print("%-40s %-24s %.5g  (%d launches)"%(k,c,x/cnt[(k,c)],cnt[(k,c)]))
PY
done
cat $out
