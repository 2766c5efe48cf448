#!/bin/bash
# SQ counters of the SSIM kernels (scripts/ssim_time.py as the workload; separate passes, kernel-trace only) -> gpurun_out/pmc_ssim.txt
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_ssim.txt; : > $out
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmcx
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcx -- python $GRAFT_REPO_ROOT/scripts/ssim_time.py > /tmp/pmcx.log 2>&1
  python - >> $out <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("/tmp/pmcx/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0].replace("void ","").replace("gsr::","")
        if "ssim" not in k: continue
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
for k,v in sorted(agg.items()):
    for c,x in sorted(v.items()): print("%-24s %-24s %.5g  (%d launches)"%(k,c,x/cnt[(k,c)],cnt[(k,c)]))
PY
done
cat $out
