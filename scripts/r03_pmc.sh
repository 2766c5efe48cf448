#!/bin/bash
# one PMC pass (kernel-trace only): scripts/r03_pmc.sh "<counters>" [lib.so]
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmcx
[ -n "$2" ] && export GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$2
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d /tmp/pmcx -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --mode rasterize --steps 3 --warmup 1 > /tmp/pmcx.log 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("/tmp/pmcx/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0].replace("void ","").replace("gsr::","")
        if not k.startswith("K_"): continue
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
names=sorted({c for v in agg.values() for c in v})
print("%-22s"%"kernel"+"".join("%22s"%c for c in names))
for k,v in agg.items(): print("%-22s"%k[:22]+"".join("%22.4g"%(v[c]/max(cnt[(k,c)],1)) for c in names))
PY
