cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/kt; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --mode rasterize --steps 20 --warmup 5 --no-cpu --no-other > $out/log.txt 2>&1
python - <<'PY'
import csv,glob,os
f=sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/kt/**/*kernel_stats.csv",recursive=True))[-1]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0][-40:]
    if "bin" in n or "preprocess" in n or "sort" in n: print(n, r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
find $out -type f ! -name '*.txt' -delete
