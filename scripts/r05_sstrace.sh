# kernel trace of bench.py --mode shard-step (sharded + unsharded loops of the same scene); per-kernel averages of the LAST mapping phase
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ss_trace; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --mode shard-step --shard-steps 10 --no-cpu > $out/log.txt 2>&1
python - <<'PY' > $out/iters.txt
import csv,glob,os
f=sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/ss_trace/**/*kernel_trace.csv",recursive=True))[-1]
rows=sorted(({"n":r["Kernel_Name"],"s":int(r["Start_Timestamp"]),"e":int(r["End_Timestamp"])} for r in csv.DictReader(open(f))),key=lambda r:r["s"])
idx=[i for i,r in enumerate(rows) if "K_preprocess" in r["n"]]
seen=set()
for j in range(len(idx)-1):
    a,b=idx[j],idx[j+1]
    names=tuple(r["n"].split("(")[0][-40:] for r in rows[a:b])
    if names in seen: continue
    seen.add(names)
    # print the LAST iteration with this signature instead of the first (warm)
    last=[k for k in range(len(idx)-1) if tuple(r["n"].split("(")[0][-40:] for r in rows[idx[k]:idx[k+1]])==names][-1]
    a,b=idx[last],idx[last+1]; t0=rows[a]["s"]
    print("---- iteration signature with",b-a,"launches, occurs",sum(1 for k in range(len(idx)-1) if tuple(r["n"].split("(")[0][-40:] for r in rows[idx[k]:idx[k+1]])==names),"times; total",(rows[b]["s"]-t0)/1e3,"us")
    for r in rows[a:b]: print(f"{(r['s']-t0)/1e3:8.1f} dur {(r['e']-r['s'])/1e3:7.1f}  {r['n'].split('(')[0][-60:]}")
PY
find $out -type f ! -name '*.txt' -delete
