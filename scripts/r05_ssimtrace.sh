cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ssim_trace; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/scripts/ssim_time.py > $out/log.txt 2>&1
python - <<'PY'
import csv,glob,os
f=sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/ssim_trace/**/*kernel_trace.csv",recursive=True))[-1]
rows=sorted(({"n":r["Kernel_Name"],"s":int(r["Start_Timestamp"]),"e":int(r["End_Timestamp"])} for r in csv.DictReader(open(f))),key=lambda r:r["s"])
for key in ("K_ssim_fwd<true>","K_ssim_fwd<false>","K_ssim_bwd<false>","K_ssim_bwd<true>"):
    d=[(r["e"]-r["s"])/1e3 for r in rows if key in r["n"]]
    # previous kernel name for each launch
    prev=[rows[i-1]["n"].split("(")[0][-30:] for i,r in enumerate(rows) if key in r["n"]]
    import collections
    by=collections.defaultdict(list)
    for x,p in zip(d,prev): by[p].append(x)
    print(key, len(d))
    for p,v in by.items(): print("   after %-32s n=%4d avg %.1f min %.1f"%(p,len(v),sum(v)/len(v),min(v)))
PY
find $out -type f ! -name '*.txt' -delete
