cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ss_trace; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --mode shard-step --shard-steps 10 --no-cpu > $out/log.txt 2>&1
python $GRAFT_REPO_ROOT/scripts/trace_timeline.py $out K_preprocess 3 > $out/timeline_track.txt 2>&1
python $GRAFT_REPO_ROOT/scripts/trace_timeline.py $out K_preprocess 14 > $out/timeline_map.txt 2>&1
find $out -name '*.csv' -size +20M -delete; find $out -type f ! -name '*.txt' ! -name '*.csv' -delete
tail -3 $out/log.txt; cat $out/timeline_map.txt
