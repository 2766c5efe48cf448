#!/bin/bash
# binning-chain A/B: per-kernel rocprof averages of the binning kernels for library variants: LIBS="..." [ARGS="--splats ..."] scripts/r05_binab.sh
cd $GRAFT_REPO_ROOT
for lib in $LIBS; do
  if [ "$lib" = default ]; then unset GSR_LIB_OVERRIDE; else export GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$lib; fi
  echo "== $lib $ARGS"; bash scripts/kstats.sh --no-other --prewarm 50 $ARGS 2>&1 | grep "K_pre\|K_bin\|K_tile_sort" | awk '{printf "%s %s us; ", substr($1,1,16), $(NF-1)} END {print ""}'
done
