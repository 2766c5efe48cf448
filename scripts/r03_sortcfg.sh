#!/bin/bash
# per-config time of the two tile-sort kernels for library variants: LIBS="build/a.so build/b.so" scripts/r03_sortcfg.sh
cd $GRAFT_REPO_ROOT
for lib in $LIBS; do
  for args in "--scale-mult 4" "--splats 2000000 --camera scannet" "--splats 10000000" ""; do
    echo "== $lib [$args]"
    GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$lib bash scripts/kstats.sh $args 2>&1 | grep 'tile_sort'
  done
done
