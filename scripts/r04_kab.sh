#!/bin/bash
# per-kernel A/B of library variants on the headline frame: LIBS="default build/x.so ..." scripts/r04_kab.sh  (rocprofv3 --kernel-trace --stats each)
cd $GRAFT_REPO_ROOT
for lib in $LIBS; do
  if [ "$lib" = default ]; then unset GSR_LIB_OVERRIDE; else export GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$lib; fi
  echo "== $lib"; bash scripts/kstats.sh --no-other --prewarm 100 $ARGS 2>&1 | grep "K_" | head -${TOP:-12}
done
