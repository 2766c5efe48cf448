#!/bin/bash
# library variants on the experiment workloads: LIBS="build/a.so build/b.so" scripts/r03_ab_cfg.sh
cd $GRAFT_REPO_ROOT
for lib in $LIBS; do echo "== $lib"; GSR_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$lib bash scripts/r03_cfgs.sh; done
