#!/bin/bash
# kasm.sh <outdir> <mangled-name substring> [-Dflags...] : compiles csrc/gsr_api.hip with -save-temps into <outdir> and writes <outdir>/k.s (one kernel's assembly)
d=$1; k=$2; shift 2; mkdir -p $d && cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared "$@" -save-temps=obj -o $d/x.so /root/repo/gsorb-slam_amd/csrc/gsr_api.hip && /root/repo/scripts/kisa.sh $d/gsr_api-hip-amdgcn-amd-amdhsa-gfx950.s $k > $d/k.s && wc -l $d/k.s
