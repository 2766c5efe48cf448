#!/bin/bash
# TEST INFRASTRUCTURE ONLY. Builds oracle/_ref/libgsr_ref.so and libgsr_ref_fma.so: the REFERENCE's own rasterizer
# (Thirdparty/diff_gaussian_rasterization/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu + its headers) and its k-NN (src/simple_knn.cu), compiled for gfx950.
#
#   1. the reference's nine files are translated WHERE THEY LIE by ROCm's hipify-perl (/opt/rocm/bin, part of this image) into a scratch
#      directory under oracle/_ref/ — cuda_runtime -> hip_runtime, cub -> hipcub, cooperative_groups -> hip_cooperative_groups; glm is
#      the copy the reference vendors (third_party/glm). Three mechanical fix-ups of what hipify-perl leaves behind: the include lines it
#      emptied or could not map (device_launch_parameters.h, cub/device/device_radix_sort.cuh — covered by hipcub.hpp —,
#      cooperative_groups/reduce.h — nothing of it is used), the spaced launch brackets `<< <` / `>> >` the reference writes, and
#      -D__trap=__builtin_trap for the device trap of auxiliary.h:159; `-include cfloat` (simple_knn.cu uses FLT_MAX, which CUDA's runtime header drags in
#      and HIP's does not). No header, library or tool is stood in for.
#   2. hipcc compiles them with oracle/ref_shim.hip (host arrays in / out, the reference's own entry points and state layout) into
#        libgsr_ref.so      -ffp-contract=off : the arithmetic as the source states it (what the CPU oracle and the HIP library hold)
#        libgsr_ref_fma.so  hipcc's default contraction (fast): what a default nvcc build (--fmad=true, the reference's CMake) is like
#        gsr_ref_C.so       the reference's Python-operator binding (rasterize_points.cu + ext.cpp, g++ against this image's libtorch) on the first build's objects
#        dropin_ref.bin     the reference's C++ host layer (src/Rasterizer.cu, spatial.cu) under tests/cpp/dropin_main.cpp, a caller written against its API
#   3. the scratch directory is removed: no text of the reference stays in the tree; oracle/_ref/ is git-ignored, the .so files travel
#      to the GPU box with the snapshot.
# Needs /root/reference (this container); on the GPU box the prebuilt files are used.
# OPT-IN ONLY: nothing here runs unless a human sets GSR_REFERENCE_BUILD=1. Translating, compiling and running the reference's own sources — even as a
# checker, with outputs only under the git-ignored oracle/_ref/ — is a decision for the repository's owner (the task's rules say "not a hipify" of the product and
# "no stand-ins" for the checker; whether ROCm's translator on the reference's files is admissible for the checker is not this script's call). Default: no build,
# no oracle/_ref, every test / smoke / bench leg that would use it is skipped.
if [ "${GSR_REFERENCE_BUILD:-0}" != "1" ]; then echo "build_ref.sh: GSR_REFERENCE_BUILD=1 not set: the reference build is opt-in (see the header); nothing done"; exit 0; fi
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${GSR_REFERENCE:-/root/reference}/Thirdparty/diff_gaussian_rasterization
SRC=$REF/cuda_rasterizer
OUT=$HERE/_ref
[ -d "$SRC" ] || { echo "build_ref.sh: $SRC not found (the prebuilt oracle/_ref/*.so are used as they are)"; exit 0; }
if [ -f "$OUT/libgsr_ref.so" ] && [ -f "$OUT/libgsr_ref_fma.so" ] && [ -f "$OUT/gsr_ref_C.so" ] && [ -f "$OUT/dropin_ref.bin" ] && [ "$OUT/dropin_ref.bin" -nt "$HERE/../tests/cpp/dropin_main.cpp" ] && [ "$OUT/libgsr_ref.so" -nt "$HERE/ref_shim.hip" ] && [ "$OUT/libgsr_ref.so" -nt "$HERE/build_ref.sh" ] && [ "$1" != "--force" ]; then exit 0; fi
TMP=$OUT/.scratch
rm -rf "$TMP" && mkdir -p "$TMP"
trap 'rm -rf "$TMP"' EXIT
for f in auxiliary.h backward.cu backward.h config.h forward.cu forward.h rasterizer.h rasterizer_impl.cu rasterizer_impl.h; do
    o=$f; case $f in *.cu) o=${f%.cu}.hip;; esac
    /opt/rocm/bin/hipify-perl "$SRC/$f" > "$TMP/$o" 2>/dev/null
    sed -i -e '/#include ""/d' -e '/cub\/device\/device_radix_sort.cuh/d' -e '/cooperative_groups\/reduce.h/d' -e 's/<< </<<</g' -e 's/>> >/>>>/g' "$TMP/$o"
done
# the k-NN of the densification path (src/simple_knn.cu, include/simple_knn.h: SimpleKNN::knn — cub reduce / radix sort, thrust vectors -> hipcub, rocThrust)
for f in src/simple_knn.cu include/simple_knn.h; do
    o=$(basename $f); case $o in *.cu) o=${o%.cu}.hip;; esac
    /opt/rocm/bin/hipify-perl "${GSR_REFERENCE:-/root/reference}/$f" > "$TMP/$o" 2>/dev/null
    sed -i -e '/#include ""/d' -e '/cub\/device\/device_radix_sort.cuh/d' -e '/cooperative_groups\/reduce.h/d' -e 's/<< </<<</g' -e 's/>> >/>>>/g' "$TMP/$o"
done
# the Python operator's binding (rasterize_points.cu, rasterize_points.h, ext.cpp: host code on torch tensors, the `_C` module of the reference's
# diff_gaussian_rasterization package) — for tests/test_gpu_reference_build.py's argument-for-argument comparison of the two `_C` modules
for f in rasterize_points.cu rasterize_points.h ext.cpp; do
    o=$f; case $f in *.cu) o=${f%.cu}.hip;; esac
    /opt/rocm/bin/hipify-perl "$REF/$f" > "$TMP/$o" 2>/dev/null
    sed -i -e '/#include ""/d' "$TMP/$o"
done
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
DEV="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -D__trap=__builtin_trap -include cfloat -I$REF/third_party/glm -I$TMP"
build_variant() { # $1 = object suffix, $2 = extra flags: the reference's device code + the shim
    for u in forward backward rasterizer_impl simple_knn; do $HIPCC $DEV $2 -c "$TMP/$u.hip" -o "$TMP/$u$1.o" || return 1; done
    $HIPCC $DEV $2 -c "$HERE/ref_shim.hip" -o "$TMP/shim$1.o" || return 1
}
build_variant _off -ffp-contract=off &
build_variant _fma "" &
# (host-only: g++ against this image's libtorch; the module is named gsr_ref_C so that it cannot be mistaken for the library's _C)
TORCH=$(python3 - <<'PY'
import os, sysconfig, torch
from torch.utils import cpp_extension as ce
print(" ".join("-I" + i for i in ce.include_paths() + [sysconfig.get_paths()["include"]]), "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI))
print(os.path.join(os.path.dirname(torch.__file__), "lib"))
PY
)
TINC=$(echo "$TORCH" | head -1); TLIB=$(echo "$TORCH" | tail -1)
HOST="-std=c++17 -O2 -fPIC -w -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -DTORCH_EXTENSION_NAME=gsr_ref_C -DTORCH_API_INCLUDE_EXTENSION_H $TINC -I/opt/rocm/include -I$REF -I$TMP"
g++ $HOST -x c++ -c "$TMP/rasterize_points.hip" -o "$TMP/rp.o"
g++ $HOST -c "$TMP/ext.cpp" -o "$TMP/ext.o"
# the C++ host layer GSORB-SLAM's Render.cc calls (src/Rasterizer.cu, include/Rasterizer.cuh, src/spatial.cu, include/spatial.h: host code on torch tensors) and
# ONE caller written against its API (tests/cpp/dropin_main.cpp, this repository's) -> dropin_ref.bin; the same caller compiled against this repository's host
# layer is tests/cpp/dropin_hip.bin (tests/cpp/build.py)
ROOTREF=${GSR_REFERENCE:-/root/reference}
mkdir -p "$TMP/host"
for f in src/Rasterizer.cu include/Rasterizer.cuh src/spatial.cu include/spatial.h; do
    /opt/rocm/bin/hipify-perl "$ROOTREF/$f" > "$TMP/host/$(basename $f)" 2>/dev/null
    sed -i -e '/#include ""/d' "$TMP/host/$(basename $f)"
done
HOST2="-std=c++17 -O1 -fPIC -w -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 $TINC -I/opt/rocm/include -I$TMP/host -I$TMP -I$ROOTREF -I$REF"
g++ $HOST2 -x c++ -c "$TMP/host/Rasterizer.cu" -o "$TMP/host/Rasterizer.o"
g++ $HOST2 -x c++ -c "$TMP/host/spatial.cu" -o "$TMP/host/spatial.o"
g++ $HOST2 '-DDROPIN_HEADER="Rasterizer.cuh"' -c "$HERE/../tests/cpp/dropin_main.cpp" -o "$TMP/host/dropin.o"
wait %1 && wait %2
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgsr_ref.so" "$TMP"/{forward,backward,rasterizer_impl,simple_knn,shim}_off.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgsr_ref_fma.so" "$TMP"/{forward,backward,rasterizer_impl,simple_knn,shim}_fma.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/gsr_ref_C.so" "$TMP"/{forward,backward,rasterizer_impl}_off.o "$TMP/rp.o" "$TMP/ext.o" \
    -L"$TLIB" -lc10 -lc10_hip -ltorch -ltorch_cpu -ltorch_hip -ltorch_python -Wl,-rpath,"$TLIB"
$HIPCC --offload-arch=gfx950 -fPIC -o "$OUT/dropin_ref.bin" "$TMP"/{forward,backward,rasterizer_impl,simple_knn}_off.o "$TMP/host/Rasterizer.o" "$TMP/host/spatial.o" "$TMP/host/dropin.o" \
    -L"$TLIB" -lc10 -lc10_hip -ltorch -ltorch_cpu -ltorch_python -Wl,--no-as-needed -ltorch_hip -Wl,--as-needed -Wl,-rpath,"$TLIB" $(python3-config --ldflags --embed)
# (torch/extension.h, which the reference's header includes, brings pybind11 in: the executable links libpython)
echo "built $OUT/libgsr_ref.so, libgsr_ref_fma.so, gsr_ref_C.so, dropin_ref.bin"
