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
#   3. the scratch directory is removed: no text of the reference stays in the tree; oracle/_ref/ is git-ignored, the .so files travel
#      to the GPU box with the snapshot.
# Needs /root/reference (this container); on the GPU box the prebuilt files are used.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${GSR_REFERENCE:-/root/reference}/Thirdparty/diff_gaussian_rasterization
SRC=$REF/cuda_rasterizer
OUT=$HERE/_ref
[ -d "$SRC" ] || { echo "build_ref.sh: $SRC not found (the prebuilt oracle/_ref/*.so are used as they are)"; exit 0; }
if [ -f "$OUT/libgsr_ref.so" ] && [ -f "$OUT/libgsr_ref_fma.so" ] && [ "$OUT/libgsr_ref.so" -nt "$HERE/ref_shim.hip" ] && [ "$OUT/libgsr_ref.so" -nt "$HERE/build_ref.sh" ] && [ "$1" != "--force" ]; then exit 0; fi
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
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -D__trap=__builtin_trap -include cfloat -I$REF/third_party/glm -I$TMP"
$HIPCC $COMMON -ffp-contract=off -o "$OUT/libgsr_ref.so" "$TMP/forward.hip" "$TMP/backward.hip" "$TMP/rasterizer_impl.hip" "$TMP/simple_knn.hip" "$HERE/ref_shim.hip" &
$HIPCC $COMMON -o "$OUT/libgsr_ref_fma.so" "$TMP/forward.hip" "$TMP/backward.hip" "$TMP/rasterizer_impl.hip" "$TMP/simple_knn.hip" "$HERE/ref_shim.hip" &
wait %1 && wait %2
echo "built $OUT/libgsr_ref.so, libgsr_ref_fma.so"
