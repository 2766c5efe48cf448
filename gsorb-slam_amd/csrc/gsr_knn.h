// gsr_knn.h — on-device exact 3-nearest-neighbour mean squared distance (the reference's
// simple_knn / distCUDA2: src/simple_knn.cu:45-219, src/spatial.cu:15-27; caller
// src/Gaussian.cc:59-69, initial scale of newly inserted Gaussians).
//
// What is computed is fixed by the reference: for every point the mean of the three smallest
// squared distances to OTHER points (self excluded by index, duplicates count with distance 0),
// (d0 + d1 + d2) / 3 with d0 <= d1 <= d2. The search structure is free; this one reuses the
// rasterizer's binning machinery instead of a global radix sort:
//   K_knn_bbox    bounding box (the reference seeds both reductions with 0, simple_knn.cu:191:
//                 the box always contains the origin) via ordered-integer atomic min/max
//   K_knn_code    30-bit Morton code (simple_knn.cu:45-61); the code's top bits pick a bucket and
//                 the returning atomic on the bucket's counter is the point's slot
//   K_scan_tiles  bucket counts -> ranges                         (shared with the rasterizer)
//   K_knn_fill    (code << 32 | index) into the bucket's segment
//   K_tile_sort   per-bucket LDS bitonic sort -> Morton order     (shared with the rasterizer;
//                 equal codes order by index, like the reference's stable radix sort)
//   K_knn_boxes   gathers the points into Morton order and forms the AABB of every 1024 of them
//   K_knn_search  256 queries per workgroup; a box whose AABB can still improve some query of
//                 the workgroup is staged in LDS once and scanned by those queries through
//                 wave-uniform LDS broadcast reads
#pragma once

#include "gsr_device.h"

namespace gsr {

#define GSR_KNN_BOX 1024

// Bucket counter padded to its own 64-byte line: device-scope atomics on counters that share a line serialise
// (measured 12 vs 23 G atomics/s on MI355X).
struct BucketRec {
    uint32_t cnt, start;
    uint32_t pad[14];
};
static_assert(sizeof(BucketRec) == 64, "one line per bucket");

struct KnnView {
    GeomHeader* hdr;
    uint32_t* bbox;     // [6] ordered-uint min xyz, max xyz
    BucketRec* buckets; // [nb]
    uint2* ranges;      // [nb]
    uint64_t* pairs;    // [P]
    uint32_t* order;    // [P] Morton-sorted point indices
    uint32_t* code;     // [P]
    uint32_t* slot;     // [P]
    float4* spts;       // [P] points in Morton order, w = original index bits
    float* boxes;       // [nbox][8] min xyz, pad, max xyz, pad
};

__host__ __device__ inline int knn_bucket_bits(int P)
{
    int b = 3;
    while (b < 18 && ((size_t)1 << b) * 128 < (size_t)P) b++;
    return b;
}
__host__ __device__ inline size_t knn_layout(char* base, int P, KnnView* v)
{
    size_t off = 0, Pz = P > 0 ? (size_t)P : 1;
    const size_t nb = (size_t)1 << knn_bucket_bits(P), nbox = (Pz + GSR_KNN_BOX - 1) / GSR_KNN_BOX;
    KnnView k;
    k.hdr = (GeomHeader*)(base + off); off = gsr_align_up(off + sizeof(GeomHeader));
    k.bbox = (uint32_t*)(base + off); off = gsr_align_up(off + 32);
    k.buckets = (BucketRec*)(base + off); off = gsr_align_up(off + nb * sizeof(BucketRec));
    k.ranges = (uint2*)(base + off); off = gsr_align_up(off + nb * 8);
    k.pairs = (uint64_t*)(base + off); off = gsr_align_up(off + Pz * 8);
    k.order = (uint32_t*)(base + off); off = gsr_align_up(off + Pz * 4);
    k.code = (uint32_t*)(base + off); off = gsr_align_up(off + Pz * 4);
    k.slot = (uint32_t*)(base + off); off = gsr_align_up(off + Pz * 4);
    k.spts = (float4*)(base + off); off = gsr_align_up(off + Pz * 16);
    k.boxes = (float*)(base + off); off = gsr_align_up(off + nbox * 32);
    if (v) *v = k;
    return off;
}

// order-preserving float <-> uint mapping (for atomic min / max on mixed-sign floats)
__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__global__ void K_knn_init(uint32_t* bbox)
{
    if (threadIdx.x < 3) bbox[threadIdx.x] = f2ord(0.0f);        // min seeded with 0 (simple_knn.cu:191)
    else if (threadIdx.x < 6) bbox[threadIdx.x] = f2ord(0.0f);   // max seeded with 0
}

__global__ void __launch_bounds__(256)
K_knn_bbox(int P, const float* __restrict__ pts, uint32_t* __restrict__ bbox)
{
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
        for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&bbox[k], f2ord(mn[k])); atomicMax(&bbox[3 + k], f2ord(mx[k])); }
    }
}

// simple_knn.cu:45-61
__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
__device__ __forceinline__ uint32_t morton_axis(float c, float lo, float hi)
{
    const float t = ((c - lo) / (hi - lo)) * (float)((1 << 10) - 1);
    return prep_morton((uint32_t)fminf(fmaxf(t, 0.f), 1023.f)); // degenerate axes (hi == lo) give NaN -> 0
}

__global__ void __launch_bounds__(256)
K_knn_code(int P, int bucket_shift, const float* __restrict__ pts, const uint32_t* __restrict__ bbox,
           BucketRec* __restrict__ buckets, uint32_t* __restrict__ code, uint32_t* __restrict__ slot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    const uint32_t c = morton_axis(x, ord2f(bbox[0]), ord2f(bbox[3])) | (morton_axis(y, ord2f(bbox[1]), ord2f(bbox[4])) << 1) |
                       (morton_axis(z, ord2f(bbox[2]), ord2f(bbox[5])) << 2);
    code[i] = c;
    slot[i] = atomicAdd(&buckets[c >> bucket_shift].cnt, 1u);
}

__global__ void __launch_bounds__(256)
K_knn_fill(int P, int bucket_shift, const uint32_t* __restrict__ code, const uint32_t* __restrict__ slot,
           const BucketRec* __restrict__ buckets, uint64_t* __restrict__ pairs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t c = code[i];
    pairs[buckets[c >> bucket_shift].start + slot[i]] = ((uint64_t)c << 32) | (uint32_t)i;
}

__global__ void __launch_bounds__(256)
K_knn_boxes(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order, float4* __restrict__ spts,
            float* __restrict__ boxes)
{
    __shared__ float red[4][6];
    const int b = blockIdx.x, tid = threadIdx.x;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = b * GSR_KNN_BOX + tid; i < min(P, (b + 1) * GSR_KNN_BOX); i += 256) {
        const uint32_t id = order[i];
        const float x = pts[3 * (size_t)id], y = pts[3 * (size_t)id + 1], z = pts[3 * (size_t)id + 2];
        spts[i] = make_float4(x, y, z, __uint_as_float(id));
        mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
        mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
    if ((tid & 63) == 0)
        for (int k = 0; k < 3; k++) { red[tid >> 6][k] = mn[k]; red[tid >> 6][3 + k] = mx[k]; }
    __syncthreads();
    if (tid < 3) {
        boxes[8 * b + tid] = fminf(fminf(red[0][tid], red[1][tid]), fminf(red[2][tid], red[3][tid]));
        boxes[8 * b + 4 + tid] = fmaxf(fmaxf(red[0][3 + tid], red[1][3 + tid]), fmaxf(red[2][3 + tid], red[3][3 + tid]));
    }
}

// simple_knn.cu:119-129
__device__ __forceinline__ float dist_box_point(const float* __restrict__ box, float3 p)
{
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (p.x < box[0] || p.x > box[4]) dx = fminf(fabsf(p.x - box[0]), fabsf(p.x - box[4]));
    if (p.y < box[1] || p.y > box[5]) dy = fminf(fabsf(p.y - box[1]), fabsf(p.y - box[5]));
    if (p.z < box[2] || p.z > box[6]) dz = fminf(fabsf(p.z - box[2]), fabsf(p.z - box[6]));
    return dx * dx + dy * dy + dz * dz;
}
// simple_knn.cu:131-145 (K = 3): keeps best[] ascending
__device__ __forceinline__ void update3(float3 ref, float3 p, float (&best)[3])
{
    const float dx = p.x - ref.x, dy = p.y - ref.y, dz = p.z - ref.z;
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
}

__global__ void __launch_bounds__(256)
K_knn_search(int P, int nbox, const float4* __restrict__ spts, const float* __restrict__ boxes,
             float* __restrict__ dists)
{
    __shared__ float4 sbox[GSR_KNN_BOX];
    const int tid = threadIdx.x, idx = blockIdx.x * 256 + tid;
    const bool live = idx < P;
    const float4 me4 = live ? spts[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float3 me = make_float3(me4.x, me4.y, me4.z);
    float best[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float reject = 3.402823466e+38f;
    if (live) { // simple_knn.cu:155-166: the +-3 Morton neighbours give the rejection bound
        for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
            if (i == idx) continue;
            const float4 q = spts[i];
            update3(me, make_float3(q.x, q.y, q.z), best);
        }
        reject = best[2];
        best[0] = best[1] = best[2] = 3.402823466e+38f;
    }
    // visit the query block's own box first (tightens best[] early), then all others
    const int own = (blockIdx.x * 256) / GSR_KNN_BOX;
    for (int t = 0; t < nbox; t++) {
        const int b = t == 0 ? own : (t <= own ? t - 1 : t);
        bool need = false;
        if (live) {
            const float d = dist_box_point(boxes + 8 * b, me);
            need = !(d > reject || d > best[2]);
        }
        if (!__syncthreads_or(need)) continue;
        const int lo = b * GSR_KNN_BOX, cnt = min(P, lo + GSR_KNN_BOX) - lo;
        for (int i = tid; i < cnt; i += 256) sbox[i] = spts[lo + i];
        __syncthreads();
        if (need) {
            for (int i = 0; i < cnt; i++) {
                if (lo + i == idx) continue;
                const float4 q = sbox[i];
                update3(me, make_float3(q.x, q.y, q.z), best);
            }
        }
        __syncthreads();
    }
    if (live) dists[__float_as_uint(me4.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

} // namespace gsr
