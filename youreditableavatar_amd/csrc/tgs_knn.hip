// tgs_knn.hip -- simple-knn's distCUDA2 for gfx950 ("next" row 4 of SURVEY.md 8f; named in BASELINE.json).
//
// distCUDA2(points[P,3]) -> float[P]: mean of the 3 smallest squared distances to OTHER points
// (Edit_core/thirdparties/simple-knn/spatial.cu:15-26, simple_knn.cu:185-221).  Same algorithm as the reference:
// Morton order, boxes of 1024 consecutive points with their AABB, per point an upper bound from its Morton
// neighbours, then every box whose AABB is not farther than the bound is scanned exhaustively -- so the result is the
// exact 3-NN mean (up to fp32 rounding of the distances).  What differs is how it runs: the Morton sort is the same
// all-ascending bitonic network as the tile sort (LDS for strides < 8192, no cub/thrust), box AABBs by wave
// reductions, and the exhaustive scan reads the Morton-ordered copy of the points with wave-uniform (scalar) loads.
#include "tgs_device.hpp"
#include <cfloat>

namespace tgs {

constexpr int KNN_BOX = 1024;          // BOX_SIZE of simple_knn.cu

struct KnnWork {
    float* partial;                    // [nblk][6] min xyz, max xyz per 256-point block
    float* minmax;                     // [6]
    unsigned long long* keys;          // [P] morton << 32 | index
    float4* sorted;                    // [P] points in Morton order (xyz, bits(original index))
    float* boxes;                      // [nbox][6]
    float* subs;                       // [nbox * 16][6] AABBs of the 64-point groups inside each box
};
__host__ __device__ inline size_t knn_carve(KnnWork& w, char* base, size_t P)
{
    char* p = base;
    const size_t nblk = (P + 255) / 256, nbox = (P + KNN_BOX - 1) / KNN_BOX;
    carve(p, w.partial, nblk * 6); carve(p, w.minmax, 8); carve(p, w.keys, P); carve(p, w.sorted, P); carve(p, w.boxes, nbox * 6); carve(p, w.subs, nbox * 16 * 6);
    return (size_t)(p - base) + 256;
}

__device__ __forceinline__ float wave_min_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_max_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }

// min/max of up to NT points held one per thread -> out[6] (thread 0 writes)
template <int NT>
__device__ __forceinline__ void block_minmax(float x, float y, float z, bool have, float* out)
{
    __shared__ float red[NT / 64][6];
    float v[6] = {have ? x : FLT_MAX, have ? y : FLT_MAX, have ? z : FLT_MAX, have ? x : -FLT_MAX, have ? y : -FLT_MAX, have ? z : -FLT_MAX};
#pragma unroll
    for (int k = 0; k < 3; k++) { v[k] = wave_min_f(v[k]); v[3 + k] = wave_max_f(v[3 + k]); }
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 6; k++) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 6; k++) {
            float r = red[0][k];
            for (int w = 1; w < NT / 64; w++) r = k < 3 ? fminf(r, red[w][k]) : fmaxf(r, red[w][k]);
            out[k] = r;
        }
    }
}

__global__ __launch_bounds__(256) void k_knn_minmax(int P, const float* __restrict__ pts, float* partial)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool have = i < P;
    const float x = have ? pts[3 * (size_t)i] : 0.f, y = have ? pts[3 * (size_t)i + 1] : 0.f, z = have ? pts[3 * (size_t)i + 2] : 0.f;
    block_minmax<256>(x, y, z, have, partial + 6 * (size_t)blockIdx.x);
}
__global__ __launch_bounds__(256) void k_knn_minmax_final(int nblk, const float* __restrict__ partial, float* minmax)
{
    float v[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int b = threadIdx.x; b < nblk; b += 256)
        for (int k = 0; k < 6; k++) v[k] = k < 3 ? fminf(v[k], partial[6 * (size_t)b + k]) : fmaxf(v[k], partial[6 * (size_t)b + k]);
    __shared__ float red[4][6];
    for (int k = 0; k < 3; k++) { v[k] = wave_min_f(v[k]); v[3 + k] = wave_max_f(v[3 + k]); }
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 6; k++) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float r = red[0][k];
        for (int w = 1; w < 4; w++) r = k < 3 ? fminf(r, red[w][k]) : fmaxf(r, red[w][k]);
        minmax[k] = r;
    }
}

// simple_knn.cu:44-61
__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
__global__ __launch_bounds__(256) void k_knn_morton(int P, const float* __restrict__ pts, const float* __restrict__ minmax, unsigned long long* keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = minmax[a], ext = minmax[3 + a] - lo;
        const float t = ext > 0.f ? (pts[3 * (size_t)i + a] - lo) / ext : 0.f;      // the reference divides by zero for a flat cloud
        code |= prep_morton((uint32_t)(t * (float)((1 << 10) - 1))) << a;
    }
    keys[i] = ((unsigned long long)code << 32) | (uint32_t)i;
}

// generic u64 sort of n keys: the tile sort's network on one array
__global__ __launch_bounds__(256) void k_sort64_local(unsigned long long* keys, uint32_t n, uint32_t k_only, uint32_t cap)
{
    extern __shared__ unsigned long long lk[];
    const uint32_t b0 = blockIdx.x * cap;
    if (b0 >= n) return;
    const uint32_t m = min(cap, n - b0), half = cap >> 1;
    unsigned long long* gk = keys + b0;
    for (uint32_t i = threadIdx.x; i < m; i += 256) lk[i] = gk[i];
    __syncthreads();
    if (k_only == 0) {
        for (uint32_t k = 2; k <= cap; k <<= 1) {
            for (uint32_t t = threadIdx.x; t < half; t += 256) { uint32_t i, l; pair_flip(t, k, i, l); cmp_swap(lk, i, l, m); }
            __syncthreads();
            for (uint32_t j = k >> 2; j > 0; j >>= 1) {
                for (uint32_t t = threadIdx.x; t < half; t += 256) { uint32_t i, l; pair_disperse(t, j, i, l); cmp_swap(lk, i, l, m); }
                __syncthreads();
            }
        }
    } else {
        for (uint32_t j = cap >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < half; t += 256) { uint32_t i, l; pair_disperse(t, j, i, l); cmp_swap(lk, i, l, m); }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < m; i += 256) gk[i] = lk[i];
}
__global__ __launch_bounds__(256) void k_sort64_global(unsigned long long* keys, uint32_t n, uint32_t npad, uint32_t k, uint32_t j, int flip)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (npad >> 1)) return;
    uint32_t i, l;
    if (flip) pair_flip(t, k, i, l); else pair_disperse(t, j, i, l);
    cmp_swap(keys, i, l, n);
}

__global__ __launch_bounds__(256) void k_knn_gather(int P, const float* __restrict__ pts, const unsigned long long* __restrict__ keys, float4* sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t id = (uint32_t)keys[i];
    sorted[i] = make_float4(pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2], __uint_as_float(id));
}

// boxMinMax (simple_knn.cu:78-117), plus the AABB of every wave's 64 points (second level of the rejection test)
__global__ __launch_bounds__(KNN_BOX) void k_knn_boxes(int P, const float4* __restrict__ sorted, float* boxes, float* subs)
{
    const int i = blockIdx.x * KNN_BOX + threadIdx.x;
    const bool have = i < P;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) p = sorted[i];
    {
        float v[6] = {have ? p.x : FLT_MAX, have ? p.y : FLT_MAX, have ? p.z : FLT_MAX, have ? p.x : -FLT_MAX, have ? p.y : -FLT_MAX, have ? p.z : -FLT_MAX};
#pragma unroll
        for (int k = 0; k < 3; k++) { v[k] = wave_min_f(v[k]); v[3 + k] = wave_max_f(v[3 + k]); }
        if ((threadIdx.x & 63) == 0) for (int k = 0; k < 6; k++) subs[6 * ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) + k] = v[k];
    }
    block_minmax<KNN_BOX>(p.x, p.y, p.z, have, boxes + 6 * (size_t)blockIdx.x);
}

__device__ __forceinline__ void update3(float px, float py, float pz, float qx, float qy, float qz, float& b0, float& b1, float& b2)
{
    // updateKBest<3> (simple_knn.cu:132-145): insertion into the ascending triple
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    float d = dx * dx + dy * dy + dz * dz;
    if (b0 > d) { const float t = b0; b0 = d; d = t; }
    if (b1 > d) { const float t = b1; b1 = d; d = t; }
    if (b2 > d) { b2 = d; }
}

// distBoxPoint (simple_knn.cu:119-130)
__device__ __forceinline__ float dist_box_point(const float* __restrict__ bx, float px, float py, float pz)
{
    const float mnx = bx[0], mny = bx[1], mnz = bx[2], mxx = bx[3], mxy = bx[4], mxz = bx[5];   // wave-uniform: scalar loads
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (px < mnx || px > mxx) dx = fminf(fabsf(px - mnx), fabsf(px - mxx));
    if (py < mny || py > mxy) dy = fminf(fabsf(py - mny), fabsf(py - mxy));
    if (pz < mnz || pz > mxz) dz = fminf(fabsf(pz - mnz), fabsf(pz - mxz));
    return dx * dx + dy * dy + dz * dz;
}

// boxMeanDist (simple_knn.cu:147-183).  One thread per point in Morton order; a wave is one 64-point group.  The
// reference seeds a rejection radius from the 6 Morton neighbours and then scans every box it cannot reject; here the
// wave first scans its own group and the two next to it (192 points, which already holds most true neighbours), then
// applies the same conservative AABB test at two levels -- 1024-point boxes, then their 64-point groups -- against the
// running third-best distance.  Every point is visited at most once, so the triple is the exact 3-NN set.
__global__ __launch_bounds__(256) void k_knn_meandist(int P, const float4* __restrict__ sorted, const float* __restrict__ boxes,
                                                      const float* __restrict__ subs, int nbox, float* __restrict__ out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool have = idx < P;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) me = sorted[idx];
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    const int g_own = __builtin_amdgcn_readfirstlane(idx >> 6);          // this wave's group
    const int nsub = (P + 63) >> 6;
    const int g_lo = max(0, g_own - 1), g_hi = min(nsub - 1, g_own + 1);
    {
        const int i0 = g_lo << 6, i1 = min(P, (g_hi + 1) << 6);
        for (int i = i0; i < i1; i++) {
            const float4 q = sorted[i];                                  // wave-uniform address
            if (have && i != idx) update3(me.x, me.y, me.z, q.x, q.y, q.z, b0, b1, b2);
        }
    }
    for (int b = 0; b < nbox; b++) {
        const bool want = have && !(dist_box_point(boxes + 6 * (size_t)b, me.x, me.y, me.z) > b2);
        if (__builtin_amdgcn_ballot_w64(want) == 0) continue;            // nobody in this wave needs the box
        const int s0 = b * 16, s1 = min(nsub, s0 + 16);
        for (int g = s0; g < s1; g++) {
            if (g >= g_lo && g <= g_hi) continue;                        // already visited
            const bool wg = want && !(dist_box_point(subs + 6 * (size_t)g, me.x, me.y, me.z) > b2);
            if (__builtin_amdgcn_ballot_w64(wg) == 0) continue;
            const int i0 = g << 6, i1 = min(P, i0 + 64);
            for (int i = i0; i < i1; i++) {
                const float4 q = sorted[i];
                if (wg) update3(me.x, me.y, me.z, q.x, q.y, q.z, b0, b1, b2);
            }
        }
    }
    if (have) out[__float_as_uint(me.w)] = (b0 + b1 + b2) / 3.0f;
}

// ---------------------------------------------------------------------------------------------
// K nearest neighbours of every point among the same points, the point itself included (distance 0, first) -- what the
// reference asks of pytorch3d: knn_points(points[None], points[None], K) at tetgs_scene/tetgs_model.py:36 (K = 4, scale
// initialisation) and :180 (K = knn_to_track = 16, neighbour tracking).  Same Morton order / box / group rejection as
// k_knn_meandist; the K best (squared distance, index) pairs are kept ascending in registers.
// ---------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void insert_k(float d, uint32_t id, float (&bd)[K], uint32_t (&bi)[K])
{
    if (!(d < bd[K - 1])) return;
#pragma unroll
    for (int j = 0; j < K; j++) {
        if (d < bd[j]) { const float td = bd[j]; const uint32_t ti = bi[j]; bd[j] = d; bi[j] = id; d = td; id = ti; }
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_knn_kbest(int P, int Kout, const float4* __restrict__ sorted, const float* __restrict__ boxes,
                                                   const float* __restrict__ subs, int nbox, float* __restrict__ dists, long long* __restrict__ idx_out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool have = idx < P;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) me = sorted[idx];
    float bd[K];
    uint32_t bi[K];
#pragma unroll
    for (int j = 0; j < K; j++) { bd[j] = FLT_MAX; bi[j] = 0xffffffffu; }
    const int g_own = __builtin_amdgcn_readfirstlane(idx >> 6);
    const int nsub = (P + 63) >> 6;
    const int g_lo = max(0, g_own - 1), g_hi = min(nsub - 1, g_own + 1);
    auto visit = [&](int i0, int i1, bool want) {
        for (int i = i0; i < i1; i++) {
            const float4 q = sorted[i];                                  // wave-uniform address
            const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
            if (want) insert_k<K>(dx * dx + dy * dy + dz * dz, __float_as_uint(q.w), bd, bi);
        }
    };
    visit(g_lo << 6, min(P, (g_hi + 1) << 6), have);
    for (int b = 0; b < nbox; b++) {
        const bool want = have && !(dist_box_point(boxes + 6 * (size_t)b, me.x, me.y, me.z) > bd[K - 1]);
        if (__builtin_amdgcn_ballot_w64(want) == 0) continue;
        const int s0 = b * 16, s1 = min(nsub, s0 + 16);
        for (int g = s0; g < s1; g++) {
            if (g >= g_lo && g <= g_hi) continue;
            const bool wg = want && !(dist_box_point(subs + 6 * (size_t)g, me.x, me.y, me.z) > bd[K - 1]);
            if (__builtin_amdgcn_ballot_w64(wg) == 0) continue;
            visit(g << 6, min(P, (g << 6) + 64), wg);
        }
    }
    if (have) {
        const size_t o = (size_t)__float_as_uint(me.w) * Kout;
#pragma unroll
        for (int j = 0; j < K; j++)
            if (j < Kout) { dists[o + j] = bd[j]; idx_out[o + j] = bi[j] == 0xffffffffu ? -1ll : (long long)bi[j]; }
    }
}

}  // namespace tgs

extern "C" {
#include "../../include/tgs_raster.h"

size_t tgs_dist2_workspace_bytes(int P)
{
    tgs::KnnWork w;
    return tgs::knn_carve(w, nullptr, (size_t)(P > 0 ? P : 0));
}

// Morton order, boxes and groups of `points` into the workspace (shared by tgs_dist2 and tgs_knn_self)
static int knn_prepare(hipStream_t st, int P, const float* points, void* workspace, size_t workspace_bytes, tgs::KnnWork& w, int& nbox)
{
    using namespace tgs;
    if (knn_carve(w, (char*)workspace, (size_t)P) > workspace_bytes) return set_error(TGS_ERR_INVALID, "k-NN workspace smaller than tgs_dist2_workspace_bytes(P)");
    const int nblk = (P + 255) / 256;
    nbox = (P + KNN_BOX - 1) / KNN_BOX;
    hipLaunchKernelGGL(k_knn_minmax, dim3(nblk), dim3(256), 0, st, P, points, w.partial);
    hipLaunchKernelGGL(k_knn_minmax_final, dim3(1), dim3(256), 0, st, nblk, w.partial, w.minmax);
    hipLaunchKernelGGL(k_knn_morton, dim3(nblk), dim3(256), 0, st, P, points, w.minmax, w.keys);
    // sort by (morton, index)
    const uint32_t n = (uint32_t)P, cap = SORT_LDS_CAP;
    uint32_t npad = 1; while (npad < n) npad <<= 1;
    const dim3 lgrid((npad + cap - 1) / cap), ggrid((npad / 2 + 255) / 256);
    hipLaunchKernelGGL(k_sort64_local, lgrid, dim3(256), (size_t)cap * 8, st, w.keys, n, 0u, cap);
    for (uint32_t k = cap * 2; k <= npad; k <<= 1) {
        hipLaunchKernelGGL(k_sort64_global, ggrid, dim3(256), 0, st, w.keys, n, npad, k, 0u, 1);
        for (uint32_t j = k >> 2; j >= cap; j >>= 1) hipLaunchKernelGGL(k_sort64_global, ggrid, dim3(256), 0, st, w.keys, n, npad, k, j, 0);
        hipLaunchKernelGGL(k_sort64_local, lgrid, dim3(256), (size_t)cap * 8, st, w.keys, n, k, cap);
    }
    hipLaunchKernelGGL(k_knn_gather, dim3(nblk), dim3(256), 0, st, P, points, w.keys, w.sorted);
    hipLaunchKernelGGL(k_knn_boxes, dim3(nbox), dim3(KNN_BOX), 0, st, P, w.sorted, w.boxes, w.subs);
    return TGS_OK;
}

int tgs_dist2(void* stream, int P, const float* points, float* mean_dist2, void* workspace, size_t workspace_bytes)
{
    using namespace tgs;
    hipStream_t st = (hipStream_t)stream;
    if (P == 0) return TGS_OK;
    if (P < 0 || !points || !mean_dist2 || !workspace) return set_error(TGS_ERR_INVALID, "tgs_dist2: P >= 0 and non-NULL points / mean_dist2 / workspace required");
    KnnWork w;
    int nbox = 0;
    const int r = knn_prepare(st, P, points, workspace, workspace_bytes, w, nbox);
    if (r < 0) return r;
    hipLaunchKernelGGL(k_knn_meandist, dim3((P + 255) / 256), dim3(256), 0, st, P, w.sorted, w.boxes, w.subs, nbox, mean_dist2);
    return hip_status("tgs_dist2");
}

int tgs_knn_self(void* stream, int P, int K, const float* points, float* dists, long long* idx, void* workspace, size_t workspace_bytes)
{
    using namespace tgs;
    hipStream_t st = (hipStream_t)stream;
    if (P == 0 || K == 0) return TGS_OK;
    if (P < 0 || K < 0 || K > 32 || !points || !dists || !idx || !workspace)
        return set_error(TGS_ERR_INVALID, "tgs_knn_self: P >= 0, 0 <= K <= 32 and non-NULL points / dists / idx / workspace required");
    KnnWork w;
    int nbox = 0;
    const int r = knn_prepare(st, P, points, workspace, workspace_bytes, w, nbox);
    if (r < 0) return r;
    const dim3 grid((P + 255) / 256), blk(256);
    if (K <= 4) hipLaunchKernelGGL((k_knn_kbest<4>), grid, blk, 0, st, P, K, w.sorted, w.boxes, w.subs, nbox, dists, idx);
    else if (K <= 8) hipLaunchKernelGGL((k_knn_kbest<8>), grid, blk, 0, st, P, K, w.sorted, w.boxes, w.subs, nbox, dists, idx);
    else if (K <= 16) hipLaunchKernelGGL((k_knn_kbest<16>), grid, blk, 0, st, P, K, w.sorted, w.boxes, w.subs, nbox, dists, idx);
    else hipLaunchKernelGGL((k_knn_kbest<32>), grid, blk, 0, st, P, K, w.sorted, w.boxes, w.subs, nbox, dists, idx);
    return hip_status("tgs_knn_self");
}
}
