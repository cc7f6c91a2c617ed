// tgs_shcolor.hip -- fused SH -> RGB of the Gaussian colours OUTSIDE the rasterizer ("next" row 1 of SURVEY.md 8f).
//
// Replaces what every training step of the reference does in ~25 PyTorch element-wise kernels over [P,3,16]
// (Edit_core/tetgs_scene/tetgs_model.py:413-442  get_points_rgb:
//      dirs = normalize(positions - camera_center);  colors = clamp_min(eval_sh(levels-1, sh, dirs) + 0.5, 0)
//  with eval_sh of Edit_core/utils/spherical_harmonics.py:117-172) and its autograd.
// One thread per Gaussian; M = 16 rows (192 B) are staged through LDS so global memory sees coalesced 16-B accesses.
#include "tgs_device.hpp"

namespace tgs {

// basis[k] = d colour / d sh[k] (same polynomials as cuda_rasterizer/forward.cu:20-71), levels = degree + 1
__device__ __forceinline__ void sh_basis(int levels, float x, float y, float z, float (&bs)[16])
{
#pragma unroll
    for (int k = 0; k < 16; k++) bs[k] = 0.f;
    bs[0] = SH_C0;
    if (levels > 1) {
        bs[1] = -SH_C1 * y; bs[2] = SH_C1 * z; bs[3] = -SH_C1 * x;
        if (levels > 2) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            bs[4] = SH_C2_0 * xy; bs[5] = SH_C2_1 * yz; bs[6] = SH_C2_2 * (2.f * zz - xx - yy); bs[7] = SH_C2_3 * xz; bs[8] = SH_C2_4 * (xx - yy);
            if (levels > 3) {
                bs[9] = SH_C3_0 * y * (3.f * xx - yy); bs[10] = SH_C3_1 * xy * z; bs[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
                bs[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy); bs[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
                bs[14] = SH_C3_5 * z * (xx - yy); bs[15] = SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

// (gx, gy, gz) = sum_c dRGB[c] * d colour_c / d (x, y, z): the direction polynomials of cuda_rasterizer/backward.cu:60-123
__device__ __forceinline__ void sh_direction_gradient(int levels, const float (&shv)[48], float x, float y, float z, const float (&dRGB)[3], float& gx, float& gy, float& gz)
{
    gx = gy = gz = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define SH(k) shv[3 * (k) + c]
        float dx_ = 0.f, dy_ = 0.f, dz_ = 0.f;
        if (levels > 1) {
            dx_ = -SH_C1 * SH(3); dy_ = -SH_C1 * SH(1); dz_ = SH_C1 * SH(2);
            if (levels > 2) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dx_ += SH_C2_0 * y * SH(4) + SH_C2_2 * 2.f * -x * SH(6) + SH_C2_3 * z * SH(7) + SH_C2_4 * 2.f * x * SH(8);
                dy_ += SH_C2_0 * x * SH(4) + SH_C2_1 * z * SH(5) + SH_C2_2 * 2.f * -y * SH(6) + SH_C2_4 * 2.f * -y * SH(8);
                dz_ += SH_C2_1 * y * SH(5) + SH_C2_2 * 2.f * 2.f * z * SH(6) + SH_C2_3 * x * SH(7);
                if (levels > 3) {
                    dx_ += SH_C3_0 * SH(9) * 3.f * 2.f * xy + SH_C3_1 * SH(10) * yz + SH_C3_2 * SH(11) * -2.f * xy + SH_C3_3 * SH(12) * -3.f * 2.f * xz +
                           SH_C3_4 * SH(13) * (-3.f * xx + 4.f * zz - yy) + SH_C3_5 * SH(14) * 2.f * xz + SH_C3_6 * SH(15) * 3.f * (xx - yy);
                    dy_ += SH_C3_0 * SH(9) * 3.f * (xx - yy) + SH_C3_1 * SH(10) * xz + SH_C3_2 * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                           SH_C3_3 * SH(12) * -3.f * 2.f * yz + SH_C3_4 * SH(13) * -2.f * xy + SH_C3_5 * SH(14) * -2.f * yz + SH_C3_6 * SH(15) * -3.f * 2.f * xy;
                    dz_ += SH_C3_1 * SH(10) * xy + SH_C3_2 * SH(11) * 4.f * 2.f * yz + SH_C3_3 * SH(12) * 3.f * (2.f * zz - xx - yy) +
                           SH_C3_4 * SH(13) * 4.f * 2.f * xz + SH_C3_5 * SH(14) * (xx - yy);
                }
            }
        }
#undef SH
        gx += dx_ * dRGB[c]; gy += dy_ * dRGB[c]; gz += dz_ * dRGB[c];
    }
}

struct ShArgs {
    int P, M, levels;
    const float* sh;          // [P, M, 3]
    const float* positions;   // [P, 3] (camera mode) or NULL
    const float* camera;      // [3] device (camera mode) or NULL
    const float* directions;  // [P, 3] (direction mode) or NULL
    float* colors;            // fwd out [P, 3]
    const float* dL_dcolors;  // bwd in
    float* dL_dsh;            // bwd out [P, M, 3] (every element written)
    float* dL_dpositions;     // bwd out [P, 3] or NULL
    float* dL_ddirections;    // bwd out [P, 3] or NULL
};

template <bool BWD>
__global__ __launch_bounds__(PRE_BLOCK) void k_sh_rgb(const ShArgs a)
{
    __shared__ float4 sh_lds[PRE_BLOCK * 12];
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const bool in_range = idx < a.P;
    const int ncoef = a.levels * a.levels;
    const bool staged = a.M == 16;
    const size_t base4 = (size_t)blockIdx.x * PRE_BLOCK * 12, total4 = (size_t)a.P * 12;
    float camx = 0.f, camy = 0.f, camz = 0.f;
    if (a.camera) { camx = a.camera[0]; camy = a.camera[1]; camz = a.camera[2]; }
    const size_t i3 = 3 * (size_t)(in_range ? idx : 0);     // the Gaussian's position / direction: asked for with the rows, not behind the barrier
    const float* pd = a.positions ? a.positions : a.directions;
    const float pd0 = pd[i3], pd1 = pd[i3 + 1], pd2 = pd[i3 + 2];
    float up0 = 0.f, up1 = 0.f, up2 = 0.f;
    if (BWD) { up0 = a.dL_dcolors[i3]; up1 = a.dL_dcolors[i3 + 1]; up2 = a.dL_dcolors[i3 + 2]; }
    if (staged) {
        const float4* s4 = reinterpret_cast<const float4*>(a.sh);
        float4 t[12];                                       // registers first: the twelve loads in flight together (see k_sh_rgb_dcrest)
#pragma unroll
        for (int q = 0; q < 12; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; t[q] = s4[i < total4 ? i : total4 - 1]; }
#pragma unroll
        for (int q = 0; q < 12; q++) sh_lds[q * PRE_BLOCK + threadIdx.x] = t[q];
        __syncthreads();
    }
    float shv[48];
#pragma unroll
    for (int q = 0; q < 48; q++) shv[q] = 0.f;
    float x = 0.f, y = 0.f, z = 1.f, vx = 0.f, vy = 0.f, vz = 0.f, inv_len = 0.f;
    if (in_range) {
        if (staged) {
#pragma unroll
            for (int q = 0; q < 12; q++) {
                if (q * 4 < ncoef * 3) { const float4 t = sh_lds[threadIdx.x * 12 + q]; shv[4 * q] = t.x; shv[4 * q + 1] = t.y; shv[4 * q + 2] = t.z; shv[4 * q + 3] = t.w; }
            }
        } else {
            const float* sh = a.sh + (size_t)idx * a.M * 3;
#pragma unroll
            for (int q = 0; q < 48; q++) if (q < ncoef * 3) shv[q] = sh[q];
        }
        if (a.positions) {      // torch.nn.functional.normalize: v / max(|v|, 1e-12)
            vx = pd0 - camx; vy = pd1 - camy; vz = pd2 - camz;
            inv_len = 1.0f / fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
            x = vx * inv_len; y = vy * inv_len; z = vz * inv_len;
        } else {
            x = pd0; y = pd1; z = pd2;
        }
    }
    float bs[16];
    sh_basis(a.levels, x, y, z, bs);
    float res[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < ncoef) { res[0] += bs[k] * shv[3 * k]; res[1] += bs[k] * shv[3 * k + 1]; res[2] += bs[k] * shv[3 * k + 2]; }
    }
    if (!BWD) {
        if (in_range) {
#pragma unroll
            for (int c = 0; c < 3; c++) a.colors[3 * (size_t)idx + c] = fmaxf(res[c] + 0.5f, 0.f);
        }
        return;
    }
    // ---- backward ----
    float dRGB[3] = {0.f, 0.f, 0.f};
    if (in_range) {
#pragma unroll
        for (int c = 0; c < 3; c++) dRGB[c] = (res[c] + 0.5f >= 0.f) ? (c == 0 ? up0 : c == 1 ? up1 : up2) : 0.f;   // clamp_min passes the gradient where x >= min
    }
    if (in_range && (a.dL_dpositions || a.dL_ddirections)) {
        // d colour / d direction (cuda_rasterizer/backward.cu:60-123 polynomials), then through the normalisation
        float gx = 0.f, gy = 0.f, gz = 0.f;
        sh_direction_gradient(a.levels, shv, x, y, z, dRGB, gx, gy, gz);
        if (a.dL_ddirections) { a.dL_ddirections[3 * (size_t)idx] = gx; a.dL_ddirections[3 * (size_t)idx + 1] = gy; a.dL_ddirections[3 * (size_t)idx + 2] = gz; }
        if (a.dL_dpositions) {   // d(v/|v|)/dv = (I - d d^T)/|v|   (the max(.,1e-12) branch has zero measure)
            const float dot = x * gx + y * gy + z * gz;
            a.dL_dpositions[3 * (size_t)idx] = (gx - x * dot) * inv_len;
            a.dL_dpositions[3 * (size_t)idx + 1] = (gy - y * dot) * inv_len;
            a.dL_dpositions[3 * (size_t)idx + 2] = (gz - z * dot) * inv_len;
        }
    }
    // dL_dsh[k][c] = basis_k * dRGB[c], zeros above the active levels
    if (staged) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 12; q++) {
            float o[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { const int i = 4 * q + t; o[t] = (i / 3 < ncoef ? bs[i / 3] : 0.f) * dRGB[i % 3]; }
            sh_lds[threadIdx.x * 12 + q] = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
        float4* d4 = reinterpret_cast<float4*>(a.dL_dsh);
#pragma unroll
        for (int q = 0; q < 12; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; if (i < total4) d4[i] = sh_lds[q * PRE_BLOCK + threadIdx.x]; }
    } else if (in_range) {
        float* dsh = a.dL_dsh + (size_t)idx * a.M * 3;
        for (int k = 0; k < a.M; k++) {
            const float ck = (k < 16 && k < ncoef) ? bs[k < 16 ? k : 0] : 0.f;
            dsh[3 * k] = ck * dRGB[0]; dsh[3 * k + 1] = ck * dRGB[1]; dsh[3 * k + 2] = ck * dRGB[2];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same colours from the TWO parameter tensors the reference's models keep (tetgs_model.py:234-239: _sh_coordinates_dc [P,1,3] and
// _sh_coordinates_rest [P,levels_max^2-1,3]) -- without the torch.cat of the `sh_coordinates` property (:268-272) that every step pays in
// front of get_points_rgb: 192 B read + 192 B written per Gaussian forward, and the split of its gradient backward.  Only the ACTIVE rows
// are read (levels^2 - 1 of the Mr rest rows); at levels == 1 -- the inpainting stage, 16 800 of the reference's ~22 800 rasterizer
// iterations -- the rest tensor is not touched at all, forward or backward (its gradient is exactly zero: the caller gets None / keeps
// its buffer).  For 1 < levels the backward writes all Mr rows of dL_drest (zeros above the active levels: autograd needs a dense tensor).
// Block-contiguous staging through LDS when every rest row is needed (levels^2 - 1 == Mr): the block's 256 * Mr * 12 B are 16-B aligned.
struct ShDcRestArgs {
    int P, Mr, levels;
    const float* dc;          // [P, 3]
    const float* rest;        // [P, Mr, 3] or NULL (Mr == 0 or levels == 1)
    const float* positions; const float* camera; const float* directions;
    float* colors;
    const float* dL_dcolors;
    float* dL_ddc;            // [P, 3]
    float* dL_drest;          // [P, Mr, 3] or NULL (levels == 1: untouched)
    float* dL_dpositions; float* dL_ddirections;
};

template <bool BWD>
__global__ __launch_bounds__(PRE_BLOCK) void k_sh_rgb_dcrest(const ShDcRestArgs a)
{
    extern __shared__ float4 rest_lds[];                    // PRE_BLOCK * Mr * 3 floats when staged
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const bool in_range = idx < a.P;
    const int ncoef = a.levels * a.levels, nrest = ncoef - 1;
    const bool use_rest = a.rest != nullptr && nrest > 0;
    const bool staged = use_rest && nrest == a.Mr;          // every row of the block is needed: coalesced 16-B traffic through LDS
    const size_t blk_f4 = (size_t)PRE_BLOCK * a.Mr * 3 / 4, base4 = (size_t)blockIdx.x * blk_f4, total4 = ((size_t)a.P * a.Mr * 3 + 3) / 4;
    float camx = 0.f, camy = 0.f, camz = 0.f;
    if (a.camera) { camx = a.camera[0]; camy = a.camera[1]; camz = a.camera[2]; }
    // a last partial float4 would read past the tensor, and a row slice such as rest[1:] (Mr = 15: 180 B in) is not 16-byte aligned: such
    // shapes / views take the direct path
    const bool tail_ok = ((size_t)a.P * a.Mr * 3) % 4 == 0 && ((reinterpret_cast<uintptr_t>(a.rest) | reinterpret_cast<uintptr_t>(a.dL_drest)) & 15) == 0;
    const bool do_stage = staged && tail_ok;
    // the Gaussian's own 12-B inputs are asked for before the rows are staged (index clamped), not behind the barrier one after the other
    const size_t i3 = 3 * (size_t)(in_range ? idx : 0);
    const float dc0 = a.dc[i3], dc1 = a.dc[i3 + 1], dc2 = a.dc[i3 + 2];
    const float* pd = a.positions ? a.positions : a.directions;
    const float pd0 = pd[i3], pd1 = pd[i3 + 1], pd2 = pd[i3 + 2];
    float up0 = 0.f, up1 = 0.f, up2 = 0.f;
    if (BWD) { up0 = a.dL_dcolors[i3]; up1 = a.dL_dcolors[i3 + 1]; up2 = a.dL_dcolors[i3 + 2]; }
    asm volatile("" ::: "memory");
    if (do_stage) {
        const float4* s4 = reinterpret_cast<const float4*>(a.rest);
        // six pieces per thread and trip with their loads in flight together (clamped indices; a load under `if` that feeds an LDS write is
        // waited for inside its branch: the block's 45 KB came in as a dozen memory round trips in a row)
        for (size_t q0 = threadIdx.x; q0 < blk_f4; q0 += 6 * PRE_BLOCK) {
            float4 t[6];
#pragma unroll
            for (int u = 0; u < 6; u++) { const size_t q = q0 + (size_t)u * PRE_BLOCK, qq = base4 + (q < blk_f4 ? q : q0); t[u] = nt_load4(s4 + (qq < total4 ? qq : total4 - 1)); }
#pragma unroll
            for (int u = 0; u < 6; u++) { const size_t q = q0 + (size_t)u * PRE_BLOCK; if (q < blk_f4 && base4 + q < total4) rest_lds[q] = t[u]; }
        }
        __syncthreads();
    }
    float shv[48];
#pragma unroll
    for (int q = 0; q < 48; q++) shv[q] = 0.f;
    float x = 0.f, y = 0.f, z = 1.f, vx = 0.f, vy = 0.f, vz = 0.f, inv_len = 0.f;
    if (in_range) {
        shv[0] = dc0; shv[1] = dc1; shv[2] = dc2;
        if (use_rest) {
            const float* r = do_stage ? reinterpret_cast<const float*>(rest_lds) + (size_t)threadIdx.x * a.Mr * 3 : a.rest + (size_t)idx * a.Mr * 3;
#pragma unroll
            for (int q = 0; q < 45; q++) if (q < nrest * 3) shv[3 + q] = r[q];
        }
        if (a.positions) {      // torch.nn.functional.normalize: v / max(|v|, 1e-12)
            vx = pd0 - camx; vy = pd1 - camy; vz = pd2 - camz;
            inv_len = 1.0f / fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
            x = vx * inv_len; y = vy * inv_len; z = vz * inv_len;
        } else {
            x = pd0; y = pd1; z = pd2;
        }
    }
    float bs[16];
    sh_basis(a.levels, x, y, z, bs);
    float res[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < ncoef) { res[0] += bs[k] * shv[3 * k]; res[1] += bs[k] * shv[3 * k + 1]; res[2] += bs[k] * shv[3 * k + 2]; }
    }
    if (!BWD) {
        if (in_range) {
#pragma unroll
            for (int c = 0; c < 3; c++) a.colors[3 * (size_t)idx + c] = fmaxf(res[c] + 0.5f, 0.f);
        }
        return;
    }
    float dRGB[3] = {0.f, 0.f, 0.f};
    if (in_range) {
#pragma unroll
        for (int c = 0; c < 3; c++) dRGB[c] = (res[c] + 0.5f >= 0.f) ? (c == 0 ? up0 : c == 1 ? up1 : up2) : 0.f;
        a.dL_ddc[3 * (size_t)idx] = SH_C0 * dRGB[0]; a.dL_ddc[3 * (size_t)idx + 1] = SH_C0 * dRGB[1]; a.dL_ddc[3 * (size_t)idx + 2] = SH_C0 * dRGB[2];
    }
    if (in_range && (a.dL_dpositions || a.dL_ddirections)) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (a.levels > 1) sh_direction_gradient(a.levels, shv, x, y, z, dRGB, gx, gy, gz);
        if (a.dL_ddirections) { a.dL_ddirections[3 * (size_t)idx] = gx; a.dL_ddirections[3 * (size_t)idx + 1] = gy; a.dL_ddirections[3 * (size_t)idx + 2] = gz; }
        if (a.dL_dpositions) {
            const float dot = x * gx + y * gy + z * gz;
            a.dL_dpositions[3 * (size_t)idx] = (gx - x * dot) * inv_len;
            a.dL_dpositions[3 * (size_t)idx + 1] = (gy - y * dot) * inv_len;
            a.dL_dpositions[3 * (size_t)idx + 2] = (gz - z * dot) * inv_len;
        }
    }
    if (!a.dL_drest || a.Mr == 0) return;                   // levels == 1: the rest rows get no gradient and no traffic
    // dL_drest[k-1][c] = basis_k * dRGB[c] (k >= 1), zeros above the active levels; all Mr rows, block-contiguous through LDS
    if (tail_ok) {
        __syncthreads();
        float* o = reinterpret_cast<float*>(rest_lds) + (size_t)threadIdx.x * a.Mr * 3;
        for (int k = 0; k < a.Mr; k++) {
            const float ck = (k + 1 < ncoef && k + 1 < 16) ? bs[(k + 1) & 15] : 0.f;
            o[3 * k] = ck * dRGB[0]; o[3 * k + 1] = ck * dRGB[1]; o[3 * k + 2] = ck * dRGB[2];
        }
        __syncthreads();
        float4* d4 = reinterpret_cast<float4*>(a.dL_drest);
        for (size_t q = threadIdx.x; q < blk_f4; q += PRE_BLOCK) if (base4 + q < total4) nt_store4(d4 + base4 + q, rest_lds[q]);
    } else if (in_range) {
        float* dr = a.dL_drest + (size_t)idx * a.Mr * 3;
        for (int k = 0; k < a.Mr; k++) {
            const float ck = (k + 1 < ncoef && k + 1 < 16) ? bs[(k + 1) & 15] : 0.f;
            dr[3 * k] = ck * dRGB[0]; dr[3 * k + 1] = ck * dRGB[1]; dr[3 * k + 2] = ck * dRGB[2];
        }
    }
}

void launch_sh_rgb_dcrest(hipStream_t st, const ShDcRestArgs& a, bool backward)
{
    const dim3 grid((unsigned)n_blocks((size_t)a.P)), blk(PRE_BLOCK);
    const size_t lds = (size_t)PRE_BLOCK * (a.Mr > 0 ? a.Mr : 1) * 3 * sizeof(float);       // <= 46 KB
    if (lds > 32 * 1024) {
        (void)hipFuncSetAttribute((const void*)k_sh_rgb_dcrest<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_sh_rgb_dcrest<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (backward) hipLaunchKernelGGL((k_sh_rgb_dcrest<true>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((k_sh_rgb_dcrest<false>), grid, blk, lds, st, a);
}

void launch_sh_rgb(hipStream_t st, const ShArgs& a, bool backward)
{
    const dim3 grid((unsigned)n_blocks((size_t)a.P)), blk(PRE_BLOCK);
    if (backward) hipLaunchKernelGGL((k_sh_rgb<true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((k_sh_rgb<false>), grid, blk, 0, st, a);
}

}  // namespace tgs

extern "C" {
#include "../../include/tgs_raster.h"

int tgs_sh_rgb_forward(void* stream, int P, int M, int levels, const float* sh, const float* positions, const float* camera_center,
                       const float* directions, float* colors)
{
    if (P == 0) return TGS_OK;
    if (P < 0 || levels < 1 || levels > 4 || M < levels * levels || M > 16 || !sh || !colors || ((positions && camera_center) == (directions != nullptr)))
        return tgs::set_error(TGS_ERR_INVALID, "tgs_sh_rgb_forward: levels in 1..4, levels^2 <= M <= 16, sh/colors non-NULL, and exactly one of (positions, camera_center) / directions");
    tgs::ShArgs a{};
    a.P = P; a.M = M; a.levels = levels; a.sh = sh; a.positions = positions; a.camera = camera_center; a.directions = directions; a.colors = colors;
    tgs::launch_sh_rgb((hipStream_t)stream, a, false);
    return tgs::hip_status("tgs_sh_rgb_forward");
}

int tgs_sh_rgb_backward(void* stream, int P, int M, int levels, const float* sh, const float* positions, const float* camera_center,
                        const float* directions, const float* dL_dcolors, float* dL_dsh, float* dL_dpositions, float* dL_ddirections)
{
    if (P == 0) return TGS_OK;
    if (P < 0 || levels < 1 || levels > 4 || M < levels * levels || M > 16 || !sh || !dL_dcolors || !dL_dsh ||
        ((positions && camera_center) == (directions != nullptr)))
        return tgs::set_error(TGS_ERR_INVALID, "tgs_sh_rgb_backward: levels in 1..4, levels^2 <= M <= 16, sh/dL_dcolors/dL_dsh non-NULL, and exactly one of (positions, camera_center) / directions");
    tgs::ShArgs a{};
    a.P = P; a.M = M; a.levels = levels; a.sh = sh; a.positions = positions; a.camera = camera_center; a.directions = directions;
    a.dL_dcolors = dL_dcolors; a.dL_dsh = dL_dsh; a.dL_dpositions = positions ? dL_dpositions : nullptr; a.dL_ddirections = directions ? dL_ddirections : nullptr;
    tgs::launch_sh_rgb((hipStream_t)stream, a, true);
    return tgs::hip_status("tgs_sh_rgb_backward");
}

int tgs_sh_rgb_dcrest_forward(void* stream, int P, int M_rest, int levels, const float* sh_dc, const float* sh_rest, const float* positions,
                              const float* camera_center, const float* directions, float* colors)
{
    if (P == 0) return TGS_OK;
    if (P < 0 || levels < 1 || levels > 4 || M_rest < 0 || M_rest > 15 || M_rest < levels * levels - 1 || !sh_dc || (levels > 1 && !sh_rest) || !colors ||
        ((positions && camera_center) == (directions != nullptr)))
        return tgs::set_error(TGS_ERR_INVALID, "tgs_sh_rgb_dcrest_forward: levels in 1..4, levels^2 - 1 <= M_rest <= 15, sh_dc (and sh_rest for levels > 1) / colors non-NULL, and exactly one of (positions, camera_center) / directions");
    tgs::ShDcRestArgs a{};
    a.P = P; a.Mr = M_rest; a.levels = levels; a.dc = sh_dc; a.rest = levels > 1 ? sh_rest : nullptr; a.positions = positions; a.camera = camera_center;
    a.directions = directions; a.colors = colors;
    tgs::launch_sh_rgb_dcrest((hipStream_t)stream, a, false);
    return tgs::hip_status("tgs_sh_rgb_dcrest_forward");
}

int tgs_sh_rgb_dcrest_backward(void* stream, int P, int M_rest, int levels, const float* sh_dc, const float* sh_rest, const float* positions,
                               const float* camera_center, const float* directions, const float* dL_dcolors, float* dL_dsh_dc, float* dL_dsh_rest,
                               float* dL_dpositions, float* dL_ddirections)
{
    if (P == 0) return TGS_OK;
    if (P < 0 || levels < 1 || levels > 4 || M_rest < 0 || M_rest > 15 || M_rest < levels * levels - 1 || !sh_dc || (levels > 1 && (!sh_rest || !dL_dsh_rest)) ||
        !dL_dcolors || !dL_dsh_dc || ((positions && camera_center) == (directions != nullptr)))
        return tgs::set_error(TGS_ERR_INVALID, "tgs_sh_rgb_dcrest_backward: levels in 1..4, levels^2 - 1 <= M_rest <= 15, sh_dc / dL_dcolors / dL_dsh_dc non-NULL (sh_rest and dL_dsh_rest too for levels > 1), and exactly one of (positions, camera_center) / directions");
    tgs::ShDcRestArgs a{};
    a.P = P; a.Mr = M_rest; a.levels = levels; a.dc = sh_dc; a.rest = levels > 1 ? sh_rest : nullptr; a.positions = positions; a.camera = camera_center;
    a.directions = directions; a.dL_dcolors = dL_dcolors; a.dL_ddc = dL_dsh_dc; a.dL_drest = levels > 1 ? dL_dsh_rest : nullptr;
    a.dL_dpositions = positions ? dL_dpositions : nullptr; a.dL_ddirections = directions ? dL_ddirections : nullptr;
    tgs::launch_sh_rgb_dcrest((hipStream_t)stream, a, true);
    return tgs::hip_status("tgs_sh_rgb_dcrest_backward");
}
}
