// driver.cpp -- host pipeline of the cross-check harness (see cuda_emu.h for status and purpose).
// Restates Rasterizer::forward/backward (cuda_rasterizer/rasterizer_impl.cu:198-434) around the
// reference's kernel text, which build.sh extracts by line range into /tmp and which is included
// below as ref_fwd.inc / ref_bwd.inc / ref_impl.inc.  cub::DeviceScan::InclusiveSum ->
// std::partial_sum, cub::DeviceRadixSort::SortPairs (stable) -> std::stable_sort on the u64 key.
#include "cuda_emu.h"
#include <glm/glm.hpp>
#include <algorithm>
#include <cstring>
#include <numeric>
thread_local EmuCtx emu_ctx;

#include "auxiliary.h"   // from the reference, via -I
namespace FWD {
#include "ref_fwd.inc"
}
namespace BWD {
#include "ref_bwd.inc"
}
namespace IMPL {
#include "ref_impl.inc"
}

extern "C" int emu_run(
    int P, int D, int M, const float* bg, int W, int H, const float* means3D, const float* shs,
    const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
    const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
    const float* cam_pos, float tan_fovx, float tan_fovy,
    // forward outputs
    float* out_color, int* radii, uint32_t* n_contrib, float* final_T, float* means2D, float* depths,
    float* conic_opacity, float* rgb, uint32_t* tiles_touched, uint32_t* ranges /*2T*/,
    uint32_t* point_list, long long point_list_cap,
    // backward (skipped when dL_dpix == NULL)
    const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    dim3 tile_grid((W + BLOCK_X - 1) / BLOCK_X, (H + BLOCK_Y - 1) / BLOCK_Y, 1);
    dim3 block(BLOCK_X, BLOCK_Y, 1);
    const size_t T = (size_t)tile_grid.x * tile_grid.y;
    std::vector<char> clamped((size_t)P * 3 + 1, 0);
    std::vector<float> cov3D((size_t)P * 6 + 1, 0.f);
    std::vector<uint32_t> offsets((size_t)P + 1, 0);
    dim3 pgrid((P + 255) / 256), pblock(256);

    emu_launch_serial(pgrid, pblock, [&]() {
        FWD::preprocessCUDA<NUM_CHANNELS>(P, D, M, means3D, (const glm::vec3*)scales, scale_modifier,
            (const glm::vec4*)rotations, opacities, shs, (bool*)clamped.data(), cov3D_precomp, colors_precomp,
            viewmatrix, projmatrix, (const glm::vec3*)cam_pos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, radii,
            (float2*)means2D, depths, cov3D.data(), rgb, (float4*)conic_opacity, tile_grid, tiles_touched, false);
    });
    std::partial_sum(tiles_touched, tiles_touched + P, offsets.begin());
    const long long R = P > 0 ? (long long)offsets[P - 1] : 0;
    if (R > point_list_cap) return -1;
    std::vector<uint64_t> keys_unsorted((size_t)R + 1), keys((size_t)R + 1);
    std::vector<uint32_t> vals_unsorted((size_t)R + 1);
    emu_launch_serial(pgrid, pblock, [&]() {
        IMPL::duplicateWithKeys(P, (const float2*)means2D, depths, offsets.data(), keys_unsorted.data(),
                                vals_unsorted.data(), radii, tile_grid);
    });
    {
        std::vector<uint32_t> perm((size_t)R);
        std::iota(perm.begin(), perm.end(), 0u);
        const int bit = (int)IMPL::getHigherMsb(tile_grid.x * tile_grid.y);
        const uint64_t mask = (32 + bit >= 64) ? ~0ull : ((1ull << (32 + bit)) - 1);
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
            return (keys_unsorted[a] & mask) < (keys_unsorted[b] & mask); });
        for (long long i = 0; i < R; i++) { keys[i] = keys_unsorted[perm[i]]; point_list[i] = vals_unsorted[perm[i]]; }
    }
    std::memset(ranges, 0, T * 8);
    if (R > 0)
        emu_launch_serial(dim3((unsigned)((R + 255) / 256)), pblock, [&]() {
            IMPL::identifyTileRanges((int)R, keys.data(), (uint2*)ranges);
        });
    const float* feature_ptr = colors_precomp != nullptr ? colors_precomp : rgb;
    emu_launch_blocks(tile_grid, block, [&]() {
        FWD::renderCUDA<NUM_CHANNELS>((const uint2*)ranges, point_list, W, H, (const float2*)means2D, feature_ptr,
                                      (const float4*)conic_opacity, final_T, n_contrib, bg, out_color);
    });
    if (!dL_dpix) return (int)R;

    emu_launch_blocks(tile_grid, block, [&]() {
        BWD::renderCUDA<NUM_CHANNELS>((const uint2*)ranges, point_list, W, H, bg, (const float2*)means2D,
                                      (const float4*)conic_opacity, feature_ptr, final_T, n_contrib, dL_dpix,
                                      (float3*)dL_dmean2D, (float4*)dL_dconic, dL_dopacity, dL_dcolor);
    });
    const float* cov3D_ptr = cov3D_precomp != nullptr ? cov3D_precomp : cov3D.data();
    emu_launch_serial(pgrid, pblock, [&]() {
        BWD::computeCov2DCUDA(P, (const float3*)means3D, radii, cov3D_ptr, focal_x, focal_y, tan_fovx, tan_fovy,
                              viewmatrix, dL_dconic, (float3*)dL_dmean3D, dL_dcov3D);
    });
    emu_launch_serial(pgrid, pblock, [&]() {
        BWD::preprocessCUDA<NUM_CHANNELS>(P, D, M, (const float3*)means3D, radii, shs, (const bool*)clamped.data(),
            (const glm::vec3*)scales, (const glm::vec4*)rotations, scale_modifier, projmatrix, (const glm::vec3*)cam_pos,
            (const float3*)dL_dmean2D, (glm::vec3*)dL_dmean3D, dL_dcolor, dL_dcov3D, dL_dsh, (glm::vec3*)dL_dscale,
            (glm::vec4*)dL_drot);
    });
    return (int)R;
}
