// tgs_forward.hip -- forward pass kernels for gfx950 (wave64).
//
// Pipeline (what it replaces in cuda_rasterizer/rasterizer_impl.cu:198-336):
//   k_preprocess_fwd   per-Gaussian EWA projection, SH colour, tile rectangle + the mask of its live tiles, per-block sums of
//                      tiles_touched                                                    (forward.cu:155-256)
//   k_bin_count        per-tile instance COUNT: the Gaussians in BIN_WGS contiguous chunks, one fat workgroup each with a
//                      per-tile table in LDS -- no global atomics                       (the tile half of duplicateWithKeys' keys)
//   k_bin_colscan      per tile: exclusive scan of the chunks' counts (-> where chunk w's instances start inside the tile's list)
//   k_scan             exclusive scans: block sums (-> Gaussian offsets) and tile counts (-> ranges)
//                                                                   (cub InclusiveSum, identifyTileRanges)
//   [host reads R, longest list]                                     (the reference's D2H copy, :280-281)
//   k_scatter          every instance goes straight to its tile's segment: the same chunks, LDS cursors   (duplicateWithKeys)
//   k_tile_sort*       per-tile sort by (depth, index) inside LDS; lists longer than       (cub SortPairs)
//                      SORT_LDS_CAP take the k_ovf_* path
//   k_finalize         gather of the per-instance records in sorted order + the 64-bit quadrant mask of each
//                      instance (which 2x2-pixel quadrants of its tile the splat reaches)   (forward.cu:315-321)
//   k_render_fwd       front-to-back compositing, four 256-thread workgroups per 16x16 tile    (forward.cu:261-374)
//
// The reference sorts 64-bit (tile|depth) keys globally with a 6-pass radix sort (12 B x R per pass);
// here the tile is known when an instance is emitted, so only the depth order inside one tile's list
// has to be established, and that happens in LDS.  Ties in depth resolve to ascending Gaussian index
// exactly like the reference's stable sort over index-ordered emission.
#include "tgs_device.hpp"

namespace tgs {

// ---------------------------------------------------------------------------------------------
// k_preprocess_fwd
// ---------------------------------------------------------------------------------------------
// Culling masks.  A splat reaches a pixel with alpha >= 1/255 iff  alpha = o*exp(power) >= 1/255  <=>  -power <= tau,
// tau = ln(255 o), where -power = f(d) = 1/2 (A dx^2 + C dy^2) + B dx dy is the conic's quadratic form in d = pixel - mean.
// The exact per-pixel tests of forward.cu:336-343 stay in the render kernels, so a conservative mask only removes
// work, never a contribution (margins: +0.01 on tau plus 0.1 % on f).  NaNs and non-convex conics keep everything.
// minimum of f over one edge of a rectangle (tile_reachable): a clamped 1-D parabola
__device__ __forceinline__ float conic_min_on_edge(float dfix, float lo, float hi, float Pfix, float Pvar, float B, float mB_over_Pvar)
{
    // minimise 1/2 Pfix dfix^2 + B dfix t + 1/2 Pvar t^2 over t in [lo, hi]; the stationary point is t = -B dfix / Pvar
    const float t = fminf(hi, fmaxf(lo, dfix * mB_over_Pvar));
    return 0.5f * (Pfix * dfix * dfix + Pvar * t * t) + B * dfix * t;
}

// 64-bit mask of the tile's 8x8 grid of 2x2-pixel quadrants (bit 8*R + C: quadrant row R, quadrant column C) that may hold a pixel the
// splat reaches with alpha >= 1/255, i.e. with f(d) <= tau (above).  On a pixel row the pixels with f <= tau are an interval in x -- f is a
// convex parabola in dx for fixed dy: |dx - c(dy)| <= w(dy), c = -B dy / A, w = sqrt(2 A tau - det dy^2) / A -- so 16 square roots give the
// pixel footprint (widened by 0.01 px) without a loop whose trip count differs between lanes.  Round 5: ONE mask per quadrant row from the
// hull of its two pixel rows' intervals (a superset of their union: conservative; a row the footprint misses has a NaN interval and drops out
// of fminf / fmaxf), and the row's terms as fused multiply-adds of per-splat constants: ~270 vector instructions per instance instead of ~500
// (k_finalize is bound by exactly these).  NaNs and non-convex conics keep every quadrant.
__device__ __forceinline__ unsigned long long quadrant_mask(float2 xy, float4 co, uint32_t tx, uint32_t ty)
{
    const float o255 = 255.0f * co.w;
    if (o255 < 0.999f) return 0ull;                     // alpha <= o < 1/255 for every pixel (G <= 1)
    // (culling only: the hardware's 1-ulp log / sqrt / rcp are far inside the margins, and the IEEE-exact forms cost ~10 VALU each)
    const float tau = (fmaxf(__logf(o255), 0.f) + 0.01f) * 1.001f;
    const float A = co.x, B = co.y, Cc = co.z;
    const float det = A * Cc - B * B;
    if (!(det > 0.f && A > 0.f && Cc > 0.f)) return ~0ull;
    const float rA = __builtin_amdgcn_rcpf(A), K = 2.f * A * tau, sl = -B * rA;
    const float ox = xy.x - (float)(tx * TILE), y0 = (float)(ty * TILE) - xy.y;       // mean relative to the tile; first row relative to the mean
    uint32_t keep[2] = {0u, 0u};
#pragma unroll
    for (int R = 0; R < 8; R++) {
        const float dy0 = y0 + (float)(2 * R), dy1 = y0 + (float)(2 * R + 1);
        const float d0 = fmaf(-det * dy0, dy0, K), d1 = fmaf(-det * dy1, dy1, K);      // A^2 w^2 of the two pixel rows; negative: the row misses
        const float w0 = fmaf(__builtin_amdgcn_sqrtf(d0), rA, 0.01f), w1 = fmaf(__builtin_amdgcn_sqrtf(d1), rA, 0.01f);
        const float c0 = fmaf(sl, dy0, ox), c1 = fmaf(sl, dy1, ox);                      // the rows' interval centres, in pixel columns of the tile
        const float lo = fmaxf(fminf(c0 - w0, c1 - w1), 0.f), hi = fminf(fmaxf(c0 + w0, c1 + w1), 15.f);   // (all NaN -> 0 .. 15: NaN keeps the row)
        const int il = (int)ceilf(lo), ih = (int)floorf(hi);
        const uint32_t ql = (uint32_t)il >> 1, qh = (uint32_t)ih >> 1;
        const uint32_t m8 = (2u << qh) - (1u << ql);         // bits ql .. qh (ql <= qh wherever `some` holds)
        const bool some = !(fmaxf(d0, d1) < 0.f) && il <= ih;
        keep[R >> 2] |= (some ? m8 : 0u) << (8 * (R & 3));
    }
    return (unsigned long long)keep[0] | ((unsigned long long)keep[1] << 32);
}
// the 4x4 blocks (bit 4*by + bx) with a live quadrant
__device__ __forceinline__ uint32_t blocks_of_quadrants(unsigned long long qm)
{
    uint32_t m = 0;
#pragma unroll
    for (int by = 0; by < 4; by++) {
        const uint32_t rp = (uint32_t)(qm >> (16 * by)) & 0xffffu;
        uint32_t c = (rp | (rp >> 8)) & 0xffu;          // the block row's two quadrant rows
        c = (c | (c >> 1)) & 0x55u;                     // pairs of quadrant columns
        m |= ((c & 1u) | ((c >> 1) & 2u) | ((c >> 2) & 4u) | ((c >> 3) & 8u)) << (4 * by);
    }
    return m;
}

// Can the splat reach alpha >= 1/255 anywhere in tile (tx, ty)?  Two stages on the hull of the tile's pixel centres: the bounding
// box of the level set f <= tau (|dx| <= sqrt(2 tau C/det), |dy| <= sqrt(2 tau A/det)), then the exact minimum of the convex f
// over the rectangle (0 if the mean is inside, otherwise on one of the 4 edges).  Same margins as quadrant_mask, and the rectangle
// contains every pixel: never "no" where a quadrant says "yes".
__device__ __forceinline__ bool tile_reachable(float2 xy, float4 co, uint32_t tx, uint32_t ty)
{
    const float o255 = 255.0f * co.w;
    if (o255 < 0.999f) return false;
    const float tau = fmaxf(__logf(o255), 0.f) + 0.01f;
    const float A = co.x, B = co.y, Cc = co.z;
    const float det = A * Cc - B * B;
    if (!(det > 0.f && A > 0.f && Cc > 0.f)) return true;              // NaNs and non-convex conics keep everything
    const float tdi = 2.f * tau * __builtin_amdgcn_rcpf(det);
    const float hx = __builtin_amdgcn_sqrtf(tdi * Cc) * 1.001f + 0.01f, hy = __builtin_amdgcn_sqrtf(tdi * A) * 1.001f + 0.01f;
    const float xl = (float)(tx * TILE) - xy.x, xh = xl + (float)(TILE - 1), yl = (float)(ty * TILE) - xy.y, yh = yl + (float)(TILE - 1);
    if (hx < xl || -hx > xh || hy < yl || -hy > yh) return false;       // bounding box of the level set misses the tile
    if (xl <= 0.f && xh >= 0.f && yl <= 0.f && yh >= 0.f) return true;  // the mean is inside
    const float rC = -B * __builtin_amdgcn_rcpf(Cc), rA = -B * __builtin_amdgcn_rcpf(A);
    const float e0 = conic_min_on_edge(xl, yl, yh, A, Cc, B, rC), e1 = conic_min_on_edge(xh, yl, yh, A, Cc, B, rC);
    const float e2 = conic_min_on_edge(yl, xl, xh, Cc, A, B, rA), e3 = conic_min_on_edge(yh, xl, xh, Cc, A, B, rA);
    return !(fminf(fminf(e0, e1), fminf(e2, e3)) > tau * 1.001f);
}

// The live-tile masks of the wave's rectangles of RANK_TILES + 1 .. COOP_TILES tiles, with the (splat, tile) pairs SPREAD OVER THE LANES
// (round 6).  Until round 5 the splat's own lane walked its rectangle tile by tile (tile_reachable, ~50 instructions a tile) while the other
// 63 waited for the largest rectangle of the wave: nothing at config 3 (2.5 tiles per splat: almost every rectangle is <= RANK_TILES), but
// with mesh-bound splats on 5..64 tiles (the trained scenes: tetgs_edit_2d.py:203) k_preprocess_fwd took 46.6 / 62.0 us at splats x 4 / x 8
// against 35.4.  Now: prefix sum of the areas, 64 pairs per step, the owner found by a 6-step search in LDS (as wave_emit_instances does for
// the keys), the answers collected by ONE ballot per step from which every owner cuts its own bits -- no atomics.  Same test, same masks.
// Convergent: every lane of the wave calls this.
struct ReachTab { float4 a[WAVE]; float4 b[WAVE]; uint32_t excl[WAVE]; };      // a = (x, y, conic.x, conic.y)  b = (conic.z, opacity, bits(minx | miny << 16), bits(rect width))
__device__ __forceinline__ unsigned long long wave_live_masks(bool mid, uint32_t area_in, float2 xy, float4 co, uint32_t minx, uint32_t miny, uint32_t rw, ReachTab& tab, int lane)
{
    const uint32_t area = mid ? area_in : 0u;
    const uint32_t incl = wave_iscan_u32(area, lane), total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), excl = incl - area;
    tab.excl[lane] = excl;
    tab.a[lane] = make_float4(xy.x, xy.y, co.x, co.y);
    tab.b[lane] = make_float4(co.z, co.w, __uint_as_float(minx | (miny << 16)), __uint_as_float(rw));
    wave_sync();
    unsigned long long live = 0ull;
    for (uint32_t base = 0; base < total; base += 64) {     // (wave-uniform)
        const uint32_t w = base + (uint32_t)lane;
        bool reach = false;
        if (w < total) {
            uint32_t lo = 0, hi = 63;                       // owner: the last lane whose first pair is <= w
#pragma unroll
            for (int it = 0; it < 6; it++) { const uint32_t m = (lo + hi + 1) >> 1; if (tab.excl[m] <= w) lo = m; else hi = m - 1; }
            const float4 a = tab.a[lo], b = tab.b[lo];
            const uint32_t k = w - tab.excl[lo], rw2 = __float_as_uint(b.w), mm = __float_as_uint(b.z);
            // k / rw2 for k < 64, 1 <= rw2 <= 64: (k + 1/2) / rw2 is at least 1 / 128 from an integer, v_rcp_f32 is good to 1 ulp
            const uint32_t ky = (uint32_t)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)rw2));
            reach = tile_reachable(make_float2(a.x, a.y), make_float4(a.z, a.w, b.x, b.y), (mm & 0xffffu) + (k - ky * rw2), (mm >> 16) + ky);
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(reach);
        // this lane's pairs of the step: [excl, excl + area) cut with [base, base + 64)
        const uint32_t s0 = max(excl, base), s1 = min(excl + area, base + 64u);
        if (s0 < s1) {
            const uint32_t n = s1 - s0;
            const unsigned long long bits = (bal >> (s0 - base)) & (n >= 64u ? ~0ull : (1ull << n) - 1ull);
            live |= bits << (s0 - excl);
        }
    }
    wave_sync();                                            // (the table is reused: the pair kernel's second view)
    return live;
}

// One Gaussian of one view: projection, EWA covariance, SH colour, tile rectangle, the mask of the rectangle's live tiles and
// the 64-B pack line.  Shared by the one-view and the all-views kernel.
// computeColorFromSH (forward.cu:20-71): colour of a Gaussian seen along (dx, dy, dz) = mean - camera position, from its 16 x 3
// coefficients shv (those of the active degree D are read); clampbits: the channels clamped at 0
__device__ __forceinline__ void sh_to_color(int D, const float (&shv)[48], float dx, float dy, float dz, float& col0, float& col1, float& col2, uint32_t& clampbits)
{
#pragma clang fp contract(off)      // un-fused like the rest of the per-Gaussian forward: the oracle's (and the reference's source's) operation order
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    float res[3];
    clampbits = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define SH(k) shv[3 * (k) + c]
        float r = SH_C0 * SH(0);
        if (D > 0) {
            r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2_0 * xy * SH(4) + SH_C2_1 * yz * SH(5) + SH_C2_2 * (2.0f * zz - xx - yy) * SH(6) +
                    SH_C2_3 * xz * SH(7) + SH_C2_4 * (xx - yy) * SH(8);
                if (D > 2) {
                    r = r + SH_C3_0 * y * (3.0f * xx - yy) * SH(9) + SH_C3_1 * xy * z * SH(10) +
                        SH_C3_2 * y * (4.0f * zz - xx - yy) * SH(11) +
                        SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        SH_C3_4 * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3_5 * z * (xx - yy) * SH(14) +
                        SH_C3_6 * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        r += 0.5f;
        if (r < 0.f) clampbits |= 1u << c;
        res[c] = fmaxf(r, 0.0f);
    }
    col0 = res[0]; col1 = res[1]; col2 = res[2];
}

// a view's colour of one Gaussian, evaluated before the per-view stage (sh_colors_half_staged); valid == false: evaluate in place
struct PreColor { bool valid; float c0, c1, c2; uint32_t clampbits; };

// The view-independent inputs of a Gaussian, asked for ONCE at the head of the kernel, all together and for every view the kernel walks: read
// where they are used, the mean is one memory round trip (the frustum test waits for it), scale / rotation / opacity a second one behind
// the test, per view.  (Index clamped, not predicated: the values of a thread past P are never used.)
struct GaussIn { float mx, my, mz, s0, s1, s2, opac; tgs_v4f q; };
template <bool HAS_SCALE_ROT>
__device__ __forceinline__ GaussIn load_gauss_in(const FwdIn& in, int idx)
{
    const size_t ic = idx < in.P ? (size_t)idx : 0;
    GaussIn gi;
    gi.mx = in.means3D[3 * ic]; gi.my = in.means3D[3 * ic + 1]; gi.mz = in.means3D[3 * ic + 2];
    gi.opac = in.opacities[ic];
    gi.s0 = gi.s1 = gi.s2 = 0.f; gi.q = tgs_v4f{0.f, 0.f, 0.f, 0.f};
    if (HAS_SCALE_ROT) {
        gi.s0 = in.scales[3 * ic]; gi.s1 = in.scales[3 * ic + 1]; gi.s2 = in.scales[3 * ic + 2];
        gi.q = reinterpret_cast<const tgs_v4f*>(in.rotations)[ic];
    }
    return gi;
}

template <bool HAS_SH, bool HAS_SCALE_ROT>
__device__ __forceinline__ uint32_t preprocess_fwd_one(const FwdIn& in, int* __restrict__ radii, const CamParams& cam, const GeomState& g, const ImgState& s,
                                                       const float4* sh_lds, bool sh_staged, int idx, const GaussIn& gi, const PreColor& pre, bool& prefilter_violation,
                                                       ReachTab& rtab)
{
#pragma clang fp contract(off)      // projection, covariance, radius and colour un-fused: the oracle's (and the reference's source's) operation order
    uint32_t tiles = 0;
    const ViewMat V = load_mat(cam.view), PM = load_mat(cam.proj);      // uniform -> scalar loads, before any store
    const float camx = cam.campos[0], camy = cam.campos[1], camz = cam.campos[2];
    int my_radius_i = 0;
    uint32_t minx = 0, miny = 0, maxx = 0, maxy = 0;
    // what the rectangle's live mask and the pack line are made of (a Gaussian with a non-empty rectangle: `have`)
    bool have = false, mid = false;
    float pix = 0.f, piy = 0.f, conx = 0.f, cony = 0.f, conz = 0.f, opac = 0.f, col0 = 0.f, col1 = 0.f, col2 = 0.f;
    unsigned long long live_mask = 0ull;
    if (idx < in.P) {
        const float mx = gi.mx, my = gi.my, mz = gi.mz;
        // in_frustum (auxiliary.h:139-164)
        const float* pm = PM.m;
        const float hx = pm[0] * mx + pm[4] * my + pm[8] * mz + pm[12];
        const float hy = pm[1] * mx + pm[5] * my + pm[9] * mz + pm[13];
        const float hw = pm[3] * mx + pm[7] * my + pm[11] * mz + pm[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;
        const float* vm = V.m;
        const float view_z = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
        bool ok = !(view_z <= 0.2f);
        prefilter_violation = !ok && in.prefiltered;      // -> GeomState::block_flags (block_sum_tiles), OR-ed into Meta::error by k_scan
        if (ok) {
            float cov3d[6];
            if (HAS_SCALE_ROT) {
                compute_cov3d(cam.scale_modifier, gi.s0, gi.s1, gi.s2, make_float4(gi.q.x, gi.q.y, gi.q.z, gi.q.w), cov3d);     // (not stored: the backward evaluates it again)
            } else {
#pragma unroll
                for (int i = 0; i < 6; i++) cov3d[i] = in.cov3D_precomp[6 * (size_t)idx + i];
            }
            const Cov2D c2 = compute_cov2d(mx, my, mz, cov3d, cam, V);
            const float cx = c2.cov.m[0][0] + 0.3f, cy = c2.cov.m[0][1], cz = c2.cov.m[1][1] + 0.3f;   // forward.cu:110-111
            const float det = (cx * cz - cy * cy);
            ok = det != 0.0f;
            if (ok) {
                const float det_inv = 1.f / det;
                conx = cz * det_inv; cony = -cy * det_inv; conz = cx * det_inv;
                const float mid_ev = 0.5f * (cx + cz);
                const float lambda1 = mid_ev + sqrtf(fmaxf(0.1f, mid_ev * mid_ev - det));
                const float lambda2 = mid_ev - sqrtf(fmaxf(0.1f, mid_ev * mid_ev - det));
                const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
                pix = ndc2pix(projx, cam.W); piy = ndc2pix(projy, cam.H);
                get_rect(pix, piy, (int)my_radius, cam.gx, cam.gy, minx, miny, maxx, maxy);
                tiles = (maxx - minx) * (maxy - miny);
                if (tiles != 0) {
                    have = true;
                    if (HAS_SH) {
                        if (pre.valid) {                       // colour of this view evaluated in the staging phase (sh_colors_half_staged)
                            col0 = pre.c0; col1 = pre.c1; col2 = pre.c2;
                            g.clamped[idx] = (uint8_t)pre.clampbits;
                        } else {
                            float shv[48];
                            const int ncoef = (in.D + 1) * (in.D + 1);
                            if (sh_staged) {
#pragma unroll
                                for (int q = 0; q < 12; q++) {
                                    if (q * 4 < ncoef * 3) { const float4 t = sh_lds[threadIdx.x * 12 + q]; shv[4 * q] = t.x; shv[4 * q + 1] = t.y; shv[4 * q + 2] = t.z; shv[4 * q + 3] = t.w; }
                                }
                            } else {
                                const float* sh = in.shs + (size_t)idx * in.M * 3;
#pragma unroll
                                for (int q = 0; q < 48; q++) if (q < ncoef * 3) shv[q] = sh[q];
                            }
                            uint32_t clampbits;
                            sh_to_color(in.D, shv, mx - camx, my - camy, mz - camz, col0, col1, col2, clampbits);
                            g.clamped[idx] = (uint8_t)clampbits;
                        }
                    } else if (in.colors_precomp) {
                        col0 = in.colors_precomp[3 * (size_t)idx]; col1 = in.colors_precomp[3 * (size_t)idx + 1]; col2 = in.colors_precomp[3 * (size_t)idx + 2];
                    }
                    g.depth[idx] = view_z;
                    my_radius_i = (int)my_radius;
                    // The 3-sigma square over-covers: a tile of the rectangle where the splat stays below alpha = 1/255 everywhere
                    // (tile_reachable: the conservative test that masks the 4x4 blocks for the render kernels, applied to the whole
                    // tile, so nothing that could be blended is lost) gets no instance at all -- no count, no key, no sort, no record
                    // (about a fifth of them).  The live tiles of a rectangle of <= COOP_TILES tiles are a 64-bit mask, row-major.
                    opac = gi.opac;
                    if (tiles <= (uint32_t)RANK_TILES) {
                        uint32_t kx = 0, ky = 0;
#pragma unroll
                        for (int k = 0; k < RANK_TILES; k++) {
                            if ((uint32_t)k < tiles) {
                                if (!in.prune || tile_reachable(make_float2(pix, piy), make_float4(conx, cony, conz, opac), minx + kx, miny + ky)) live_mask |= 1ull << k;
                                if (++kx == maxx - minx) { kx = 0; ky++; }
                            }
                        }
                        tiles = (uint32_t)__builtin_popcountll(live_mask);
                    } else if (tiles <= (uint32_t)COOP_TILES) {
                        if (in.prune) mid = true;              // (below, with the whole wave)
                        else live_mask = tiles >= 64u ? ~0ull : (1ull << tiles) - 1ull;
                    }
                }
            }
        }
    }
    // rectangles of RANK_TILES + 1 .. COOP_TILES tiles: their (splat, tile) pairs spread over the wave's lanes
    if (__builtin_amdgcn_ballot_w64(mid) != 0ull) {          // (wave-uniform; never taken when every splat of the wave is small)
        const unsigned long long lm = wave_live_masks(mid, tiles, make_float2(pix, piy), make_float4(conx, cony, conz, opac), minx, miny, maxx - minx, rtab, threadIdx.x & 63);
        if (mid) { live_mask = lm; tiles = (uint32_t)__builtin_popcountll(lm); }
    }
    if (have) {
        // the 64-B pack line in one go (.w of the third quarter: slab offset, k_scatter)
        float4* pk = g.pack + 4 * (size_t)idx;
        pk[0] = make_float4(pix, piy, conx, cony);
        pk[1] = make_float4(conz, opac, col0, col1);
        pk[2] = make_float4(col2, __uint_as_float(minx | (miny << 16)), __uint_as_float(maxx | (maxy << 16)), 0.f);
        pk[3] = make_float4(__uint_as_float((uint32_t)live_mask), __uint_as_float((uint32_t)(live_mask >> 32)), 0.f, 0.f);
        g.live[idx] = make_uint2((uint32_t)live_mask, (uint32_t)(live_mask >> 32));
    }
    if (idx < in.P) {
        if (radii) radii[idx] = my_radius_i;
        g.tiles_touched[idx] = tiles;
        g.rect[idx] = make_ushort4((unsigned short)minx, (unsigned short)miny, (unsigned short)maxx, (unsigned short)maxy);
    }
    return tiles;
}

// per-block sum of tiles_touched (first level of the offsets scan) and the block's prefiltered-violation flag; block 0 also clears the
// frame's Meta (nothing else touches Meta before k_scan: the host-side memset in front of every frame is gone)
__device__ __forceinline__ void block_sum_tiles(uint32_t tiles, bool violation, const GeomState& g, const ImgState& s)
{
    __shared__ uint32_t wsum[PRE_BLOCK / WAVE];
    __shared__ uint32_t wflag[PRE_BLOCK / WAVE];
    const uint32_t w = wave_sum_u32(tiles);
    const bool any = __builtin_amdgcn_ballot_w64(violation) != 0ull;
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = w; wflag[threadIdx.x >> 6] = any ? 1u : 0u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        g.block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        g.block_flags[blockIdx.x] = wflag[0] | wflag[1] | wflag[2] | wflag[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < sizeof(Meta) / sizeof(uint32_t)) reinterpret_cast<uint32_t*>(s.meta)[threadIdx.x] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(ScanAux) / sizeof(uint32_t)) reinterpret_cast<uint32_t*>(s.aux)[threadIdx.x] = 0u;
}

// SH rows of the workgroup's 256 Gaussians (M = 16: 192 B each, 48 KB in all) are fetched with fully coalesced
// 16-B-per-lane loads into LDS; a per-thread walk over its own 192-B row would touch 64 lines per instruction.
__device__ __forceinline__ void stage_sh_rows(const FwdIn& in, float4* sh_lds)
{
    const float4* s4 = reinterpret_cast<const float4*>(in.shs);
    const size_t base4 = (size_t)blockIdx.x * PRE_BLOCK * 12, total4 = (size_t)in.P * 12;
    float4 r[12];                                           // registers first, the twelve loads in flight together: a load under `if` that feeds an LDS write is waited for inside its branch
#pragma unroll
    for (int q = 0; q < 12; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; r[q] = nt_load4(&s4[i < total4 ? i : total4 - 1]); }
#pragma unroll
    for (int q = 0; q < 12; q++) sh_lds[q * PRE_BLOCK + threadIdx.x] = r[q];
    __syncthreads();
}

// The colours of the workgroup's 256 Gaussians for NV views, with HALF of their SH rows in LDS at a time (24 KB instead of 48: the
// kernel's occupancy is bound by LDS -- 3 workgroups per CU with the whole set staged, and its time follows the occupancy: 63 us at 2
// workgroups per CU, 52 at 3).  Rows of Gaussians [0, 128) of the block go to LDS (coalesced 16-B-per-lane loads; a per-thread walk
// over its own 192-B row would touch 64 lines per instruction), threads 0..127 evaluate their colour for every view, then the same for
// [128, 256) -- whose rows were fetched into registers before the first half was evaluated.  A Gaussian behind the near plane of a
// view gets no colour (it is never read).
constexpr int SH_HALF = PRE_BLOCK / 2;
template <int NV>
__device__ __forceinline__ void sh_colors_half_staged(const FwdIn& in, const FwdView* __restrict__ views, int idx, const GaussIn& gi, float4* sh_lds, PreColor (&pre)[NV])
{
    const float4* s4 = reinterpret_cast<const float4*>(in.shs);
    const size_t base4 = (size_t)blockIdx.x * PRE_BLOCK * 12, total4 = (size_t)in.P * 12;
    constexpr int PER = SH_HALF * 12 / PRE_BLOCK;            // float4 per thread and half (6)
    float4 r0[PER], r1[PER];
    // (both halves' loads are issued before the first half goes to LDS, indices clamped: the second half's were behind the wait for the first)
#pragma unroll
    for (int q = 0; q < PER; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; r0[q] = nt_load4(&s4[i < total4 ? i : total4 - 1]); }
#pragma unroll
    for (int q = 0; q < PER; q++) { const size_t i = base4 + SH_HALF * 12 + q * PRE_BLOCK + threadIdx.x; r1[q] = nt_load4(&s4[i < total4 ? i : total4 - 1]); }
#pragma unroll
    for (int q = 0; q < PER; q++) sh_lds[q * PRE_BLOCK + threadIdx.x] = r0[q];
    const float mx = gi.mx, my = gi.my, mz = gi.mz;
    auto eval = [&]() {
        float shv[48];
        const int ncoef = (in.D + 1) * (in.D + 1);
        const float4* row = sh_lds + (threadIdx.x & (SH_HALF - 1)) * 12;
#pragma unroll
        for (int q = 0; q < 12; q++) {
            if (q * 4 < ncoef * 3) { const float4 t = row[q]; shv[4 * q] = t.x; shv[4 * q + 1] = t.y; shv[4 * q + 2] = t.z; shv[4 * q + 3] = t.w; }
        }
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const CamParams& cam = views[v].cam;
            const float view_z = cam.view[2] * mx + cam.view[6] * my + cam.view[10] * mz + cam.view[14];
            pre[v].valid = true;
            if (!(view_z <= 0.2f)) sh_to_color(in.D, shv, mx - cam.campos[0], my - cam.campos[1], mz - cam.campos[2], pre[v].c0, pre[v].c1, pre[v].c2, pre[v].clampbits);
        }
    };
#pragma unroll
    for (int v = 0; v < NV; v++) { pre[v].valid = true; pre[v].c0 = pre[v].c1 = pre[v].c2 = 0.f; pre[v].clampbits = 0u; }
    __syncthreads();
    if (threadIdx.x < SH_HALF && idx < in.P) eval();
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; q++) sh_lds[q * PRE_BLOCK + threadIdx.x] = r1[q];
    __syncthreads();
    if (threadIdx.x >= SH_HALF && idx < in.P) eval();
}

template <bool HAS_SH, bool HAS_SCALE_ROT>
__global__ __launch_bounds__(PRE_BLOCK) void k_preprocess_fwd(const FwdIn in, const FwdView vw)
{
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    __shared__ float4 sh_lds[HAS_SH ? SH_HALF * 12 : 1];
    __shared__ ReachTab rtab[PRE_BLOCK / WAVE];
    PreColor pre[1];
    pre[0].valid = false;
    const GaussIn gi = load_gauss_in<HAS_SCALE_ROT>(in, idx);
    if (HAS_SH && in.M == 16) sh_colors_half_staged<1>(in, &vw, idx, gi, sh_lds, pre);
    bool bad = false;
    const uint32_t tiles = preprocess_fwd_one<HAS_SH, HAS_SCALE_ROT>(in, vw.radii, vw.cam, vw.g, vw.s, sh_lds, false, idx, gi, pre[0], bad, rtab[threadIdx.x >> 6]);
    block_sum_tiles(tiles, bad, vw.g, vw.s);
}

// Two views of a batch in one launch (tgs_forward_views, the default group): the 192-B SH row -- more than half of what the stage
// reads -- is fetched once for both.
template <bool HAS_SH, bool HAS_SCALE_ROT>
__global__ __launch_bounds__(PRE_BLOCK) void k_preprocess_fwd_pair(const FwdIn in, const FwdView v0, const FwdView v1)
{
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    __shared__ float4 sh_lds[HAS_SH ? SH_HALF * 12 : 1];
    __shared__ ReachTab rtab[PRE_BLOCK / WAVE];
    const FwdView vws[2] = {v0, v1};
    PreColor pre[2];
    pre[0].valid = pre[1].valid = false;
    const GaussIn gi = load_gauss_in<HAS_SCALE_ROT>(in, idx);
    if (HAS_SH && in.M == 16) sh_colors_half_staged<2>(in, vws, idx, gi, sh_lds, pre);
#pragma unroll
    for (int v = 0; v < 2; v++) {
        const FwdView& vw = v == 0 ? v0 : v1;
        bool bad = false;
        const uint32_t tiles = preprocess_fwd_one<HAS_SH, HAS_SCALE_ROT>(in, vw.radii, vw.cam, vw.g, vw.s, sh_lds, false, idx, gi, pre[v], bad, rtab[threadIdx.x >> 6]);
        block_sum_tiles(tiles, bad, vw.g, vw.s);
        __syncthreads();                                   // wsum is reused by the next view
    }
}

// More than two views of a batch in one launch (tgs_set_forward_group > 2): the whole set of rows staged once, every view then projects,
// colours and counts from it.
template <bool HAS_SH, bool HAS_SCALE_ROT>
__global__ __launch_bounds__(PRE_BLOCK) void k_preprocess_fwd_batch(const FwdIn in, const FwdViews views)
{
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    __shared__ float4 sh_lds[HAS_SH ? PRE_BLOCK * 12 : 1];
    __shared__ ReachTab rtab[PRE_BLOCK / WAVE];
    const bool sh_staged = HAS_SH && in.M == 16;
    const GaussIn gi = load_gauss_in<HAS_SCALE_ROT>(in, idx);
    if (sh_staged) stage_sh_rows(in, sh_lds);
    PreColor pre;
    pre.valid = false;
#pragma unroll 1
    for (int v = 0; v < views.n; v++) {
        const FwdView& vw = views.v[v];
        bool bad = false;
        const uint32_t tiles = preprocess_fwd_one<HAS_SH, HAS_SCALE_ROT>(in, vw.radii, vw.cam, vw.g, vw.s, sh_lds, sh_staged, idx, gi, pre, bad, rtab[threadIdx.x >> 6]);
        block_sum_tiles(tiles, bad, vw.g, vw.s);
        __syncthreads();                                   // wsum is reused by the next view
    }
}

// ---------------------------------------------------------------------------------------------
// k_scan: workgroup 0 scans the per-block sums (-> Gaussian offsets, R); workgroups 1.. take SCAN_THREADS tiles each (-> ranges, tile order,
// descriptors, the frame's counts).
//
// Round 5.  Until round 4 ONE workgroup walked all tiles: scan, length histogram, then a second pass that took every tile's position in
// tile_order with a returning LDS atomic -- 15.3 us at 1920 x 1080 (8160 tiles), a quarter of the binning chain, with 255 CUs idle.  Now
//   * k_bin_colscan, which forms every tile's count anyway, also adds the tiles of its 64 to a global histogram of list-length buckets
//     (ScanAux::hist: one atomic per workgroup and occupied bucket), so the bucket sizes of the WHOLE frame are known when k_scan starts;
//   * a tile workgroup needs nothing from its peers: the instances in front of its first tile it sums itself from tile_count (at most
//     T - 1024 values out of L2), the bucket stretches of tile_order follow from the histogram, and its tiles' places inside a stretch
//     come from ONE returning global atomic per workgroup and bucket (ScanAux::cursor) on top of a local rank (order inside a bucket
//     does not matter);
//   * longest list / overflow tiles go to Meta with global atomics (a few per workgroup); the copy of Meta in pinned host memory is
//     completed by whichever tile workgroup finishes last (ScanAux::done: a count, nobody waits for anybody).
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_ITEMS = 4;

__device__ __forceinline__ unsigned long long block_exscan_u64(unsigned long long v, unsigned long long* lds, unsigned long long& total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long inc = wave_iscan_u64(v, lane);
    if (lane == 63) lds[wv] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / WAVE; i++) { const unsigned long long t = lds[i]; if (i < wv) base += t; tot += t; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}
__device__ __forceinline__ uint32_t length_bucket(uint32_t n) { return n ? 32u - (uint32_t)__builtin_clz(n) : 0u; }

// host_meta (may be NULL): a copy of Meta in pinned, device-visible HOST memory, written by the kernel itself -- the speculative
// forward's read-back without a copy engine or blit kernel in the stream (a D2H blit between k_scan and k_scatter cost 4 us + a 6-us gap)
// tile_bound: the sync-free grids behind the scan cover that many entries of tile_order; a frame with more non-empty tiles is rejected
// like one that exceeds r_capacity (T: no bound)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan(const GeomState g, const ImgState s, uint32_t nblocks, uint32_t T, uint32_t sort_cap, unsigned long long r_capacity, uint32_t tile_bound,
                                                       uint32_t heavy_bound, uint32_t mid_bound, Meta* host_meta, int light)
{
    __shared__ unsigned long long lds[SCAN_THREADS / WAVE];
    __shared__ uint32_t lds_max[SCAN_THREADS / WAVE];
    if (blockIdx.x == 0) {
        unsigned long long carry = 0;
        uint32_t flags = 0;                                 // OR of the blocks' prefiltered-violation flags
        for (uint32_t base = 0; base < nblocks; base += SCAN_THREADS * SCAN_ITEMS) {
            uint32_t v[SCAN_ITEMS];
            unsigned long long sum = 0;
            const uint32_t i0 = base + threadIdx.x * SCAN_ITEMS;
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = (i0 + k < nblocks) ? g.block_sums[i0 + k] : 0u; sum += v[k]; flags |= (i0 + k < nblocks) ? g.block_flags[i0 + k] : 0u; }
            unsigned long long tot;
            unsigned long long ex = block_exscan_u64(sum, lds, tot) + carry;
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; k++) { if (i0 + k < nblocks) g.block_sums[i0 + k] = (uint32_t)ex; ex += v[k]; }
            carry += tot;
        }
        const int any_flag = __syncthreads_or((int)flags);
        if (threadIdx.x == 0) {
            s.meta->R = carry;
            uint32_t err = any_flag ? 1u : 0u;              // bit 0: a prefiltered Gaussian was culled (k_preprocess_fwd* -> block_flags)
            if (carry > r_capacity) err |= META_ERR_CAPACITY;                                                    // tgs_forward_async: the frame does not fit
            if (err) atomicOr(&s.meta->error, err);         // (a tile workgroup may OR META_ERR_CAPACITY concurrently)
            if (host_meta) { host_meta->R = carry; host_meta->error = err; }
        }
        return;
    }
    // ---- SCAN_THREADS tiles, one per thread
    __shared__ uint32_t hist[40], start[40], lcount[40], gbase[40];
    __shared__ uint32_t cls[3];                            // n_nonempty, n_heavy, n_mid of the frame
    const uint32_t w = blockIdx.x - 1u, t0 = w * SCAN_THREADS, t = t0 + threadIdx.x;
    const bool in = t < T;
    // (a) instances in front of this workgroup's first tile: its own sum over tile_count[0, t0) -- w loads per thread, all in flight
    unsigned long long before = 0;
    for (uint32_t k = 0; k < w; k++) before += s.tile_count[k * SCAN_THREADS + threadIdx.x];
    const uint32_t c = in ? s.tile_count[t] : 0u;
    if (threadIdx.x < 40) { hist[threadIdx.x] = threadIdx.x < 34 ? __builtin_nontemporal_load(&s.aux->hist[threadIdx.x]) : 0u; lcount[threadIdx.x] = 0u; }
    __syncthreads();                                        // hist / lcount are published
    // (c) FIRST (round 6: it stood behind the two scans, a memory round trip at the end of the kernel's chain): rank among this workgroup's tiles of the
    // same bucket, then ONE global returning atomic per occupied bucket -- issued here, its answer is only needed where the positions are formed below
    const uint32_t bkt = length_bucket(c);
    const uint32_t lrank = in ? atomicAdd(&lcount[bkt], 1u) : 0u;
    __syncthreads();
    uint32_t gb = 0;
    if (threadIdx.x < 34) { const uint32_t lc = lcount[threadIdx.x]; if (lc) gb = atomicAdd(&s.aux->cursor[threadIdx.x], lc); }
    unsigned long long tot, pre_tot;
    const unsigned long long pre = block_exscan_u64(before, lds, pre_tot);    // (only the total is used)
    (void)pre;
    const unsigned long long ex = block_exscan_u64((unsigned long long)c, lds, tot) + pre_tot;
    if (in) s.ranges[t] = make_uint2((uint32_t)ex, (uint32_t)(ex + c));
    // (b) the frame's bucket stretches: tiles by descending list length (power-of-two buckets; order inside a bucket does not matter)
    if (threadIdx.x < 34) {
        uint32_t acc = 0;
        for (uint32_t b2 = 33; b2 > threadIdx.x; b2--) acc += hist[b2];
        start[threadIdx.x] = acc;
        if (threadIdx.x == 1) cls[0] = acc + hist[1];      // tiles with at least one instance
        if (threadIdx.x == 11) cls[1] = acc + hist[11];    // ... with >= 1024
        if (threadIdx.x == 8) cls[2] = acc + hist[8];      // ... with >= 128
    }
    if (threadIdx.x < 34) gbase[threadIdx.x] = gb;          // (the atomic's answer, asked for above)
    __syncthreads();
    const uint32_t n_nonempty = cls[0], n_heavy = cls[1], n_mid = cls[2];
    if (in) {
        const uint32_t pos = start[bkt] + gbase[bkt] + lrank;
        s.tile_order[pos] = t;
        s.tile_desc[pos] = make_uint4(t, (uint32_t)ex, (uint32_t)(ex + c), 0u);
        if (pos >= n_mid && pos < n_nonempty) s.light_desc[pos - n_mid] = make_uint4(t, (uint32_t)ex, (uint32_t)(ex + c), 0u);   // (always: one 16-B store per tile below 128 entries)
        if (c > sort_cap) s.ovf_tiles[atomicAdd(&s.meta->n_overflow, 1u)] = t;
    }
    uint32_t mx = wave_max_u32(c);
    if ((threadIdx.x & 63) == 0) lds_max[threadIdx.x >> 6] = mx;
    if (w == 0 && threadIdx.x == 0) {                      // the frame's class counts (from the histogram: every tile workgroup knows them, the first one publishes them)
        s.meta->n_nonempty = n_nonempty; s.meta->n_heavy = n_heavy; s.meta->n_mid = n_mid;
        const bool too_many = n_nonempty > tile_bound || n_heavy > heavy_bound || n_mid > mid_bound;   // more tiles (of a class) than the grids behind the scan cover
        if (too_many) atomicOr(&s.meta->error, META_ERR_CAPACITY);
        if (host_meta) { host_meta->pad[0] = too_many ? 1u : 0u; host_meta->n_nonempty = n_nonempty; host_meta->n_heavy = n_heavy; host_meta->n_mid = n_mid; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        for (int i = 0; i < SCAN_THREADS / WAVE; i++) m = lds_max[i] > m ? lds_max[i] : m;
        if (m) atomicMax(&s.meta->max_count, m);
        // the last tile workgroup to get here completes the host's copy (max_count / n_overflow are final once every workgroup has added its share)
        if (host_meta) {
            __threadfence();
            const uint32_t nw = gridDim.x - 1u;
            if (atomicAdd(&s.aux->done, 1u) == nw - 1u) {
                __threadfence();
                host_meta->max_count = atomicMax(&s.meta->max_count, 0u);
                host_meta->n_overflow = atomicAdd(&s.meta->n_overflow, 0u);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Binning without global atomics.  The per-tile counters of a frame would be ~1 M memory-side atomic requests (one per instance, ~18 G/s
// chip-wide whatever the address pattern: half of k_preprocess_fwd's time when it took them).  Instead the Gaussians are cut into
// `nchunks` contiguous chunks, one 1024-thread workgroup each (a multiple of 1024 Gaussians: every thread walks the same number of
// steps), and a chunk keeps the per-tile table in LDS:
//   k_bin_count    table[w][t] = instances of chunk w in tile t                                  (ds_add_u32)
//   k_bin_colscan  per tile: exclusive scan over w in place, total -> tile_count[t]
//   k_scan         ranges[t] from tile_count                                                     (as before)
//   k_scatter      chunk w again: cursor[t] = ranges[t].x + table[w][t] in LDS, every instance takes its position with a returning
//                  ds_add and stores its key there
// The order of a chunk's instances inside a tile is whatever the LDS atomics make it; k_tile_sort orders by (depth, index) anyway.
// Tile grids beyond BIN_LDS_TILES are walked in bands of that many tiles (one more pass over the chunk per band).
// ---------------------------------------------------------------------------------------------
// rectangles of 5..COOP_TILES tiles: the wave's (splat, tile) pairs are spread over its lanes, 64 pairs per step
struct MidTab { uint32_t excl[WAVE]; uint2 rect[WAVE]; unsigned long long live[WAVE]; unsigned long long key[WAVE]; };

// Calls f(tile, key) once for every instance of the wave's 64 Gaussians (lane l: `tiles` instances after pruning, rectangle r, live
// mask, key).  Convergent: every lane of the wave calls this.  <= RANK_TILES tiles (almost all): the lane itself; up to COOP_TILES:
// pairs spread over the lanes (prefix sum of the areas + a 6-step search in LDS; one lane walking a 64-tile rectangle of its own
// would hold the wave for 64 steps); larger: the whole wave, one rectangle after the other (the reference's thread-serial double
// loop, rasterizer_impl.cu:98-109, is its tail-latency problem).
#ifndef TGS_EMIT_RANK
#define TGS_EMIT_RANK 9
#endif
constexpr int EMIT_RANK = TGS_EMIT_RANK;     // rectangles up to this many tiles are walked by the splat's own lane (<= 32: the live mask's low word)
template <typename F>
__device__ __forceinline__ void wave_emit_instances(uint32_t tiles, ushort4 r, uint2 live2, unsigned long long key, uint32_t gx, MidTab& tab, int lane, F&& f)
{
    const uint32_t rw = (uint32_t)r.z - r.x, area = tiles ? rw * ((uint32_t)r.w - r.y) : 0u;
    const unsigned long long live = (unsigned long long)live2.x | ((unsigned long long)live2.y << 32);
    if (area != 0 && area <= (uint32_t)EMIT_RANK) {
        uint32_t kx = 0, t = (uint32_t)r.y * gx + r.x;
#pragma unroll
        for (int k = 0; k < EMIT_RANK; k++) {
            if ((uint32_t)k < area) {
                if ((live2.x >> k) & 1u) f(t + kx, key);
                if (++kx == rw) { kx = 0; t += gx; }
            }
        }
    }
    const uint32_t mid_area = (area > (uint32_t)EMIT_RANK && area <= (uint32_t)COOP_TILES) ? area : 0u;
    const uint32_t incl = wave_iscan_u32(mid_area, lane), total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total > 0) {                                        // wave-uniform
        tab.excl[lane] = incl - mid_area;
        tab.rect[lane] = make_uint2((uint32_t)r.x | ((uint32_t)r.y << 16), rw);
        tab.live[lane] = live;
        tab.key[lane] = key;
        wave_sync();
        for (uint32_t w = lane; w < total; w += 64) {
            uint32_t lo = 0, hi = 63;                       // owner: the last lane whose first pair is <= w (lanes without pairs share their successor's start)
#pragma unroll
            for (int it = 0; it < 6; it++) { const uint32_t mid = (lo + hi + 1) >> 1; if (tab.excl[mid] <= w) lo = mid; else hi = mid - 1; }
            const uint2 rc = tab.rect[lo];
            const uint32_t k = w - tab.excl[lo], rw2 = rc.y;
            if (!((tab.live[lo] >> k) & 1ull)) continue;    // a tile of the rectangle the splat cannot reach: no instance
            const uint32_t ky = k / rw2;
            f(((rc.x >> 16) + ky) * gx + (rc.x & 0xffffu) + (k - ky * rw2), tab.key[lo]);
        }
        wave_sync();                                        // the table is reused by the wave's next step
    }
    unsigned long long bigs = __builtin_amdgcn_ballot_w64(area > (uint32_t)COOP_TILES);
    while (bigs) {                                          // wave-uniform
        const int src = __builtin_ctzll(bigs);
        bigs &= bigs - 1;
        const uint32_t x0 = (uint32_t)__builtin_amdgcn_readlane((int)r.x, src), y0 = (uint32_t)__builtin_amdgcn_readlane((int)r.y, src);
        const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)rw, src), n = (uint32_t)__builtin_amdgcn_readlane((int)area, src);
        const unsigned long long kk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), src) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, src);
        for (uint32_t k = lane; k < n; k += 64) { const uint32_t ky = k / w2; f((y0 + ky) * gx + x0 + (k - ky * w2), kk); }
    }
}

__global__ __launch_bounds__(BIN_THREADS) void k_bin_count(int P, uint32_t chunk, const GeomState g, const ImgState s, uint32_t gx, uint32_t T, uint32_t band)
{
    extern __shared__ uint32_t bin_lds[];                   // min(T, band) counters
    __shared__ MidTab tabs[BIN_THREADS / WAVE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t lo = blockIdx.x * chunk;
    uint32_t* row = s.bin_table + (size_t)blockIdx.x * T;
    for (uint32_t band0 = 0; band0 < T; band0 += band) {
        const uint32_t nb = min(band, T - band0);
        for (uint32_t i = threadIdx.x; i < nb; i += BIN_THREADS) bin_lds[i] = 0u;
        __syncthreads();
        // (the three loads of a step do not wait for each other -- rect and live of a culled Gaussian are simply not used -- and the next
        // step's are in flight while this one counts: a chunk is a handful of steps, each a full memory round trip otherwise)
        uint32_t n_tiles = 0; ushort4 n_r = make_ushort4(0, 0, 0, 0); uint2 n_live = make_uint2(0u, 0u);
        auto fetch = [&](uint32_t idx) {
            if (idx < (uint32_t)P) { n_tiles = g.tiles_touched[idx]; n_r = g.rect[idx]; n_live = g.live[idx]; } else n_tiles = 0u;
        };
        fetch(lo + threadIdx.x);
#pragma unroll 1
        for (uint32_t base = lo; base < lo + chunk; base += BIN_THREADS) {
            const uint32_t tiles = n_tiles; const ushort4 r = n_r; const uint2 live = n_live;
            if (base + BIN_THREADS < lo + chunk) fetch(base + BIN_THREADS + threadIdx.x);
            wave_emit_instances(tiles, r, live, 0ull, gx, tabs[wv], lane, [&](uint32_t t, unsigned long long) {
                if (t - band0 < nb) atomicAdd(&bin_lds[t - band0], 1u);
            });
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb; i += BIN_THREADS) row[band0 + i] = bin_lds[i];
        __syncthreads();
    }
}

// per tile: table[w][t] <- sum of table[w'][t], w' < w; tile_count[t] <- the total.  64 tiles per workgroup, wave q takes 1 / COL_WAVES of the chunks.
constexpr int COL_WAVES = BIN_WGS_MAX / 32;       // waves per workgroup of k_bin_colscan: every thread carries 32 chunks' counts (4 waves at 128 chunks, 8 at 256)
__global__ __launch_bounds__(64 * COL_WAVES) void k_bin_colscan(const ImgState s, uint32_t T, uint32_t nchunks)
{
    __shared__ uint32_t part[COL_WAVES][WAVE];
    __shared__ uint32_t bh[34];
    const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (threadIdx.x < 34) bh[threadIdx.x] = 0u;
    const uint32_t t = blockIdx.x * WAVE + l;
    const uint32_t per = (nchunks + COL_WAVES - 1) / COL_WAVES, w0 = min(nchunks, q * per), w1 = min(nchunks, w0 + per);
    uint32_t* col = s.bin_table + t;
    uint32_t c[BIN_WGS_MAX / COL_WAVES];
    uint32_t sum = 0;
    if (t < T) {
#pragma unroll
        for (int k = 0; k < BIN_WGS_MAX / COL_WAVES; k++) { c[k] = w0 + k < w1 ? col[(size_t)(w0 + k) * T] : 0u; }
#pragma unroll
        for (int k = 0; k < BIN_WGS_MAX / COL_WAVES; k++) sum += c[k];
    }
    part[q][l] = sum;
    __syncthreads();
    if (t < T) {
        uint32_t run = 0;
        for (int i = 0; i < q; i++) run += part[i][l];
#pragma unroll
        for (int k = 0; k < BIN_WGS_MAX / COL_WAVES; k++) { if (w0 + k < w1) col[(size_t)(w0 + k) * T] = run; run += c[k]; }
        if (q == COL_WAVES - 1) {
            s.tile_count[t] = run;
            atomicAdd(&bh[length_bucket(run)], 1u);         // the frame's histogram of list lengths (k_scan: tile order), first per workgroup in LDS
        }
    }
    __syncthreads();
    if (threadIdx.x < 34 && bh[threadIdx.x]) atomicAdd(&s.aux->hist[threadIdx.x], bh[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// k_scatter: instance -> its tile's segment.  Also finishes the Gaussian offsets (exclusive scan).
// ---------------------------------------------------------------------------------------------
#ifndef TGS_SCATTER_XCD_MAJOR
#define TGS_SCATTER_XCD_MAJOR 1
#endif
// bijection workgroup b -> chunk (n of each): the workgroups with b % 8 == x take chunks [start_x, start_x + count_x) in the order of b / 8
__device__ __forceinline__ uint32_t xcd_major(uint32_t b, uint32_t n)
{
    const uint32_t x = b & 7u, q = n >> 3, r = n & 7u;
    return x * q + min(x, r) + (b >> 3);
}
__global__ __launch_bounds__(BIN_THREADS) void k_scatter(int P, uint32_t chunk, const GeomState g, const ImgState s, const BinState b, uint32_t gx, uint32_t T, uint32_t band)
{
    extern __shared__ uint32_t bin_lds[];                   // min(T, band) cursors: absolute positions in b.keys
    __shared__ MidTab tabs[BIN_THREADS / WAVE];
    __shared__ uint32_t wtot[2][BIN_THREADS / WAVE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (frame_rejected(s)) return;
    // Which chunk this workgroup scatters.  Inside a tile's segment the instances lie in chunk order (k_bin_colscan), ~2 keys = 18 B per chunk
    // and tile at config 3: every store is a partial line, and the L2 of an XCD can only merge the partial lines its OWN workgroups write.
    // Workgroups b, b + 8, b + 16 ... share an XCD (observed round-robin placement; used for speed only, any mapping is correct), so XCD x
    // takes a CONSECUTIVE range of chunks: its 16 runs in a tile's segment are adjacent (~290 B) and leave the L2 as whole lines
    // (WRITE_SIZE of this kernel 38 MB -> see DESIGN.md for 6 MB of keys with chunk = workgroup index).
    const uint32_t w = TGS_SCATTER_XCD_MAJOR ? xcd_major(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t lo = w * chunk;
    const uint32_t* row = s.bin_table + (size_t)w * T;
    for (uint32_t band0 = 0; band0 < T; band0 += band) {
        const uint32_t nb = min(band, T - band0);
        // cursors of the band: eight tiles per thread and trip with their sixteen loads in flight together (one tile per trip is one memory
        // round trip per trip: eight in a row at 1080p, a third of this kernel's time alone on the GPU)
        for (uint32_t i0 = threadIdx.x; i0 < nb; i0 += 8 * BIN_THREADS) {
            uint32_t st[8], cu[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t ic = min(i0 + u * BIN_THREADS, nb - 1u); st[u] = s.ranges[band0 + ic].x; cu[u] = row[band0 + ic]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t i = i0 + u * BIN_THREADS; if (i < nb) bin_lds[i] = st[u] + cu[u]; }
        }
        __syncthreads();
        int par = 0;
        uint32_t n_tiles = 0, n_depth = 0, n_bsum = 0; ushort4 n_r = make_ushort4(0, 0, 0, 0); uint2 n_live = make_uint2(0u, 0u);
        auto fetch = [&](uint32_t idx) {                    // (as in k_bin_count: independent loads, one step ahead)
            if (idx < (uint32_t)P) { n_tiles = g.tiles_touched[idx]; n_r = g.rect[idx]; n_live = g.live[idx]; n_depth = __float_as_uint(g.depth[idx]); n_bsum = g.block_sums[idx / PRE_BLOCK]; }
            else n_tiles = 0u;
        };
        fetch(lo + threadIdx.x);
#pragma unroll 1
        for (uint32_t base = lo; base < lo + chunk; base += BIN_THREADS, par ^= 1) {
            const uint32_t idx = base + threadIdx.x;
            const uint32_t tiles = n_tiles, bsum = n_bsum; const ushort4 r = n_r; const uint2 live = n_live;
            const unsigned long long key = ((unsigned long long)n_depth << 32) | idx;
            if (base + BIN_THREADS < lo + chunk) fetch(base + BIN_THREADS + threadIdx.x);
            if (band0 == 0) {                               // offsets[idx] = block_sums[its PRE_BLOCK] + prefix inside the block
                const uint32_t inc = wave_iscan_u32(tiles, lane);
                if (lane == 63) wtot[par][wv] = inc;
                __syncthreads();
                if (idx < (uint32_t)P) {
                    uint32_t off = bsum + inc - tiles;
                    for (int i = wv & ~3; i < wv; i++) off += wtot[par][i];
                    g.offsets[idx] = off;
                    if (tiles > 0) reinterpret_cast<uint32_t*>(g.pack + 4 * (size_t)idx + 2)[3] = off;
                }
            }
            wave_emit_instances(tiles, r, live, key, gx, tabs[wv], lane, [&](uint32_t t, unsigned long long k) {
                if (t - band0 < nb) b.keys[atomicAdd(&bin_lds[t - band0], 1u)] = k;
            });
        }
        __syncthreads();
    }
}

// (the bitonic network helpers pair_flip / pair_disperse / cmp_swap live in tgs_device.hpp)

// the per-instance record of sorted entry `pos` of a tile (what renderCUDA fetches per entry: forward.cu:315-321,355 and
// backward.cu:470-480) from the Gaussian's 64-B pack line p0..p3

// Per-tile sort in LDS, tiles visited in tile_order (longest lists first).  Three classes by list length n, all in ONE launch of
// 1024-thread workgroups (the classes are latency bound on their longest lists; launched one after the other their critical paths add up:
// 25 + 20 + 7 us at config 3, 28 us together):
//   n >= 1024 : one workgroup per tile                 } all steps that stay inside an aligned 128-key chunk are run by one
//   n >= 128  : four tiles per workgroup, 256 threads  } wave without workgroup barriers (21 barriers instead of 78 at 4096 keys)
//               each; the four run the network of the longest of them so that every thread meets the same barriers -- on a
//               sorted list the extra all-ascending steps exchange nothing
//   n <  128  : one WAVE per tile, sixteen tiles per workgroup, no barrier at all
// The grid is an upper bound per class (exact after tgs_forward's read-back); the tile indices come from Meta.
// sorts keys [0, n) of lk with NT threads (tid in [0, NT)) that all call this; npad_loop >= next_pow2(n): the merge sizes to run
template <int NT>
__device__ __forceinline__ void sort_tile_lds(unsigned long long* lk, uint32_t tid, uint32_t n, uint32_t npad_loop)
{
    const uint32_t half = npad_loop >> 1;
    const uint32_t lane = tid & 63, wv = tid >> 6;
    constexpr uint32_t W = NT / 64;
    // every aligned chunk of 128 keys is sorted, and later re-merged, in the registers of one wave (regs_sort128, tgs_device.hpp)
    for (uint32_t c = wv * SORT_CHUNK; c < n; c += W * SORT_CHUNK) {
        unsigned long long k0, k1;
        regs_load_chunk(lk, c, n, lane, k0, k1);
        regs_sort128(k0, k1, lane);
        regs_store_chunk(lk, c, n, lane, k0, k1);
    }
    __syncthreads();
    for (uint32_t k = 2 * SORT_CHUNK; k <= npad_loop; k <<= 1) {
        for (uint32_t p = tid; p < half; p += NT) { uint32_t i, l; pair_flip(p, k, i, l); cmp_swap(lk, i, l, n); }
        __syncthreads();
        for (uint32_t j = k >> 2; j >= SORT_CHUNK; j >>= 1) {
            for (uint32_t p = tid; p < half; p += NT) { uint32_t i, l; pair_disperse(p, j, i, l); cmp_swap(lk, i, l, n); }
            __syncthreads();
        }
        for (uint32_t c = wv * SORT_CHUNK; c < n; c += W * SORT_CHUNK) {
            unsigned long long k0, k1;
            regs_load_chunk(lk, c, n, lane, k0, k1);
            regs_disperse_from64(k0, k1, lane);
            regs_store_chunk(lk, c, n, lane, k0, k1);
        }
        __syncthreads();
    }
}

// ---- overflow path: lists longer than the LDS sort (sort_cap keys), sorted in global memory by workgroups of the SAME launch ----
// The reference sorts any R globally (rasterizer_impl.cu:303-308).  Here the rare list that does not fit LDS runs the same all-ascending
// bitonic network in global memory, LDS for the strides below `cap`: the last OVF_WORKERS workgroups of k_tile_sort's grid take the
// overflow tiles (worker w: tiles w, w + OVF_WORKERS, ...), ONE workgroup per tile walking the whole network with workgroup barriers.
// They read n_overflow on the device and return at once when there is nothing to do, so the sync-free forward needs neither host-sized
// launches nor a rejection, and a frame with ordinary lists pays nothing.  Deliberately no barrier BETWEEN workgroups: a first version
// spread one list over 128 workgroups with a spinning grid barrier, and with four streams in flight the 4 x 128 spinning workgroups filled
// every CU while their peers waited for a slot -- config 5 took 37 ms per k_render_bwd instead of 0.9.  A workgroup that depends on no other
// cannot be starved; the price is the time of one long list on one workgroup (~0.1 ms per 10 k entries), which the render kernels spend on
// such a tile many times over.
constexpr uint32_t OVF_WORKERS = 32;
constexpr uint32_t OVF_THREADS = 1024;

// all global stores of the workgroup are visible to all of its waves (L1 of the CU invalidated) before anybody goes on
__device__ __forceinline__ void ovf_wg_sync()
{
    __threadfence();
    __syncthreads();
}

__device__ __forceinline__ void ovf_local(const ImgState& s, const BinState& b, unsigned long long* lk, uint32_t ot, uint32_t blk, uint32_t k_only, uint32_t cap)
{
    const uint32_t tile = s.ovf_tiles[ot];
    const uint2 rg = s.ranges[tile];
    const uint32_t n = rg.y - rg.x;
    const uint32_t b0 = blk * cap;
    if (b0 >= n) return;                                    // (uniform over the workgroup)
    if (k_only > next_pow2(n)) return;
    const uint32_t m = min(cap, n - b0);                    // real keys in this block
    unsigned long long* gk = b.keys + rg.x + b0;
    for (uint32_t i = threadIdx.x; i < m; i += OVF_THREADS) lk[i] = gk[i];
    __syncthreads();
    const uint32_t half = cap >> 1;
    if (k_only == 0) {
        for (uint32_t k = 2; k <= cap; k <<= 1) {
            for (uint32_t t = threadIdx.x; t < half; t += OVF_THREADS) { uint32_t i, l; pair_flip(t, k, i, l); cmp_swap(lk, i, l, m); }
            __syncthreads();
            for (uint32_t j = k >> 2; j > 0; j >>= 1) {
                for (uint32_t t = threadIdx.x; t < half; t += OVF_THREADS) { uint32_t i, l; pair_disperse(t, j, i, l); cmp_swap(lk, i, l, m); }
                __syncthreads();
            }
        }
    } else {
        for (uint32_t j = cap >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < half; t += OVF_THREADS) { uint32_t i, l; pair_disperse(t, j, i, l); cmp_swap(lk, i, l, m); }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < m; i += OVF_THREADS) gk[i] = lk[i];
    __syncthreads();                                        // lk is reused by the workgroup's next item
}

// OVF_THREADS comparators of overflow tile `ot`, starting at comparator c0; flip != 0: flip step of merge size k, else disperse step of distance j
__device__ __forceinline__ void ovf_global(const ImgState& s, const BinState& b, uint32_t ot, uint32_t c0, uint32_t k, uint32_t j, int flip)
{
    const uint32_t tile = s.ovf_tiles[ot];
    const uint2 rg = s.ranges[tile];
    const uint32_t n = rg.y - rg.x, npad = next_pow2(n);
    const uint32_t t = c0 + threadIdx.x;
    if (k > npad || t >= (npad >> 1)) return;
    uint32_t i, l;
    if (flip) pair_flip(t, k, i, l); else pair_disperse(t, j, i, l);
    cmp_swap(b.keys + rg.x, i, l, n);
}

__device__ __forceinline__ void finalize_entry(float4 p0, float4 p1, float4 p2, float4 p3, uint32_t pos, uint32_t tile, uint32_t gx, const BinState& b)
{
    const uint32_t rmin = __float_as_uint(p2.y), rmax = __float_as_uint(p2.z);
    const uint32_t minx = rmin & 0xffffu, miny = rmin >> 16, maxx = rmax & 0xffffu, maxy = rmax >> 16;
    const uint32_t tx = tile % gx, ty = tile / gx;
    b.recA[pos] = p0;
    b.recB[pos] = p1;
    const unsigned long long qm = quadrant_mask(make_float2(p0.x, p0.y), make_float4(p0.z, p0.w, p1.x, p1.y), tx, ty);
    b.recC[pos] = make_float2(p2.x, __uint_as_float(blocks_of_quadrants(qm)));
    b.qmask[pos] = make_uint2((uint32_t)qm, (uint32_t)(qm >> 32));
    // row of this instance in the gradient slab: the Gaussian's rows are its LIVE tiles in rectangle order
    const uint32_t rw = maxx - minx, k = (ty - miny) * rw + (tx - minx);
    uint32_t ord = k;
    if (rw * (maxy - miny) <= (uint32_t)COOP_TILES) {       // p3: 64-bit mask of the rectangle's live tiles
        const unsigned long long live = (unsigned long long)__float_as_uint(p3.x) | ((unsigned long long)__float_as_uint(p3.y) << 32);
        ord = (uint32_t)__builtin_popcountll(live & ((1ull << k) - 1ull));
    }
    b.slot[pos] = __float_as_uint(p2.w) + ord;
}

__device__ __forceinline__ void ovf_worker(const GeomState& g, const ImgState& s, const BinState& b, unsigned long long* lk, uint32_t w, uint32_t nw, uint32_t cap, uint32_t gx)
{
    const uint32_t n_ovf = s.meta->n_overflow;              // written by k_scan
    for (uint32_t ot = w; ot < n_ovf; ot += nw) {           // one workgroup per overflow tile: no dependency on any other workgroup
        const uint2 rg = s.ranges[s.ovf_tiles[ot]];
        const uint32_t npad = next_pow2(rg.y - rg.x);       // >= 2 cap: the list is longer than cap
        const uint32_t bpt = npad / cap;                    // aligned blocks of cap keys
        const uint32_t half = npad >> 1;                    // comparators per step
        for (uint32_t blk = 0; blk < bpt; blk++) ovf_local(s, b, lk, ot, blk, 0u, cap);
        for (uint32_t k = cap * 2; k <= npad; k <<= 1) {
            ovf_wg_sync();
            for (uint32_t c0 = 0; c0 < half; c0 += OVF_THREADS) ovf_global(s, b, ot, c0, k, 0u, 1);
            for (uint32_t j = k >> 2; j >= cap; j >>= 1) {
                ovf_wg_sync();
                for (uint32_t c0 = 0; c0 < half; c0 += OVF_THREADS) ovf_global(s, b, ot, c0, k, j, 0);
            }
            ovf_wg_sync();
            for (uint32_t blk = 0; blk < bpt; blk++) ovf_local(s, b, lk, ot, blk, k, cap);
        }
        ovf_wg_sync();
    }
}

__global__ __launch_bounds__(1024) void k_tile_sort(const GeomState g, const ImgState s, const BinState b, uint32_t gx, uint32_t sort_cap, uint32_t heavy_blocks,
                                                    uint32_t mid_blocks, uint32_t small_blocks, uint32_t ovf_blocks)
{
    extern __shared__ unsigned long long lk[];
    __shared__ uint32_t grp_npad[4];
    if (frame_rejected(s)) return;
    const uint32_t n_nonempty = s.meta->n_nonempty;
    const uint32_t n_heavy = min(s.meta->n_heavy, n_nonempty), n_mid = min(max(s.meta->n_mid, n_heavy), n_nonempty);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (blockIdx.x >= heavy_blocks + mid_blocks + small_blocks) {          // the grid's tail: workers of the lists beyond the LDS sort
        ovf_worker(g, s, b, lk, blockIdx.x - (heavy_blocks + mid_blocks + small_blocks), ovf_blocks, sort_cap, gx);
        return;
    }
    if (blockIdx.x < heavy_blocks) {
        const uint32_t t = blockIdx.x;
        if (t >= n_heavy) return;
        const uint4 td = s.tile_desc[t];
        const uint32_t n = td.z - td.y;
        for (uint32_t i = threadIdx.x; i < n; i += 1024) b.tile_of[td.y + i] = td.x;
        if (n > sort_cap) return;                           // (longer lists: the overflow workers -- they sort AND finalize them)
        for (uint32_t i = threadIdx.x; i < n; i += 1024) lk[i] = b.keys[td.y + i];
        __syncthreads();
        if (n >= 2) {
            sort_tile_lds<1024>(lk, threadIdx.x, n, next_pow2(n));
            for (uint32_t i = threadIdx.x; i < n; i += 1024) b.keys[td.y + i] = lk[i];
        }
    } else if (blockIdx.x < heavy_blocks + mid_blocks) {
        const uint32_t grp = threadIdx.x >> 8, tid = threadIdx.x & 255u;
        const uint32_t t = n_heavy + (blockIdx.x - heavy_blocks) * 4 + grp;
        if (n_heavy + (blockIdx.x - heavy_blocks) * 4 >= n_mid) return;          // the whole workgroup is behind the class
        uint32_t n = 0, start = 0, tile = 0;
        if (t < n_mid) { const uint4 td = s.tile_desc[t]; tile = td.x; start = td.y; n = td.z - td.y; }
        for (uint32_t i = tid; i < n; i += 256) b.tile_of[start + i] = tile;
        if (n > 1024u || n > sort_cap) n = 0;               // (a list beyond sort_cap belongs to the overflow workers; n > 1024 cannot happen by the class boundary)
        unsigned long long* seg = lk + grp * 1024;
        for (uint32_t i = tid; i < n; i += 256) seg[i] = b.keys[start + i];
        if (tid == 0) grp_npad[grp] = n < 2 ? 2u : next_pow2(n);
        __syncthreads();
        const uint32_t npad_wg = max(max(grp_npad[0], grp_npad[1]), max(grp_npad[2], grp_npad[3]));
        sort_tile_lds<256>(seg, tid, n, npad_wg);
        for (uint32_t i = tid; i < n; i += 256) b.keys[start + i] = seg[i];
    } else {
        const uint32_t t = n_mid + (blockIdx.x - heavy_blocks - mid_blocks) * 16 + wv;
        if (t >= n_nonempty) return;                        // wave-uniform: no workgroup barrier below
        const uint4 td = s.tile_desc[t];
        const uint32_t n = td.z - td.y;
        for (uint32_t i = lane; i < n; i += 64) b.tile_of[td.y + i] = td.x;
        if (n < 2 || n > sort_cap || n > SORT_CHUNK) return;    // (n < 128 by the class boundary in k_scan)
        unsigned long long k0, k1;                          // global memory -> registers -> global memory: no LDS
        regs_load_chunk(b.keys + td.y, 0u, n, lane, k0, k1);
        if (n >= 2) {
            regs_sort128(k0, k1, lane);
            regs_store_chunk(b.keys + td.y, 0u, n, lane, k0, k1);
        }
    }
}

// FIN_E sorted instances per thread, evenly over all R of them: the tile comes with the sorted keys (tile_of; a binary search in the
// range starts was 13 dependent loads), the thread gathers the Gaussians' 64-B lines and writes the 40-B records, the quadrant masks
// and the slab rows.  The loads of the FIN_E instances are issued together, level by level (keys, then lines): with one instance per
// thread the whole grid fitted the chip ~1.5 times, and every wave of a pass was in the same phase -- loading, then computing, then storing.
#ifndef TGS_FIN_E
#define TGS_FIN_E 2
#endif
constexpr int FIN_E = TGS_FIN_E;
__global__ __launch_bounds__(256) void k_finalize(const GeomState g, const ImgState s, const BinState b, uint32_t gx, uint32_t T)
{
    if (frame_rejected(s)) return;
    const uint32_t R = (uint32_t)s.meta->R;
    const uint32_t p0 = blockIdx.x * (256 * FIN_E) + threadIdx.x;
    if (p0 >= R) return;
    uint32_t id[FIN_E], tile[FIN_E];
#pragma unroll
    for (int e = 0; e < FIN_E; e++) {
        const uint32_t pc = min(p0 + e * 256, R - 1u);        // clamped, not predicated: a load under its own branch is waited for inside it, and the
        id[e] = (uint32_t)b.keys[pc];                           // FIN_E key loads (then the pack lines) are meant to be in flight together
        tile[e] = b.tile_of[pc];
    }
    float4 q0[FIN_E], q1[FIN_E], q2[FIN_E], q3[FIN_E];
#pragma unroll
    for (int e = 0; e < FIN_E; e++) {
        const float4* pk = g.pack + 4 * (size_t)id[e];         // one 64-B line per Gaussian
        q0[e] = pk[0]; q1[e] = pk[1]; q2[e] = pk[2]; q3[e] = pk[3];
    }
#pragma unroll
    for (int e = 0; e < FIN_E; e++) {
        const uint32_t p = p0 + e * 256;
        if (p < R) finalize_entry(q0[e], q1[e], q2[e], q3[e], p, tile[e], gx, b);
    }
}

// A tile nothing is blended into: C = 0, T = 1 -> background (forward.cu:366-373 with an empty range).  Also what a frame
// rejected by tgs_forward_async renders, so its image is defined.
__device__ __forceinline__ void fill_tile_background(const ImgState& s, uint32_t tile, uint32_t t, int W, int H, uint32_t gx,
                                                     const float* __restrict__ bg, float* __restrict__ out_color)
{
    const int px = (tile % gx) * TILE + (t & 15), py = (tile / gx) * TILE + (t >> 4);
    if (px < W && py < H) {
        const size_t pix_id = (size_t)W * py + px, N = (size_t)W * H;
        s.final_T[pix_id] = 1.0f;
        s.n_contrib[pix_id] = 0;
        out_color[pix_id] = bg[0]; out_color[N + pix_id] = bg[1]; out_color[2 * N + pix_id] = bg[2];
    }
}

// ---------------------------------------------------------------------------------------------
// k_render_fwd: front-to-back compositing (renderCUDA, forward.cu:261-374) in 256-thread workgroups = 4 waves, one wave per 4x4-pixel block.
// Inside a wave every 16-lane DPP row is one 2x2-pixel quadrant of the block: 4 pixels x 4 CONSECUTIVE entries of the quadrant's OWN list
// (the entries whose 64-bit quadrant mask names it; tgs_device.hpp "quadrant culling").  Every lane evaluates one (pixel, entry) pair --
// conic power, exp, alpha -- and the four lanes of a quad then walk the transmittance chain of their pixel together (DPP quad broadcasts
// of 1-alpha, fwd_chain4), so one pass of the loop retires four entries of each quadrant's list with the sequential semantics of
// forward.cu:325-362 intact (T is multiplied in list order; the first entry that would push T below 1e-4 stops the pixel and is not
// blended).  The quad chains never leave the quad, so the four rows may walk four different lists in one instruction stream.
// Why: the kernel is bound by VALU issue, and only the lanes inside the splat's alpha >= 1/255 footprint do useful work -- 33 % of them
// when a wave's 16 pixels all took the block's entries, 64 % quadrant by quadrant (tools/culling_potential.py).
//
// Workgroups (round 3; rounds 1-2 and most of round 3 ran one 1024-thread workgroup of 16 waves per tile): a tile with >= LIGHT_MAX
// instances is composited by FOUR workgroups, one per 8x8-pixel quarter (its 4 blocks, one wave per block: the per-list chain stays as
// short as with 16 waves); a light tile (light != 0) by ONE, its wave w walking four blocks one after the other -- a light tile has a
// single staging round, so no pixel state outlives a block, and a short tile is mostly descriptor -> records -> staging latency that 16
// waves should not sit through.  Eight workgroups fit a CU.
// Why (per-wave busy stamps): in a 16-wave workgroup the waves were busy half of the workgroup's span -- the block
// with the most contributors sets the span, and nothing else can use the slots of the waves that wait for it.  Pixels of different
// quarters never interact, so a quarter leaves as soon as ITS four blocks are done, and eight workgroups per CU overlap each other's
// latencies instead of two.  Price: every quarter stages the whole list (4 x the L2 -> LDS traffic of the records; the quarters of a tile
// get the same blockIdx modulo 8 -- the same XCD's L2 -- and consecutive slots there).  Measured against the 16-wave kernel with its
// light groups (A/B by library on one box): kernel alone 63.4 -> 59.0 us, the 8-view step 2.01 -> 1.93 ms (the finer workgroups also
// leave the other streams' kernels more room); 512 entries per round instead of 256: no gain alone, 2.00 ms per step (5 workgroups per CU).
// Long lists go first (tile_order): the kernel ends when its longest list does.
// ---------------------------------------------------------------------------------------------
constexpr int FQ_THREADS = 256;
#ifndef TGS_FQ_CH
#define TGS_FQ_CH 256
#endif
constexpr int FQ_CH = TGS_FQ_CH;           // list entries staged per round: FQ_CH / 256 whole entries per thread
constexpr int FQ_EPT = FQ_CH / FQ_THREADS;
static_assert(FQ_CH % FQ_THREADS == 0 && FQ_CH >= LIGHT_MAX && FQ_CH <= 768, "whole entries per thread; a light tile is one round; slots are 10-bit in the block lists");
constexpr int FQ_NULL = FQ_CH;

// one block, one staged round: this wave's pass over the entries of the round that reach its block (forward.cu:325-362
// semantics: the kernel's header).  Returns true when the block's last live pixel ended.
__device__ __forceinline__ bool fwd_q_block_round(const float4* sA, const float4* sB, const float* sC, const uint2* sQ, unsigned short* list,
                                                  unsigned short (*ql)[QL_ROW_F], uint32_t cnt, int blk, int lane, uint32_t cbase, float pixfx, float pixfy,
                                                  float vone, const QuadMasks& qm, bool& done, float& T, float& C0, float& C1, float& C2, uint32_t& last_contributor)
{
    const int qd = lane >> 4, e = lane & 3;
    // (both counts are wave-uniform -- sums of ballot popcounts -- but reach the loops in VGPRs: readfirstlane makes the loop tests scalar
    // compares instead of lane-mask algebra on the exec mask, 7 scalar instructions per pass less)
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)build_own_list_q<FQ_CH>(list, sQ, cnt, blk, lane));
    const unsigned short* myq = &ql[qd][e];
#pragma unroll 1
    for (uint32_t c0 = 0; c0 < n; c0 += QCH_F) {
        const uint32_t nq = (uint32_t)__builtin_amdgcn_readfirstlane((int)build_chunk_quadrant_lists_128(ql, list, c0, n, lane, FQ_NULL));
        uint32_t jn = myq[0];                               // the index one pass ahead: the pass itself then waits for ONE LDS round trip (the records), not two
#pragma unroll 1
        for (uint32_t k = 0; k < nq; k += 4) {
            const uint32_t j = jn;
            jn = myq[k + 4];                                // (behind the list's end: null slots up to QL_ROW_F)
            const float4 a = sA[j];
            const float4 bb = sB[j];
            const float cc = sC[j];
            const float dx = a.x - pixfx, dy = a.y - pixfy;
#if TGS_FAST_MATH
            const float power2 = pair_power2(a.z, a.w, bb.x, dx, dy);
            const float alpha = fminf(0.99f, bb.y * __builtin_amdgcn_exp2f(power2));
#else
            const float power2 = -0.5f * (a.z * dx * dx + bb.x * dy * dy) - a.w * dx * dy;
            const float alpha = fminf(0.99f, bb.y * expf(power2));
#endif
            const bool live = !done && !(power2 > 0.0f) && !(alpha < 1.0f / 255.0f);
            const float pown = live ? 1.f - alpha : 1.0f;
            float y, x, x3;
            fwd_chain4(pown, T, y, x, vone, qm);
            const bool fail = live && (x < 0.0001f);
            const bool upd = live && !fail;
            float cand = fail ? y : -1.0f;
            quad_max_bcast3(cand, x, x3);
            const float w = upd ? alpha * y : 0.f;
            C0 += bb.z * w; C1 += bb.w * w; C2 += cc * w;
            last_contributor = upd ? cbase + j : last_contributor;
            const bool stop = cand >= 0.0f;
            T = stop ? cand : x3;
            done = done || stop;
            if (__builtin_amdgcn_ballot_w64(!done) == 0) return true;
        }
    }
    return false;
}

// the block's pixels are complete: quad shares -> pixel values, stores; returns the deepest blended position of the block
__device__ __forceinline__ uint32_t fwd_q_block_store(const ImgState& s, float* __restrict__ out_color, int W, int H, int px, int py, bool inside, int lane,
                                                      float bg0, float bg1, float bg2, float T, float C0, float C1, float C2, uint32_t last_contributor)
{
    TGS_DPP_ADD(C0, 0xB1, 0xf); TGS_DPP_ADD(C0, 0x4E, 0xf);
    TGS_DPP_ADD(C1, 0xB1, 0xf); TGS_DPP_ADD(C1, 0x4E, 0xf);
    TGS_DPP_ADD(C2, 0xB1, 0xf); TGS_DPP_ADD(C2, 0x4E, 0xf);
    {
        uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)last_contributor, 0xB1, 0xf, 0xf, false);
        last_contributor = max(last_contributor, o);
        o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)last_contributor, 0x4E, 0xf, 0xf, false);
        last_contributor = max(last_contributor, o);
    }
    if (inside && (lane & 3) == 0) {
        const size_t pix_id = (size_t)W * py + px, N = (size_t)W * H;
        s.final_T[pix_id] = T;
        s.n_contrib[pix_id] = last_contributor;
        out_color[pix_id] = C0 + T * bg0;
        out_color[N + pix_id] = C1 + T * bg1;
        out_color[2 * N + pix_id] = C2 + T * bg2;
    }
    return wave_max_u32(inside ? last_contributor : 0u);
}

__global__ __launch_bounds__(FQ_THREADS, 8) void k_render_fwd(const ImgState s, const BinState b, int W, int H, uint32_t gx,
                                                              const float* __restrict__ bg, float* __restrict__ out_color, uint32_t n_tiles, int light)
{
    __shared__ float4 sA[FQ_CH + 1];
    __shared__ float4 sB[FQ_CH + 1];
    __shared__ float sC[FQ_CH + 1];
    __shared__ uint2 sQ[FQ_CH];
    __shared__ __attribute__((aligned(16))) unsigned short lists[4][FQ_CH + 8];
    __shared__ __attribute__((aligned(16))) unsigned short qlists[4][4][QL_ROW_F];
    __shared__ uint32_t wave_alive[2][4];
    __shared__ uint32_t wave_qmax[4];
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    // heavy role: blockIdx = 8 * (4 * (t / 8) + quarter) + t % 8 for the tile of rank t: a tile's quarters share an XCD, the 8 longest tiles come first
    const uint32_t hk = blockIdx.x >> 3, ht8 = (hk >> 2) * 8u + (blockIdx.x & 7u), quarter = hk & 3u;
    const uint4 tdh = s.tile_desc[min(ht8, n_tiles - 1u)];           // (candidate descriptor, in flight beside the frame's counts)
    const uint4 ff = frame_counts(s);
    const uint32_t n_all = (ff.x & META_ERR_CAPACITY) ? 0u : ff.y;     // a frame tgs_forward_async rejected renders the background everywhere
    const uint32_t n_ne = light ? min(ff.w, n_all) : n_all;           // tiles composited by four workgroups: the first n_ne entries of tile_order
    const uint32_t n_light = n_all - n_ne;
    const uint32_t heavy_wgs = 4u * ((n_ne + 7u) & ~7u);
    // (cannot happen with a grid sized by launch_render_fwd from bounds k_scan has enforced; a tile left out would be a silent hole in the image)
    if (blockIdx.x == 0 && threadIdx.x == 0 && heavy_wgs + n_light > gridDim.x) atomicOr(&s.meta->error, META_ERR_TILE_BOUND);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qd = lane >> 4, pq = (lane >> 2) & 3;
    float vone = 1.0f;
    asm volatile("" : "+v"(vone));
    const QuadMasks qm = quad_masks();
    if (threadIdx.x == 0) { sA[FQ_NULL] = make_float4(0.f, 0.f, 0.f, 0.f); sB[FQ_NULL] = make_float4(0.f, 0.f, 0.f, 0.f); sC[FQ_NULL] = 0.f; }

    if (blockIdx.x >= heavy_wgs) {
        const uint32_t li = blockIdx.x - heavy_wgs;
        if (li >= n_light) {
            // the workgroups behind the tiles with instances share the EMPTY tiles -- the tail of tile_order
            const uint32_t nfill = gridDim.x - heavy_wgs - n_light, j = li - n_light;
            for (uint32_t i = n_all + j; i < n_tiles; i += nfill)
                fill_tile_background(s, s.tile_desc[i].x, threadIdx.x, W, H, gx, bg, out_color);
            return;
        }
        // ---- a light tile: one staging round, wave w walks four blocks (one of every block row and column) ----
        const uint4 td = s.light_desc[li];
        const uint32_t tile = td.x, n = td.z - td.y;        // 1 <= n < LIGHT_MAX <= FQ_CH
        const uint32_t tx = tile % gx, ty = tile / gx;
        stamp(s, tile, 0);
        if (threadIdx.x < n) {
            const uint32_t pos = td.y + threadIdx.x;
            float4 va = b.recA[pos], vb = b.recB[pos];
            const float vc = b.recC[pos].x; const uint2 vq = b.qmask[pos];
#if TGS_FAST_MATH
            stage_conic_a(va); stage_conic_b(vb);
#endif
            sA[threadIdx.x] = va; sB[threadIdx.x] = vb; sC[threadIdx.x] = vc; sQ[threadIdx.x] = vq;
        }
        __syncthreads();
        uint32_t wq = 0;
#pragma unroll 1
        for (int bi = 0; bi < 4; bi++) {
            const int bx = (wv + bi) & 3, by = bi, blk = 4 * by + bx;
            const int px = tx * TILE + bx * 4 + (qd & 1) * 2 + (pq & 1);
            const int py = ty * TILE + by * 4 + (qd >> 1) * 2 + (pq >> 1);
            const bool inside = px < W && py < H;
            bool done = !inside;
            if (__builtin_amdgcn_ballot_w64(!done) == 0) continue;      // a block outside the image
            float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
            uint32_t last_contributor = 0;
            fwd_q_block_round(sA, sB, sC, sQ, lists[wv], qlists[wv], n, blk, lane, 1u, (float)px, (float)py, vone, qm, done, T, C0, C1, C2, last_contributor);
            wq = max(wq, fwd_q_block_store(s, out_color, W, H, px, py, inside, lane, bg0, bg1, bg2, T, C0, C1, C2, last_contributor));
        }
        if (lane == 0) wave_qmax[wv] = wq;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t q = max(max(wave_qmax[0], wave_qmax[1]), max(wave_qmax[2], wave_qmax[3]));
            reinterpret_cast<uint32_t*>(&s.light_desc[li])[3] = q;             // deepest blended position of the tile: k_render_bwd's descriptor load brings it along (both copies)
            reinterpret_cast<uint32_t*>(&s.tile_desc[n_ne + li])[3] = q;
        }
        stamp(s, tile, 1);
        return;
    }
    // ---- a quarter of a tile with a long list: one wave per block ----
    if (ht8 >= n_ne) return;
    const uint32_t tile = tdh.x;
    const uint32_t tx = tile % gx, ty = tile / gx;
    const int bx = 2 * (int)(quarter & 1u) + (wv & 1), by = 2 * (int)(quarter >> 1) + (wv >> 1), blk = 4 * by + bx;
    const int px = tx * TILE + bx * 4 + (qd & 1) * 2 + (pq & 1);
    const int py = ty * TILE + by * 4 + (qd >> 1) * 2 + (pq >> 1);
    const bool inside = px < W && py < H;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint2 rg = make_uint2(tdh.y, tdh.z);
    set_wave_priority(rg.y - rg.x);
    if (quarter == 0) stamp(s, tile, 0);
    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_contributor = 0;
    // register-staged prefetch of the next round: thread t carries entry t whole
    float4 ra[FQ_EPT], rb[FQ_EPT]; float rc[FQ_EPT]; uint2 rq[FQ_EPT];
#pragma unroll
    for (int i = 0; i < FQ_EPT; i++) { ra[i] = make_float4(0.f, 0.f, 0.f, 0.f); rb[i] = ra[i]; rc[i] = 0.f; rq[i] = make_uint2(0u, 0u); }
#define TGS_FQ_FETCH(BASE) { _Pragma("unroll") for (int i = 0; i < FQ_EPT; i++) { const uint32_t pos = (BASE) + (uint32_t)(i * FQ_THREADS) + threadIdx.x; \
        if (pos < rg.y) { ra[i] = b.recA[pos]; rb[i] = b.recB[pos]; rc[i] = b.recC[pos].x; rq[i] = b.qmask[pos]; } } }
    TGS_FQ_FETCH(rg.x)
    bool wave_live = __builtin_amdgcn_ballot_w64(!done) != 0;
    int round = 0;
    for (uint32_t base = rg.x; base < rg.y; base += FQ_CH, round++) {
        // two barriers per round: (A) everybody has finished reading the previous round's LDS and has posted its liveness; (B) the new
        // round is staged.  The quarter stops when none of its waves has a live pixel (forward.cu:307-310).
        if (lane == 0) wave_alive[round & 1][wv] = wave_live ? 1u : 0u;
        __syncthreads();
        {
            const uint4 f = *reinterpret_cast<const uint4*>(wave_alive[round & 1]);
            if ((f.x | f.y | f.z | f.w) == 0u) break;
        }
        const uint32_t cnt = min((uint32_t)FQ_CH, rg.y - base);
#pragma unroll
        for (int i = 0; i < FQ_EPT; i++) {
            const uint32_t h = (uint32_t)(i * FQ_THREADS) + threadIdx.x;
            if (h < cnt) {
                float4 va = ra[i], vb = rb[i];
#if TGS_FAST_MATH
                stage_conic_a(va); stage_conic_b(vb);
#endif
                sA[h] = va; sB[h] = vb; sC[h] = rc[i]; sQ[h] = rq[i];
            }
        }
        __syncthreads();
        if (base + FQ_CH < rg.y) TGS_FQ_FETCH(base + FQ_CH)
        if (wave_live && fwd_q_block_round(sA, sB, sC, sQ, lists[wv], qlists[wv], cnt, blk, lane, base - rg.x + 1, pixfx, pixfy, vone, qm, done, T, C0, C1, C2, last_contributor))
            wave_live = false;
    }
#undef TGS_FQ_FETCH
    const uint32_t m = fwd_q_block_store(s, out_color, W, H, px, py, inside, lane, bg0, bg1, bg2, T, C0, C1, C2, last_contributor);
    if (lane == 0) wave_qmax[wv] = m;
    __syncthreads();
    // deepest blended position of the tile (its descriptor's .w, zero from k_scan): the maximum over the four quarters -- in BOTH copies of
    // the descriptor when the tile is a light one composited here by quarters (light == 0): a backward that sets light tiles apart reads light_desc
    if (threadIdx.x == 0) {
        const uint32_t q = max(max(wave_qmax[0], wave_qmax[1]), max(wave_qmax[2], wave_qmax[3]));
        atomicMax(reinterpret_cast<uint32_t*>(&s.tile_desc[ht8]) + 3, q);
        if (ht8 >= ff.w) atomicMax(reinterpret_cast<uint32_t*>(&s.light_desc[ht8 - ff.w]) + 3, q);
    }
    stamp_max(s, tile, 1);                                  // (diagnostic builds: the last quarter's end; a frame's stamps start at zero only in a fresh buffer)
}

// ---------------------------------------------------------------------------------------------
// k_mark_visible (rasterizer_impl.cu:54-66)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mark_visible(int P, const float* __restrict__ means3D, const float* __restrict__ view, uint8_t* present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float mx = means3D[3 * (size_t)idx], my = means3D[3 * (size_t)idx + 1], mz = means3D[3 * (size_t)idx + 2];
    const float z = view[2] * mx + view[6] * my + view[10] * mz + view[14];
    present[idx] = !(z <= 0.2f);
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
void launch_preprocess_fwd(hipStream_t st, const FwdIn& in, const CamParams& cam, const GeomState& g, const ImgState& s)
{
    const dim3 grid((unsigned)n_blocks(in.P)), blk(PRE_BLOCK);
    const bool sh = in.colors_precomp == nullptr, sr = in.cov3D_precomp == nullptr;
    FwdView vw;
    vw.cam = cam; vw.g = g; vw.s = s; vw.radii = in.radii;
    if (sh && sr) hipLaunchKernelGGL((k_preprocess_fwd<true, true>), grid, blk, 0, st, in, vw);
    else if (sh) hipLaunchKernelGGL((k_preprocess_fwd<true, false>), grid, blk, 0, st, in, vw);
    else if (sr) hipLaunchKernelGGL((k_preprocess_fwd<false, true>), grid, blk, 0, st, in, vw);
    else hipLaunchKernelGGL((k_preprocess_fwd<false, false>), grid, blk, 0, st, in, vw);
}
void launch_preprocess_fwd_batch(hipStream_t st, const FwdIn& in, const FwdViews& views)
{
    const dim3 grid((unsigned)n_blocks(in.P)), blk(PRE_BLOCK);
    const bool sh = in.colors_precomp == nullptr, sr = in.cov3D_precomp == nullptr;
    if (views.n == 2) {
        if (sh && sr) hipLaunchKernelGGL((k_preprocess_fwd_pair<true, true>), grid, blk, 0, st, in, views.v[0], views.v[1]);
        else if (sh) hipLaunchKernelGGL((k_preprocess_fwd_pair<true, false>), grid, blk, 0, st, in, views.v[0], views.v[1]);
        else if (sr) hipLaunchKernelGGL((k_preprocess_fwd_pair<false, true>), grid, blk, 0, st, in, views.v[0], views.v[1]);
        else hipLaunchKernelGGL((k_preprocess_fwd_pair<false, false>), grid, blk, 0, st, in, views.v[0], views.v[1]);
        return;
    }
    if (sh && sr) hipLaunchKernelGGL((k_preprocess_fwd_batch<true, true>), grid, blk, 0, st, in, views);
    else if (sh) hipLaunchKernelGGL((k_preprocess_fwd_batch<true, false>), grid, blk, 0, st, in, views);
    else if (sr) hipLaunchKernelGGL((k_preprocess_fwd_batch<false, true>), grid, blk, 0, st, in, views);
    else hipLaunchKernelGGL((k_preprocess_fwd_batch<false, false>), grid, blk, 0, st, in, views);
}
void launch_scan(hipStream_t st, const GeomState& g, const ImgState& s, uint32_t nblocks, uint32_t T, uint32_t sort_cap, unsigned long long r_capacity,
                 uint32_t tile_bound, uint32_t heavy_bound, uint32_t mid_bound, Meta* host_meta, int light)
{
    hipLaunchKernelGGL(k_scan, dim3(1u + (T + SCAN_THREADS - 1) / SCAN_THREADS), dim3(SCAN_THREADS), 0, st, g, s, nblocks, T, sort_cap, r_capacity, tile_bound, heavy_bound, mid_bound,
                       host_meta, light);
}
// Binning chunks: `nchunks` workgroups of BIN_THREADS threads, `chunk` Gaussians each (a multiple of BIN_THREADS)
void bin_shape(int P, uint32_t T, uint32_t& nchunks, uint32_t& chunk, uint32_t& band, size_t& lds)
{
    static const int bin_wgs = [] {                         // TGS_BIN_WGS: tuning knob, read once (thread-safe initialisation)
        const char* e = getenv("TGS_BIN_WGS");
        const int v = e ? atoi(e) : BIN_WGS_MAX;
        return v < 1 ? 1 : (v > BIN_WGS_MAX ? BIN_WGS_MAX : v);
    }();
    const uint32_t per = ((uint32_t)P + (uint32_t)bin_wgs - 1) / (uint32_t)bin_wgs;
    chunk = (per + BIN_THREADS - 1) / BIN_THREADS * BIN_THREADS;
    if (chunk == 0) chunk = BIN_THREADS;
    nchunks = ((uint32_t)P + chunk - 1) / chunk;
    band = T < BIN_LDS_TILES ? T : BIN_LDS_TILES;
    lds = (size_t)band * sizeof(uint32_t);
}
void launch_bin_count(hipStream_t st, int P, const GeomState& g, const ImgState& s, uint32_t gx, uint32_t T)
{
    uint32_t nchunks, chunk, band; size_t lds;
    bin_shape(P, T, nchunks, chunk, band, lds);
    if (lds > 32 * 1024) (void)hipFuncSetAttribute((const void*)k_bin_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_bin_count, dim3(nchunks), dim3(BIN_THREADS), lds, st, P, chunk, g, s, gx, T, band);
    hipLaunchKernelGGL(k_bin_colscan, dim3((T + WAVE - 1) / WAVE), dim3(64 * COL_WAVES), 0, st, s, T, nchunks);
}
void launch_scatter(hipStream_t st, int P, const GeomState& g, const ImgState& s, const BinState& b, uint32_t gx, uint32_t T)
{
    uint32_t nchunks, chunk, band; size_t lds;
    bin_shape(P, T, nchunks, chunk, band, lds);
    if (lds > 32 * 1024) (void)hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_scatter, dim3(nchunks), dim3(BIN_THREADS), lds, st, P, chunk, g, s, b, gx, T, band);
}
static uint32_t host_next_pow2(uint32_t n) { uint32_t p = 1; while (p < n) p <<= 1; return p; }
// Exact sizes (after the forward's read-back of Meta) or, with m == nullptr, upper bounds for the sync-free forward:
// workgroups beyond the device-side counts return at once.
// tile_bound (m == nullptr): the caller's bound on the tiles with instances (k_scan rejects a frame with more), or T
void launch_tile_sort(hipStream_t st, const GeomState& g, const ImgState& s, const BinState& b, uint32_t gx, uint32_t T, uint64_t r_bound,
                      const Meta* m, uint32_t sort_cap, uint32_t tile_bound, uint32_t heavy_bound, uint32_t mid_bound)
{
    if (!m && tile_bound < T) T = tile_bound;               // (T only bounds the class sizes below)
    const uint32_t max_count = m ? m->max_count : sort_cap;
    const uint32_t cap = max_count < sort_cap ? max_count : sort_cap;
    const size_t lds = (size_t)(cap ? cap : 1) * 8;
    const uint32_t nonempty = m ? m->n_nonempty : (uint32_t)(r_bound < T ? r_bound : T);
    uint32_t heavy = m ? m->n_heavy : (uint32_t)(r_bound / 1024 < T ? r_bound / 1024 : T);
    uint32_t mid = m ? m->n_mid : (uint32_t)(r_bound / 128 < T ? r_bound / 128 : T);     // tiles with >= 128 instances, heavy ones included
    if (!m && heavy > heavy_bound) heavy = heavy_bound;      // (k_scan has rejected a frame with more)
    if (!m && mid > mid_bound) mid = mid_bound;
    if (heavy > nonempty) heavy = nonempty;
    if (mid < heavy) mid = heavy;
    if (mid > nonempty) mid = nonempty;
    const uint32_t small = m ? nonempty - mid : nonempty;
    {
        const uint32_t mid_only = m ? mid - heavy : mid;    // (without the read-back both classes are bounded on their own)
        const uint32_t mid_blocks = (mid_only + 3) / 4, small_blocks = (small + 15) / 16;
        size_t bytes = 0;                                   // LDS for the largest class present: keys of one heavy tile / 4 x 1024 / 16 x 128
        if (heavy > 0) bytes = lds;
        if (mid_blocks > 0 && bytes < 4 * 1024 * 8) bytes = 4 * 1024 * 8;
        if (small_blocks > 0 && bytes < 16 * SORT_CHUNK * 8) bytes = 16 * SORT_CHUNK * 8;
        // workers for lists longer than sort_cap: none when the host knows there is no such list, otherwise (and always without the
        // read-back) OVF_WORKERS workgroups at the end of the grid -- they return at once when Meta says n_overflow == 0
        const uint32_t ovf_blocks = (m && m->n_overflow == 0) ? 0u : OVF_WORKERS;
        if (ovf_blocks > 0 && bytes < (size_t)sort_cap * 8) bytes = (size_t)sort_cap * 8;
        if (heavy + mid_blocks + small_blocks + ovf_blocks > 0)
            hipLaunchKernelGGL(k_tile_sort, dim3(heavy + mid_blocks + small_blocks + ovf_blocks), dim3(1024), bytes, st, g, s, b, gx, sort_cap, heavy, mid_blocks,
                               small_blocks, ovf_blocks);
    }
    if (r_bound > 0) hipLaunchKernelGGL(k_finalize, dim3((unsigned)((r_bound + 256 * FIN_E - 1) / (256 * FIN_E))), dim3(256), 0, st, g, s, b, gx, T);
}
// mid_bound (sync-free, with a tile bound): upper bound on the tiles with >= LIGHT_MAX instances (k_scan rejects a frame with more)
void launch_render_fwd(hipStream_t st, const ImgState& s, const BinState& b, int W, int H, uint32_t gx, uint32_t T, const Meta* m,
                       const float* bg, float* out_color, uint32_t tile_bound, uint32_t mid_bound, int light)
{
    // four workgroups per tile of the (bound on the) tiles with >= LIGHT_MAX instances -- all tiles with instances when no light tiles are
    // set apart --, one per light tile, one per 4 tiles beyond for the background.  m: the frame's counts when the host has read them; otherwise
    // the caller's bounds (k_scan has rejected a frame that exceeds them), surplus workgroups join the background fill or leave at once.
    const uint32_t tb = m ? m->n_nonempty : (tile_bound < T ? tile_bound : T);
    const uint32_t hb = light ? (m ? (m->n_mid < tb ? m->n_mid : tb) : (mid_bound < tb ? mid_bound : tb)) : tb;
    // (+ 7: with n_ne <= hb tiles in the first class the kernel needs 4 * roundup8(n_ne) + (n_all - n_ne) workgroups, and 4 * roundup8(x) - x
    // is not monotone inside a block of 8 -- n_ne = hb - 7 needs 7 more than n_ne = hb)
    const uint32_t grid = 4u * ((hb + 7u) & ~7u) + 7u + (tb - hb) + (T - tb + 3u) / 4u;
    if (grid > 0) hipLaunchKernelGGL(k_render_fwd, dim3(grid), dim3(FQ_THREADS), 0, st, s, b, W, H, gx, bg, out_color, T, light);
}
void launch_mark_visible(hipStream_t st, int P, const float* means3D, const float* view, uint8_t* present)
{
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, st, P, means3D, view, present);
}

}  // namespace tgs
