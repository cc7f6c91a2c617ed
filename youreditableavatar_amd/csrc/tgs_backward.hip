// tgs_backward.hip -- backward pass kernels for gfx950 (wave64).
//
//   k_render_bwd      back-to-front gradient of the compositing                      (backward.cu:399-557)
//   k_preprocess_bwd  per-Gaussian: sum of the tile partials, then cov2D / projection / SH / cov3D
//                     backward fused in one pass          (backward.cu:144-274 + :346-396, two kernels there)
//
// The reference adds 9 floats per (pixel, Gaussian) fragment with global atomicAdd
// (backward.cu:523,545-554).  Here all 64 lanes of a wave work on the SAME list entry at the same
// time, so the 9 partials are first summed across the wave with DPP row operations, then across the
// tile's 4 waves through LDS, and the tile stores ONE 36-byte row per (tile, Gaussian) instance with
// plain stores into a slab indexed in Gaussian order (row = offsets[g] + ordinal of the tile inside
// g's rectangle).  k_preprocess_bwd then sums each Gaussian's contiguous rows in a fixed order: no
// float atomics at all, and gradients are bitwise reproducible run to run (the reference's are not).
#include "tgs_device.hpp"

namespace tgs {

constexpr int NACC = 9;        // colour rgb, mean2D xy, conic xx/xy/yy, opacity

// The per-pixel backward visits the first gridDim.x tiles of tile_order -- all tiles, or the caller's bound on the tiles with instances
// (tgs_options_t::tile_bound).  A bound below the frame's real count would silently drop the remaining tiles' gradients: the first
// workgroup raises META_ERR_TILE_BOUND instead (tgs_frame_status then fails; TGS_FRAME_TILE_BOUND).
__device__ __forceinline__ void check_tile_bound(const ImgState& s)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && s.meta->n_nonempty > gridDim.x) atomicOr(&s.meta->error, META_ERR_TILE_BOUND);
}

// A tile's share of dL_dconic (one of xx, xy, yy) = -0.5 * opacity * (the f64 LDS sum of w d d over the tile's pixels), stored in the slab
// row as hi + lo (two floats: the row's padding holds the three lo parts -- no extra byte moves).  dL_dconic is the one input of the
// per-Gaussian chain that chain amplifies (a needle-shaped splat: x 1e2 .. 1e4 into dL_dcov3D / dL_dscale / dL_drot); with fp32 rows
// and an fp32 sum over a splat's tiles the HIP path sat 1e-4 from exact arithmetic on such splats -- as far as the reference's own
// fp32 atomics, but not the same way.  hi + lo rows summed in double (slab_sum) leave the per-pixel fp32 products as the only rounding.
struct ConicHiLo { float hi, lo; };
__device__ __forceinline__ ConicHiLo conic_hilo(float op, double sum)
{
    const double t = -0.5 * (double)op * sum;
    ConicHiLo r;
    r.hi = (float)t;
    r.lo = (float)(t - (double)r.hi);
    return r;
}

__global__ __launch_bounds__(256) void k_render_bwd_det(const ImgState s, const BinState b, int W, int H, uint32_t gx,
                                                    const float* __restrict__ bg, const float* __restrict__ dL_dpix)
{
    __shared__ float4 sA[RCHUNK + 1];
    __shared__ float4 sB[RCHUNK + 1];
    __shared__ float sC[RCHUNK + 1];
    __shared__ uint32_t sSlot[RCHUNK];
    __shared__ float wacc[4][NACC][RCHUNK + 1];            // per-wave partial sums of the current round (+1: null slot)
    __shared__ unsigned long long touched[4][RCHUNK / 64];
    __shared__ QuadLists L;
    __shared__ uint32_t wmax[4];

    if (frame_rejected(s)) return;
    check_tile_bound(s);
    const uint4 td = s.tile_desc[blockIdx.x];
    const uint32_t tile = td.x;
    const uint32_t tx = tile % gx, ty = tile / gx;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = tx * TILE + (wv & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint2 rg = make_uint2(td.y, td.z);
    const uint32_t n = rg.y - rg.x;
    if (n == 0) return;
    set_wave_priority(n);
    stamp(s, tile, 2);
    const size_t pix_id = (size_t)W * py + px, N = (size_t)W * H;
    if (threadIdx.x == 0) { sA[RNULL] = make_float4(0.f, 0.f, 0.f, 0.f); sB[RNULL] = make_float4(0.f, 0.f, 0.f, 0.f); sC[RNULL] = 0.f; }

    const float T_final = inside ? s.final_T[pix_id] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? s.n_contrib[pix_id] : 0u;
    float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f;
    if (inside) { dpx0 = dL_dpix[pix_id]; dpx1 = dL_dpix[N + pix_id]; dpx2 = dL_dpix[2 * N + pix_id]; }
    float bg_dot_dpixel = 0.f;                              // backward.cu:533-535
    bg_dot_dpixel += bg[0] * dpx0; bg_dot_dpixel += bg[1] * dpx1; bg_dot_dpixel += bg[2] * dpx2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;               // accum_rec
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);   // backward.cu:460-461

    // Entries behind every pixel's last contributor get no gradient (backward.cu:487-488): find the
    // deepest one any pixel of the tile needs and start there.
    uint32_t mq = wave_max_u32(last_contributor);
    if (lane == 0) wmax[wv] = mq;
    __syncthreads();
    const uint32_t qmax = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));

    // rows of the never-visited tail are zero
    for (uint32_t q = qmax + threadIdx.x; q < n; q += 256) {
        float4* row = b.slab + (size_t)b.slot[rg.x + q] * SLAB_ROW;
        row[0] = make_float4(0.f, 0.f, 0.f, 0.f); row[1] = make_float4(0.f, 0.f, 0.f, 0.f); row[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // register-staged prefetch (slot t of a round = list position qhi-1-t: back to front)
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    float2 rc = make_float2(0.f, 0.f);
    uint32_t rs = 0;
    if (threadIdx.x < qmax) { const uint32_t pos = rg.x + qmax - 1 - threadIdx.x; ra = b.recA[pos]; rb = b.recB[pos]; rc = b.recC[pos]; rs = b.slot[pos]; }

    for (uint32_t qhi = qmax; qhi > 0; qhi = qhi > RCHUNK ? qhi - RCHUNK : 0) {
        const uint32_t cnt = min((uint32_t)RCHUNK, qhi);
        __syncthreads();                                    // previous round's flush has read wacc / sSlot
        uint32_t qm = 0;
        if (threadIdx.x < cnt) { sA[threadIdx.x] = ra; sB[threadIdx.x] = rb; sC[threadIdx.x] = rc.x; qm = block_to_quadrant_mask(__float_as_uint(rc.y)); sSlot[threadIdx.x] = rs; }
        build_quad_lists(L, qm, wv, lane);
        if (lane < RCHUNK / 64) touched[wv][lane] = 0ull;
        __syncthreads();
        if (qhi > RCHUNK && threadIdx.x < qhi - RCHUNK) {
            const uint32_t pos = rg.x + qhi - RCHUNK - 1 - threadIdx.x;
            ra = b.recA[pos]; rb = b.recB[pos]; rc = b.recC[pos]; rs = b.slot[pos];
        }

#pragma unroll 1
        for (int sw = 0; sw < 4; sw++) {
            const uint32_t nl = __builtin_amdgcn_readfirstlane(L.cnt[wv][sw]);
            unsigned long long tmask = 0;
#pragma unroll 1
            for (uint32_t k = 0; k < nl; k += RUNROLL) {    // only entries that can reach this wave's quadrant
                const uint2 pk = *reinterpret_cast<const uint2*>(&L.idx[wv][sw][k]);
                const uint32_t j[RUNROLL] = {pk.x & 0xffffu, pk.x >> 16, pk.y & 0xffffu, pk.y >> 16};
                float4 a[RUNROLL], bb[RUNROLL];
                float cc[RUNROLL], dx[RUNROLL], dy[RUNROLL], G[RUNROLL], alpha[RUNROLL];
                bool valid[RUNROLL];
#pragma unroll
                for (int u = 0; u < RUNROLL; u++) { a[u] = sA[j[u]]; bb[u] = sB[j[u]]; cc[u] = sC[j[u]]; }
                bool any = false;
#pragma unroll
                for (int u = 0; u < RUNROLL; u++) {
                    dx[u] = a[u].x - pixfx; dy[u] = a[u].y - pixfy;
#if TGS_FAST_MATH
                    // alpha exactly as k_render_fwd evaluated it (pair_power2 on the conic scaled as stage_conic_* scales it, v_exp_f32): the backward replays
                    // the forward's alpha >= 1/255 decisions, so it must round what the forward rounded (round 6; until then this kernel evaluated the
                    // reference's expression with expf -- another rounding than the forward's, the same hazard pair_power2 removed from the default kernel)
                    const float power = pair_power2(a[u].z * (-0.5f * LOG2E), a[u].w * (-LOG2E), bb[u].x * (-0.5f * LOG2E), dx[u], dy[u]);
                    G[u] = __builtin_amdgcn_exp2f(power);
#else
                    const float power = -0.5f * (a[u].z * dx[u] * dx[u] + bb[u].x * dy[u] * dy[u]) - a[u].w * dx[u] * dy[u];
                    G[u] = tgs_exp(power);
#endif
                    alpha[u] = fminf(0.99f, bb[u].y * G[u]);
                    // list position of slot j is qhi-1-j; "contributor >= last_contributor" skip of backward.cu:487
                    valid[u] = (qhi - 1 - j[u] < last_contributor) && (j[u] < cnt) && !(power > 0.0f) && !(alpha[u] < 1.0f / 255.0f);
                    any = any || valid[u];
                }
                if (__builtin_amdgcn_ballot_w64(any) == 0) continue;
                float v[36];
#pragma unroll
                for (int u = 0; u < RUNROLL; u++) {
#pragma unroll
                    for (int c = 0; c < NACC; c++) v[u * NACC + c] = 0.f;
                    if (valid[u]) {                         // backward.cu:507-555
                        const float om = 1.f - alpha[u];
                        T = tgs_div(T, om);
                        const float dchannel_dcolor = alpha[u] * T;
                        float dL_dalpha = 0.0f;
                        const float c0 = bb[u].z, c1 = bb[u].w, c2 = cc[u];
                        acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = c0; dL_dalpha += (c0 - acc0) * dpx0;
                        acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = c1; dL_dalpha += (c1 - acc1) * dpx1;
                        acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = c2; dL_dalpha += (c2 - acc2) * dpx2;
                        v[u * NACC + 0] = dchannel_dcolor * dpx0; v[u * NACC + 1] = dchannel_dcolor * dpx1; v[u * NACC + 2] = dchannel_dcolor * dpx2;
                        dL_dalpha *= T;
                        last_alpha = alpha[u];
                        dL_dalpha += tgs_div(-T_final, om) * bg_dot_dpixel;
                        const float dL_dG = bb[u].y * dL_dalpha;
                        const float gdx = G[u] * dx[u], gdy = G[u] * dy[u];
                        const float dG_ddelx = -gdx * a[u].z - gdy * a[u].w;
                        const float dG_ddely = -gdy * bb[u].x - gdx * a[u].w;
                        v[u * NACC + 3] = dL_dG * dG_ddelx * ddelx_dx;
                        v[u * NACC + 4] = dL_dG * dG_ddely * ddely_dy;
                        v[u * NACC + 5] = -0.5f * gdx * dx[u] * dL_dG;
                        v[u * NACC + 6] = -0.5f * gdx * dy[u] * dL_dG;
                        v[u * NACC + 7] = -0.5f * gdy * dy[u] * dL_dG;
                        v[u * NACC + 8] = G[u] * dL_dalpha;
                    }
                }
                float r[NACC];
                wave_reduce36(v, r);                        // row e of r[k]: total of entry e, component k
                // lane 16e stores entry e's nine sums (entries with no valid lane store zeros; null slots go to the spare column)
                const int row = lane >> 4;
                const uint32_t jr = row == 0 ? j[0] : row == 1 ? j[1] : row == 2 ? j[2] : j[3];
                if ((lane & 15) == 0) {
#pragma unroll
                    for (int c = 0; c < NACC; c++) wacc[wv][c][jr] = r[c];
                }
#pragma unroll
                for (int u = 0; u < RUNROLL; u++) if (j[u] < RCHUNK) tmask |= 1ull << (j[u] & 63);
            }
            if (lane == 0 && tmask) touched[wv][sw] = tmask;
        }
        __syncthreads();
        // flush: thread j adds the (up to) 4 wave partials of entry j in wave order and stores the row
        if (threadIdx.x < cnt) {
            const uint32_t j = threadIdx.x;
            float r[NACC];
            double rc[3] = {0.0, 0.0, 0.0};                  // the conic sums: hi + lo (slab row layout, tgs_device.hpp)
#pragma unroll
            for (int k = 0; k < NACC; k++) r[k] = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if ((touched[w][j >> 6] >> (j & 63)) & 1ull) {
#pragma unroll
                    for (int k = 0; k < NACC; k++) r[k] += wacc[w][k][j];
#pragma unroll
                    for (int k = 0; k < 3; k++) rc[k] += (double)wacc[w][5 + k][j];
                }
            }
            float lo[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { r[5 + k] = (float)rc[k]; lo[k] = (float)(rc[k] - (double)r[5 + k]); }
            float4* row = b.slab + (size_t)sSlot[j] * SLAB_ROW;
            row[0] = make_float4(r[0], r[1], r[2], r[3]); row[1] = make_float4(r[4], r[5], r[6], r[7]); row[2] = make_float4(r[8], lo[0], lo[1], lo[2]);
        }
    }
    stamp(s, tile, 3);
}

// ---------------------------------------------------------------------------------------------
// k_render_bwd (default): same geometry as k_render_fwd -- 16 waves per tile, one wave per 4x4-pixel
// block, every 16-lane row one 2x2-pixel quadrant walking its own list (back to front), 4 pixels x 4
// consecutive entries per row and pass.  Every lane evaluates its own (pixel, entry) pair; the 4 lanes
// of a quad walk the pixel's sequential state (T and dL_dpixel . accum_rec -- the one scalar of
// backward.cu:505-531's three accum_rec channels that dL_dalpha needs; bwd_chain4s) through the group's 4
// entries with DPP quad broadcasts and each lane keeps the state of its own step; a skipped entry is walked as alpha = 0, which leaves T and every later
// accum_rec value bit-identical to not visiting it
// (acc' = la*lc + (1-la)*acc, (la,lc) <- (0,c);  next: 0*c + 1*acc' = acc').
// The nine per-entry sums over the quadrant's 4 pixels are formed with DPP row rotations (18
// v_add_f32_dpp), then every row adds them to the tile's per-round accumulator in LDS -- in f64:
// ds_add_f32 is a per-lane loop on gfx950 (~3 clk per active lane), ds_add_f64 runs at the rate of a
// plain LDS access (tools/microbench/lds_atomic_rate.hip).  Up to 64 adds meet in one entry, so the
// in-tile summation order can vary run to run, but the f32 terms are summed in f64 and rounded once at
// the flush: differences are rare last-bit events.  k_render_bwd_det above keeps a fixed order
// (bitwise reproducible) at about 2.5x the time; tgs_set_deterministic(1) selects it.
// ---------------------------------------------------------------------------------------------
constexpr int BWD_THREADS = 1024;
#ifndef TGS_BWD_BOUNDED
#define TGS_BWD_BOUNDED true
#endif
#ifndef TGS_BCH
#define TGS_BCH 384
#endif
constexpr int BCH = TGS_BCH;               // list entries per round
constexpr int BNULL = BCH;

// ---------------------------------------------------------------------------------------------
// The passes of one wave over the current chunk's quadrant lists (heavy path and light groups alike): every 16-lane row takes 4 entries of ITS list
// per pass.  Round 5 -- inside a pass every vector instruction counts (profiles/r05_render_decomposition.txt):
//   * the lists hold byte offsets into the 16-byte record arrays (slot << 4): no shift in front of the record reads, one for the 4- and 8-byte arrays;
//   * "list position in front of the pixel's last contributor" (backward.cu:487) is ONE signed compare of the offset with a per-lane threshold;
//   * the cross-pixel sums arrive in the lanes that add them (row_stride4_sum9_banked): no selects.
// ---------------------------------------------------------------------------------------------
// Per-round sums of a tile's staged entries, f64 (ds_add_f64 runs ~20x the rate of ds_add_f32 on gfx950), in 16-BYTE cells like the records, so that
// a list's byte offset (slot << 4) addresses them without a shift: cell v[i][j] holds components i and 4 + i of entry j -- the two sums the lane
// of pixel i of a quadrant adds (row_stride4_sum9_banked) --, cell t[j] component 8 and, in its spare half, the record's colour b (the ninth float
// of a record; the other eight are sA / sB).  Column BNULL swallows the padding entries.
struct AccTail { double w; float cb; uint32_t pad; };
template <int CH>
struct alignas(16) BwdAccT {
    static constexpr int PLANE = 2 * (CH + 1) + 1; // doubles per pixel-of-quadrant plane: an ODD number of 8-byte slots, so that the four planes' cells of one
                                                   // entry fall into both halves of the 16-byte bank groups (with an even stride the 64 lanes' atomics used every
                                                   // second 8-byte slot: twice the bank conflicts of the old 8-byte layout, measured + 1 us)
    double v[4][PLANE]; AccTail t[CH + 1];
};
using BwdAcc = BwdAccT<BCH>;
constexpr int ACC_PLANE = BwdAcc::PLANE;
static_assert(sizeof(AccTail) == 16 && offsetof(BwdAcc, t) == 4 * ACC_PLANE * 8 && offsetof(BwdAcc, t) % 16 == 0, "BwdAcc: 16-byte cells");
template <int CH> __device__ __forceinline__ double acc_get(const BwdAccT<CH>& A, int c, uint32_t j) { return c < 8 ? A.v[c & 3][2 * j + (c >> 2)] : A.t[j].w; }
template <int CH> __device__ __forceinline__ void acc_clear(BwdAccT<CH>& A, uint32_t j)
{
#pragma unroll
    for (int i = 0; i < 4; i++) { A.v[i][2 * j] = 0.0; A.v[i][2 * j + 1] = 0.0; }
    A.t[j].w = 0.0;
}
// (zero every sum of the accumulator, all threads of the workgroup; the colour halves of the tail cells are not touched)
template <int CH> __device__ __forceinline__ void acc_clear_all(BwdAccT<CH>& A, uint32_t tid, uint32_t nthreads)
{
    double* v = &A.v[0][0];
    for (uint32_t i = tid; i < 4u * BwdAccT<CH>::PLANE; i += nthreads) v[i] = 0.0;
    for (uint32_t i = tid; i < (uint32_t)(CH + 1); i += nthreads) A.t[i].w = 0.0;
}
struct alignas(16) BwdShared {
    float4 sA[BCH + 1];                                    // staged records: mean2D, conic xx / xy (pre-scaled for exp2); slot BNULL = the null record
    float4 sB[BCH + 1];                                    // conic yy, opacity, colour r g
    BwdAcc acc;                                            // per-round sums + colour b (above)
    uint32_t sSlot[2][BCH];                                // double-buffered, like sFl: the flush of round r overlaps the staging of r+1,
    float2 sFl[2][BCH];                                    // and these are written by the OTHER half of the workgroup: (conic yy, opacity) for the flush
    uint2 sQ[BCH];                                         // quadrant masks of the staged entries
    alignas(16) unsigned short lists[16][BCH + 8];         // one list per block (= per wave): slot | quadrant nibble << 10
    alignas(16) unsigned short qlists[16][4][QL_ROW];      // per wave: the current chunk's four quadrant lists (byte offsets: slot << 4)
};
static_assert(offsetof(BwdShared, sSlot) < 65536 && sizeof(BwdShared) <= 80 * 1024, "k_render_bwd: records + accumulator inside a 16-bit LDS offset; two workgroups per CU");
struct BwdPixel { float fx, fy, d0, d1, d2, tfinal_bg; int thr16; uint32_t acc_off; };   // per lane: pixel centre, dL_dpixel, T_final * (bg . dL_dpixel), threshold, byte offset of acc[pixel-of-quadrant]
template <int CH>
__device__ __forceinline__ void bwd_passes(const unsigned short* myq, uint32_t nq, const float4* sA, const float4* sB, BwdAccT<CH>& acc, uint32_t base16,
                                           const BwdPixel& px, bool first_bank, float& T, float& arA, float vone, float vzero, const QuadMasks& qm)
{
    const char* cA = reinterpret_cast<const char*>(sA);
    const char* cB = reinterpret_cast<const char*>(sB);
    char* cAcc = reinterpret_cast<char*>(&acc);
    // One pass over the entries at list offset jl16 of every row (skipped as a whole when no lane has anything to add).
    auto pass = [&](const uint32_t jl16) __attribute__((always_inline)) {
        const uint32_t j16 = jl16 + base16;
        const float4 a = *reinterpret_cast<const float4*>(cA + j16);       // mean2D, conic xx / xy pre-scaled for exp2 (stage_conic)
        const float4 bb = *reinterpret_cast<const float4*>(cB + j16);      // conic yy pre-scaled, opacity, colour r g
        const float c0 = bb.z, c1 = bb.w, c2 = *reinterpret_cast<const float*>(cAcc + offsetof(BwdAccT<CH>, t) + offsetof(AccTail, cb) + j16);
        const float dx = a.x - px.fx, dy = a.y - px.fy;
        const float power2 = pair_power2(a.z, a.w, bb.x, dx, dy);            // log2(e) * power of forward.cu:336, rounded exactly as k_render_fwd rounds it
        const float G = __builtin_amdgcn_exp2f(power2);
        const float alpha = fminf(0.99f, bb.y * G);
        // "contributor >= last_contributor" skip of backward.cu:487 as offset > threshold.  A padding entry has opacity 0 and fails the alpha test.
        const bool valid = ((int)jl16 > px.thr16) && !(power2 > 0.0f) && !(alpha < 1.0f / 255.0f);
        if (__builtin_amdgcn_ballot_w64(valid) == 0) return;
        const float aeff = valid ? alpha : 0.f;             // a skipped entry is walked as alpha = 0, G = 0
        const float Geff = valid ? G : 0.f;
        // the quad walks the pixel's state through the group's 4 entries (bwd_chain4s, tgs_device.hpp)
        float Town, inv_om, Aown;
        float sdot = c0 * px.d0;
        sdot += c1 * px.d1; sdot += c2 * px.d2;             // dL_dpixel . colour of this entry
        bwd_chain4s(aeff, sdot, T, arA, Town, inv_om, Aown, vone, vzero, qm);
        // this lane's (pixel, entry) terms, backward.cu:507-555 (all zero for a skipped entry).  Everything that is constant per entry --
        // opacity, the conic, -0.5, the ndc scale -- is applied once per entry at the flush, so a lane only forms the moments of
        // w = G * dL_dalpha over dx, dy.
        const float dchannel_dcolor = aeff * Town;
        float dL_dalpha = sdot - Aown;                      // sum_ch (c_ch - accum_rec_ch) * dL_dpixel_ch
        dL_dalpha = dL_dalpha * Town - px.tfinal_bg * inv_om;   // ... + (-T_final / (1 - alpha)) * bg_dot_dpixel
        const float w = Geff * dL_dalpha;
        const float wdx = w * dx, wdy = w * dy;
        float v[NACC];
        v[0] = dchannel_dcolor * px.d0; v[1] = dchannel_dcolor * px.d1; v[2] = dchannel_dcolor * px.d2;
        v[3] = wdx; v[4] = wdy;
        v[5] = wdx * dx; v[6] = wdx * dy; v[7] = wdy * dy;
        v[8] = w;
        float s0, s1;
        row_stride4_sum9_banked(v, s0, s1);                 // lane (pixel i, slot e): s0 = component i, s1 = component 4 + i of entry e; v[8] complete
        // rows whose list is shorter than the longest of the chunk idle on the null record: their sums are zero, and without this
        // test all of them would add into the ONE spare column -- same-address LDS atomics serialise
        if (j16 != (uint32_t)CH * 16u) {
            double* p = reinterpret_cast<double*>(cAcc + px.acc_off + j16);
            atomicAdd(p, (double)s0);
            atomicAdd(p + 1, (double)s1);
            if (first_bank) atomicAdd(reinterpret_cast<double*>(cAcc + offsetof(BwdAccT<CH>, t) + j16), (double)v[8]);
        }
    };
    // Two passes per trip, the offsets one pass ahead in alternating registers (null slots behind the list's end, up to QL_ROW): no register move
    // and one address step per two passes.  The empty asm keeps an offset a 32-bit value: without it the compiler carries the 16-bit load through
    // the loop and masks it with 0xffff in front of every use -- ds_read_u16 has zero-extended it already.
    nq = (uint32_t)__builtin_amdgcn_readfirstlane((int)nq);         // (the same in every lane: the loop test is scalar)
    uint32_t ja = myq[0];
    asm("" : "+v"(ja));
#pragma unroll 1
    for (uint32_t k = 0; k < nq; k += 8) {
        uint32_t jb = myq[k + 4];
        asm("" : "+v"(jb));
        pass(ja);
        if (k + 4 >= nq) break;
        ja = myq[k + 8];
        asm("" : "+v"(ja));
        pass(jb);
    }
}

// ---------------------------------------------------------------------------------------------
// Light groups of k_render_bwd: THREE light tiles (fewer than LIGHT_MAX instances, one round) per workgroup -- the staging arrays hold
// BCH = 384 = 3 x 128 entries --, tile q on waves 4q .. 4q+3 (waves 12..15 only meet the barriers), wave w of a tile walking its blocks
// 4w .. 4w+3 one after the other (the forward gives a light tile a 256-thread workgroup of its own: k_render_fwd; why light tiles are set apart: tgs_device.hpp, LIGHT_MAX).  Same arithmetic per (pixel, entry) pair, same f64
// LDS accumulator and flush as the heavy path; the pixel inputs of a wave's four blocks are fetched up front.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bwd_light_group(const ImgState& s, const BinState& b, int W, int H, uint32_t gx, const float* __restrict__ bg,
                                                const float* __restrict__ dL_dpix, uint4 td, bool active, float4* sA, float4* sB, uint32_t* sSlot,
                                                BwdAcc& acc, uint2* sQ, unsigned short (*lists)[BCH + 8], unsigned short (*qlists)[4][QL_ROW])
{
    const int sub = threadIdx.x >> 8, lt = threadIdx.x & 255;
    const int wv = threadIdx.x >> 6, w4 = wv & 3, lane = threadIdx.x & 63;
    const int qd = lane >> 4, pq = (lane >> 2) & 3, e = lane & 3;
    const uint32_t base = (uint32_t)LIGHT_MAX * (sub < BWD_LIGHT_PER_WG ? sub : 0);
    const uint32_t tile = td.x, n = active ? td.z - td.y : 0u;
    const uint32_t tx = tile % gx, ty = tile / gx;
    stamp_if(s, tile, 2, active && lt == 0);
    const size_t N = (size_t)W * H;
    const uint32_t qmax = active ? min(td.w, n) : 0u;                  // deepest position any pixel of the tile blended (k_render_fwd wrote it into the descriptor)
    // pixel inputs of a block (5 loads), asked for ONE BLOCK AHEAD: block 0's here, block bi + 1's in front of block bi's passes (round 5: all four
    // blocks' inputs held at once were 20 VGPRs -- with the pass's own registers the light path spilled them right behind their loads)
    struct PixIn { float Tf, d0, d1, d2; uint32_t lc; bool in; };     // (raw loads from a clamped address; select_loaded at the use)
    auto load_block = [&](int bi) {
        const int blk = 4 * w4 + bi;
        const int px = tx * TILE + (blk & 3) * 4 + (qd & 1) * 2 + (pq & 1);
        const int py = ty * TILE + (blk >> 2) * 4 + (qd >> 1) * 2 + (pq >> 1);
        const bool inside = active && px < W && py < H;
        const size_t pix_id = inside ? (size_t)W * py + px : 0;          // (clamped, not predicated: the five loads are in flight together)
        PixIn r;
        r.in = inside;
        r.Tf = s.final_T[pix_id];
        r.lc = inside ? s.n_contrib[pix_id] : 0u;
        r.d0 = dL_dpix[pix_id]; r.d1 = dL_dpix[N + pix_id]; r.d2 = dL_dpix[2 * N + pix_id];
        return r;
    };
    PixIn nxt = load_block(0);
    // rows of the never-visited tail are zero
    for (uint32_t q = qmax + lt; q < n; q += 256) {
        float4* row = b.slab + (size_t)b.slot[td.y + q] * SLAB_ROW;
        row[0] = make_float4(0.f, 0.f, 0.f, 0.f); row[1] = make_float4(0.f, 0.f, 0.f, 0.f); row[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {   // stage back to front: slot t = list position qmax-1-t.  Thread t < 128 of the quarter: recA + quadrant mask, thread 128 + t: recB + recC + slot
        const uint32_t ht = lt & (LIGHT_MAX - 1);
        const bool upper = lt >= LIGHT_MAX;
        if (ht < qmax) {
            const uint32_t pos = td.y + qmax - 1 - ht;
            if (!upper) { float4 r4 = b.recA[pos]; const uint2 q = b.qmask[pos]; stage_conic_a(r4); sA[base + ht] = r4; sQ[base + ht] = q; }
            else { float4 r4 = b.recB[pos]; const float c = b.recC[pos].x; const uint32_t sl = b.slot[pos]; stage_conic_b(r4); sB[base + ht] = r4; acc.t[base + ht].cb = c; sSlot[base + ht] = sl; }
        }
        acc_clear_all(acc, threadIdx.x, BWD_THREADS);
    }
    __syncthreads();                                        // (the null record was written in front of the kernel's first barrier)
    float vone = 1.0f, vzero = 0.0f;
    const QuadMasks qm = quad_masks();
    asm volatile("" : "+v"(vone), "+v"(vzero));
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);   // backward.cu:460-461
    const float bgr = bg[0], bgg = bg[1], bgb = bg[2];
    const unsigned short* myq = &qlists[wv][qd][e];
    const uint32_t null_local = (uint32_t)BNULL - base;
    if (qmax > 0) {
#pragma unroll
    for (int bi = 0; bi < 4; bi++) {
        const int blk = 4 * w4 + bi;
        const int px = tx * TILE + (blk & 3) * 4 + (qd & 1) * 2 + (pq & 1);
        const int py = ty * TILE + (blk >> 2) * 4 + (qd >> 1) * 2 + (pq >> 1);
        const float pixfx = (float)px, pixfy = (float)py;
        const PixIn cur = nxt;
        if (bi < 3) nxt = load_block(bi + 1);
        asm volatile("" ::: "memory");                      // (the next block's loads stay here, in front of this block's passes)
        const float T_final = select_loaded(cur.in, cur.Tf), dpx0 = select_loaded(cur.in, cur.d0), dpx1 = select_loaded(cur.in, cur.d1), dpx2 = select_loaded(cur.in, cur.d2);
        const uint32_t last_contributor = cur.lc;
        float T = T_final;
        float bg_dot_dpixel = 0.f;                          // backward.cu:533-535
        bg_dot_dpixel += bgr * dpx0; bg_dot_dpixel += bgg * dpx1; bg_dot_dpixel += bgb * dpx2;
        const float tfinal_bg = T_final * bg_dot_dpixel;
        float arA = 0.f;
        const BwdPixel pxl = {pixfx, pixfy, dpx0, dpx1, dpx2, tfinal_bg, ((int)qmax - 1 - (int)last_contributor) * 16, (uint32_t)pq * (uint32_t)(ACC_PLANE * 8)};
        uint32_t qlast[4];
        {
            uint32_t m = last_contributor;
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xf, 0xf, false));     // row_ror:4
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, false));     // row_ror:8
#pragma unroll
            for (int q = 0; q < 4; q++) qlast[q] = (uint32_t)__builtin_amdgcn_readlane((int)m, 16 * q);
        }
        if ((qlast[0] | qlast[1] | qlast[2] | qlast[3]) == 0u) continue;     // nothing was blended into this block
        const uint32_t nl = build_own_list_q<LIGHT_MAX>(lists[wv], sQ + base, qmax, blk, lane);
#pragma unroll 1
        for (uint32_t c0 = 0; c0 < nl; c0 += QCH) {
            const uint32_t nq = build_chunk_quadrant_lists<TGS_BWD_BOUNDED, 4>(qlists[wv], lists[wv], c0, nl, lane, (int)null_local, qmax - 1, qlast);
            bwd_passes(myq, nq, sA, sB, acc, base * 16u, pxl, pq == 0, T, arA, vone, vzero, qm);
        }
    }
    }
    __syncthreads();
    if (lt < qmax) {                                        // flush: one 48-B row per instance (flush of the heavy path; op and conic yy from the staged record)
        const uint32_t j = base + lt;
        const float4 a = sA[j], bb = sB[j];
        const float cxx = a.z * UNSCALE_CONIC, cxy = a.w * UNSCALE_CONIC_XY, cyy = bb.x * UNSCALE_CONIC, op = bb.y;
        const float Sx = (float)acc_get(acc, 3, j), Sy = (float)acc_get(acc, 4, j);
        float4* row = b.slab + (size_t)sSlot[j] * SLAB_ROW;
        row[0] = make_float4((float)acc_get(acc, 0, j), (float)acc_get(acc, 1, j), (float)acc_get(acc, 2, j), op * (-Sx * cxx - Sy * cxy) * ddelx_dx);
        const ConicHiLo c5 = conic_hilo(op, acc_get(acc, 5, j)), c6 = conic_hilo(op, acc_get(acc, 6, j)), c7 = conic_hilo(op, acc_get(acc, 7, j));
        row[1] = make_float4(op * (-Sy * cyy - Sx * cxy) * ddely_dy, c5.hi, c6.hi, c7.hi);
        row[2] = make_float4((float)acc_get(acc, 8, j), c5.lo, c6.lo, c7.lo);
    }
    stamp_if(s, tile, 3, active && lt == 0);
}

// light != 0: tiles with fewer than LIGHT_MAX instances are composited three per workgroup by the LAST workgroups of the grid (bwd_light_group);
// this kernel's one-tile workgroups end at Meta::n_mid
__global__ __launch_bounds__(BWD_THREADS, 8) void k_render_bwd(const ImgState s, const BinState b, int W, int H, uint32_t gx,
                                                            const float* __restrict__ bg, const float* __restrict__ dL_dpix, int light, uint32_t n_tiles)
{
    // ONE object, members in the order of their use in a pass: the arrays a pass addresses (records, accumulator) sit in the first 64 KB of the
    // workgroup's LDS, so their base is an instruction's 16-bit offset and not a VGPR (the compiler orders separate __shared__ arrays by size)
    __shared__ BwdShared S;
    auto& sA = S.sA; auto& sB = S.sB; auto& sSlot = S.sSlot; auto& sFl = S.sFl; auto& acc = S.acc; auto& sQ = S.sQ; auto& lists = S.lists; auto& qlists = S.qlists;

    const uint4 td = s.tile_desc[blockIdx.x];               // (in flight beside the frame's flags)
    // candidate light tile of this thread's quarter: light group g is workgroup gridDim.x - 1 - g and takes light tiles 3 g .. 3 g + 2
    // (every forward records the light tiles' descriptors -- k_scan, k_render_fwd --, so a backward may set them apart whatever its forward did)
    const uint32_t lgroup = gridDim.x - 1u - blockIdx.x, lsub = threadIdx.x >> 8, li = (uint32_t)BWD_LIGHT_PER_WG * lgroup + lsub;
    const uint4 tdl = (light && lsub < (uint32_t)BWD_LIGHT_PER_WG && li < n_tiles) ? s.light_desc[li] : make_uint4(0u, 0u, 0u, 0u);
    const uint4 ff = frame_counts(s);
    if (ff.x & META_ERR_CAPACITY) return;
    if (light) {
        const uint32_t n_ne = min(ff.w, ff.y), n_light = ff.y - n_ne, n_lgroups = (n_light + BWD_LIGHT_PER_WG - 1) / BWD_LIGHT_PER_WG;
        // (check_tile_bound: the grid must hold the one-tile workgroups and the light groups side by side)
        if (blockIdx.x == 0 && threadIdx.x == 0 && n_ne + n_lgroups > gridDim.x) atomicOr(&s.meta->error, META_ERR_TILE_BOUND);
        if (blockIdx.x >= n_ne) {
            if (lgroup >= n_lgroups) return;                // (uniform over the workgroup)
            if (threadIdx.x == 0) { sA[BNULL] = make_float4(0.f, 0.f, 0.f, 0.f); sB[BNULL] = make_float4(0.f, 0.f, 0.f, 0.f); acc.t[BNULL].cb = 0.f; }
            bwd_light_group(s, b, W, H, gx, bg, dL_dpix, tdl, lsub < (uint32_t)BWD_LIGHT_PER_WG && li < n_light, sA, sB, sSlot[0], acc, sQ, lists, qlists);
            return;
        }
    } else check_tile_bound(s);
    const uint32_t tile = td.x;
    const uint32_t tx = tile % gx, ty = tile / gx;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qd = lane >> 4, pq = (lane >> 2) & 3, e = lane & 3;   // DPP row = 2x2 quadrant of the block, pixel of the quadrant, entry slot
    const int px = tx * TILE + (wv & 3) * 4 + (qd & 1) * 2 + (pq & 1);
    const int py = ty * TILE + (wv >> 2) * 4 + (qd >> 1) * 2 + (pq >> 1);
    const bool inside = px < W && py < H;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint2 rg = make_uint2(td.y, td.z);
    const uint32_t n = rg.y - rg.x;
    if (n == 0) return;
    set_wave_priority(n);
    stamp(s, tile, 2);
    const size_t pix_id = (size_t)W * py + px, N = (size_t)W * H;
    if (threadIdx.x == 0) { sA[BNULL] = make_float4(0.f, 0.f, 0.f, 0.f); sB[BNULL] = make_float4(0.f, 0.f, 0.f, 0.f); acc.t[BNULL].cb = 0.f; }

    const uint32_t qmax = min(td.w, n);                     // deepest position any pixel of the tile blended (k_render_fwd wrote it into the descriptor: no dependent load)
    // Register-staged prefetch of the next round, split over the two halves of the workgroup so that it costs 6 VGPRs, not 11
    // (64 VGPRs keep two workgroups per CU): thread t < BCH carries recA + the quadrant mask of entry t, thread BCH + t recB + recC + slot.
    const bool upper = threadIdx.x >= BCH;
    const uint32_t ht = threadIdx.x < 2 * BCH ? (upper ? threadIdx.x - BCH : threadIdx.x) : 0xffffffffu;   // (threads beyond 2 BCH stage nothing)
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 r2 = make_uint2(0u, 0u);
    auto fetch = [&](uint32_t pos) {
        if (!upper) { r4 = b.recA[pos]; r2 = b.qmask[pos]; }
        else { r4 = b.recB[pos]; r2 = make_uint2(__float_as_uint(b.recC[pos].x), b.slot[pos]); }
    };
    // Everything the descriptor alone decides is asked for HERE, together: the pixel state AND the first round's records (round 5: the records were
    // asked for behind a barrier that waited for the pixel state -- two memory round trips in a row at the head of every one of 2 900 workgroups).
    const size_t pixc = inside ? pix_id : 0;                // (clamped, not predicated: the loads are in flight together)
    const float T_raw = s.final_T[pixc];
    const uint32_t last_contributor = inside ? s.n_contrib[pixc] : 0u;
    const float d_raw0 = dL_dpix[pixc], d_raw1 = dL_dpix[N + pixc], d_raw2 = dL_dpix[2 * N + pixc];
    if (ht < qmax) fetch(rg.x + qmax - 1 - ht);
    asm volatile("" ::: "memory");
    // (the selects behind the loads' issue: an asm statement waits for its operand where it stands)
    const float T_final = select_loaded(inside, T_raw);
    float dpx0 = select_loaded(inside, d_raw0), dpx1 = select_loaded(inside, d_raw1), dpx2 = select_loaded(inside, d_raw2);
    float T = T_final;
    float bg_dot_dpixel = 0.f;                              // backward.cu:533-535
    bg_dot_dpixel += bg[0] * dpx0; bg_dot_dpixel += bg[1] * dpx1; bg_dot_dpixel += bg[2] * dpx2;
    const float tfinal_bg = T_final * bg_dot_dpixel;
    float arA = 0.f;                                        // dL_dpixel . accum_rec with (last_alpha, last_color) already applied (bwd_chain4s)
    float vone = 1.0f, vzero = 0.0f;                        // identity elements, pinned to VGPRs for the DPP selects
    const QuadMasks qm = quad_masks();
    asm volatile("" : "+v"(vone), "+v"(vzero));
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);   // backward.cu:460-461

    uint32_t qlast[4];                                      // the deepest blended position per quadrant of this wave's block (wave-uniform)
    {
        uint32_t m = last_contributor;                      // max over the row's 4 pixels (lanes 4 apart), then one lane per row
        m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xf, 0xf, false));     // row_ror:4
        m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, false));     // row_ror:8
#pragma unroll
        for (int q = 0; q < 4; q++) qlast[q] = (uint32_t)__builtin_amdgcn_readlane((int)m, 16 * q);
    }
    // (no barrier here: the null record, written by thread 0 above, is first read behind the staging barrier below)

    // rows of the never-visited tail are zero
    for (uint32_t q = qmax + threadIdx.x; q < n; q += BWD_THREADS) {
        float4* row = b.slab + (size_t)b.slot[rg.x + q] * SLAB_ROW;
        row[0] = make_float4(0.f, 0.f, 0.f, 0.f); row[1] = make_float4(0.f, 0.f, 0.f, 0.f); row[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    auto stage = [&](int buf) {
        uint32_t h = ht;
        asm volatile("" : "+v"(h));                        // keeps the five LDS addresses from being hoisted into (spilled) VGPRs
        if (!upper) { stage_conic_a(r4); sA[h] = r4; sQ[h] = r2; }
        else { sFl[buf][h] = make_float2(r4.x, r4.y); stage_conic_b(r4); sB[h] = r4; acc.t[h].cb = __uint_as_float(r2.x); sSlot[buf][h] = r2.y; }
    };
    // Two barriers per round: [compute r] | flush r + zero its accumulator column + stage r+1 | [compute r+1] ...
    acc_clear_all(acc, threadIdx.x, BWD_THREADS);
    {
        const uint32_t cnt0 = min((uint32_t)BCH, qmax);
        if (ht < cnt0) stage(0);
    }
    __syncthreads();
    int rnd = 0;
    unsigned long long busy = 0ull;                        // (diagnostic builds only: tests/tools/timeline.py)
    // slot t of a round is list position qhi - 1 - t, blended by this pixel iff that is < last_contributor  <=>  16 t > 16 (qhi - 1 - last_contributor):
    // the threshold of the first round, lowered by a round's worth of slots behind every round (last_contributor itself is not needed again)
    int thr16 = ((int)qmax - 1 - (int)last_contributor) * 16;
    for (uint32_t qhi = qmax; qhi > 0; qhi = qhi > BCH ? qhi - BCH : 0, rnd ^= 1, thr16 -= BCH * 16) {
        const uint32_t cnt = min((uint32_t)BCH, qhi);
        // The records of the round AFTER this one are asked for here, behind the barrier and in front of this round's arithmetic.  Asked for
        // in front of the barrier (until round 4) they were waited for AT it: __syncthreads() is a workgroup fence, i.e. s_waitcnt vmcnt(0)
        // before s_barrier -- the prefetch bought nothing and every round paid a memory round trip with all sixteen waves idle.
        if (qhi > BCH && ht < qhi - BCH) fetch(rg.x + qhi - BCH - 1 - ht);
        const unsigned long long tb0 = busy_clock();
        {
            const uint32_t nl = build_own_list_q<BCH>(lists[wv], sQ, cnt, wv, lane);
            const unsigned short* myq = &qlists[wv][qd][e];
            const BwdPixel pxl = {pixfx, pixfy, dpx0, dpx1, dpx2, tfinal_bg, thr16, (uint32_t)pq * (uint32_t)(ACC_PLANE * 8)};
#pragma unroll 1
            for (uint32_t c0 = 0; c0 < nl; c0 += QCH) {
            const uint32_t nq = build_chunk_quadrant_lists<TGS_BWD_BOUNDED, 4>(qlists[wv], lists[wv], c0, nl, lane, BNULL, qhi - 1, qlast);
            bwd_passes(myq, nq, sA, sB, acc, 0u, pxl, pq == 0, T, arA, vone, vzero, qm);
            }
        }
        busy += busy_clock() - tb0;
        __syncthreads();                                    // every wave is done with the records and the accumulator of this round
        if (threadIdx.x < cnt) {                            // flush: one 48-B row per instance, then clear the column for the next round
            const uint32_t j = threadIdx.x;
            // moments -> gradients (backward.cu:537-555): dL_dG = opacity * dL_dalpha, dG/ddel = -G (conic . d), conic terms * -0.5
            // (sA[j] is restaged by this same thread below; sB[j] by thread BCH + j, possibly already: hence sFl)
            const float4 a = sA[j]; const float2 fl = sFl[rnd][j];
            const float cxx = a.z * UNSCALE_CONIC, cxy = a.w * UNSCALE_CONIC_XY, cyy = fl.x, op = fl.y;
            const float Sx = (float)acc_get(acc, 3, j), Sy = (float)acc_get(acc, 4, j);
            float4* row = b.slab + (size_t)sSlot[rnd][j] * SLAB_ROW;
            row[0] = make_float4((float)acc_get(acc, 0, j), (float)acc_get(acc, 1, j), (float)acc_get(acc, 2, j), op * (-Sx * cxx - Sy * cxy) * ddelx_dx);
            const ConicHiLo c5 = conic_hilo(op, acc_get(acc, 5, j)), c6 = conic_hilo(op, acc_get(acc, 6, j)), c7 = conic_hilo(op, acc_get(acc, 7, j));
            row[1] = make_float4(op * (-Sy * cyy - Sx * cxy) * ddely_dy, c5.hi, c6.hi, c7.hi);
            row[2] = make_float4((float)acc_get(acc, 8, j), c5.lo, c6.lo, c7.lo);
            acc_clear(acc, j);
        }
        if (qhi > BCH) {                                    // stage the next round (its records were prefetched into registers)
            const uint32_t qn = qhi - BCH, cntn = min((uint32_t)BCH, qn);
            if (ht < cntn) stage(rnd ^ 1);
            __syncthreads();
        }
    }
    busy_report(s, tile, 1, busy);
    stamp(s, tile, 3);
}


// ---------------------------------------------------------------------------------------------
// Per-Gaussian half of the backward for ONE view (computeCov2DCUDA backward.cu:144-274, preprocessCUDA :346-396,
// computeCov3D :278-341, computeColorFromSH :20-139), shared by k_preprocess_bwd (one view per launch) and
// k_preprocess_bwd_batch (all views of a batch per launch).
// ---------------------------------------------------------------------------------------------
// a[0..8] <- sum of this Gaussian's tile partials (its slab rows are contiguous).  A splat on few tiles is summed by its own
// lane in tile order; one on >= SLAB_COOP tiles -- a per-lane loop of thousands of dependent iterations otherwise, the
// tail of the whole kernel on scenes with oversized splats -- is summed by the whole wave: lane l takes rows l, l + 64, ...
// and the nine totals are formed with wave_sum.  Fixed order either way.  Call from convergent code.
constexpr uint32_t SLAB_COOP = 128;
// Round 5: the threshold is chosen per WAVE.  A wave walks its lanes' own loops as long as ANY lane has rows left (four rows = one memory round
// trip per trip), so a few large splats among small ones set the trip count for all 64 -- and a cooperative pass is one round trip per large
// splat.  In round trips: threshold t costs (lanes with >= t rows) + t / 4; the cheapest of 16 / 32 / 64 / 128 is taken (all lanes at 20 rows:
// 32 -- five trips, nobody cooperates; three lanes at 50 rows among small ones: 16 -- three passes + at most four trips).  Measured with a fixed
// threshold (tools/stage_times.py, splats x4 / x8): 128 -> 88.6 / 122.5 us, 16 -> 74.3 / 103.5 us for the one-view kernel; x1 unchanged.
// The same lanes meet in a wave of every kernel that sums a view's rows (64 consecutive Gaussians), so all of them decide alike.
__device__ __forceinline__ uint32_t slab_coop_threshold(uint32_t tiles)
{
    const uint32_t n16 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(tiles >= 16u)), n32 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(tiles >= 32u));
    const uint32_t n64 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(tiles >= 64u)), n128 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(tiles >= SLAB_COOP));
    uint32_t thr = SLAB_COOP, best = n128 + SLAB_COOP / 4u;
    if (n64 + 16u < best) { best = n64 + 16u; thr = 64u; }
    if (n32 + 8u < best) { best = n32 + 8u; thr = 32u; }
    if (n16 + 4u < best) { best = n16 + 4u; thr = 16u; }
    return thr;
}
// (tiles, off): the Gaussian's tiles_touched and offsets of this view, fetched by the caller
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// cn[0..2]: the conic sums (xx, xy, yy) in double from the rows' hi + lo parts (conic_hilo); a[5..7] are their fp32 roundings
__device__ __forceinline__ void slab_sum(bool live, uint32_t tiles_in, uint32_t off_in, const BinState& b, float (&a)[NACC], double (&cn)[3])
{
    const int lane = threadIdx.x & 63;
    const uint32_t tiles = live ? tiles_in : 0u, off = live ? off_in : 0u;
#pragma unroll
    for (int c = 0; c < NACC; c++) a[c] = 0.f;
    cn[0] = cn[1] = cn[2] = 0.0;
    const uint32_t coop = slab_coop_threshold(tiles);
    unsigned long long big = __builtin_amdgcn_ballot_w64(tiles >= coop);
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)tiles, src), o = (uint32_t)__builtin_amdgcn_readlane((int)off, src);
        float p[NACC];
        double pc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < NACC; c++) p[c] = 0.f;
        for (uint32_t k = lane; k < n; k += 64) {
            const float4* row = b.slab + (size_t)(o + k) * SLAB_ROW;
            const float4 r0 = row[0], r1 = row[1], r2 = row[2];
            p[0] += r0.x; p[1] += r0.y; p[2] += r0.z; p[3] += r0.w; p[4] += r1.x; p[8] += r2.x;
            pc[0] += (double)r1.y + (double)r2.y; pc[1] += (double)r1.z + (double)r2.z; pc[2] += (double)r1.w + (double)r2.w;
        }
#pragma unroll
        for (int c = 0; c < NACC; c++) { if (c >= 5 && c <= 7) continue; const float tot = wave_sum(p[c]); if (lane == src) a[c] = tot; }
#pragma unroll
        for (int c = 0; c < 3; c++) { const double tot = wave_sum_f64(pc[c]); if (lane == src) cn[c] = tot; }
    }
    if (tiles < coop) {
        // four rows per trip, their loads issued together (round 5: row by row a splat's rows were one memory round trip EACH in the batch
        // kernels' view loop -- 2.5 rows per visible splat and view at config 3: 197 -> 181 us per 8 views --, two per trip in the one-view
        // kernel: 60.5 -> 59.5 us); same order of the additions; rows past the last are the last one again (a valid address: no load under a branch of its own) and are not added
        const tgs_v4f* row = reinterpret_cast<const tgs_v4f*>(b.slab) + (size_t)off * SLAB_ROW;
        for (uint32_t k = 0; k < tiles; k += 4) {
            const uint32_t last = tiles - 1u;
            const tgs_v4f* pa = row + (size_t)k * SLAB_ROW;
            const tgs_v4f* pb = row + (size_t)min(k + 1u, last) * SLAB_ROW;
            const tgs_v4f* pc = row + (size_t)min(k + 2u, last) * SLAB_ROW;
            const tgs_v4f* pd = row + (size_t)min(k + 3u, last) * SLAB_ROW;
            const tgs_v4f r0 = pa[0], r1 = pa[1], r2 = pa[2], t0 = pb[0], t1 = pb[1], t2 = pb[2], u0 = pc[0], u1 = pc[1], u2 = pc[2], w0 = pd[0], w1 = pd[1], w2 = pd[2];
            asm volatile("" ::: "memory");
            a[0] += r0.x; a[1] += r0.y; a[2] += r0.z; a[3] += r0.w; a[4] += r1.x; a[8] += r2.x;
            cn[0] += (double)r1.y + (double)r2.y; cn[1] += (double)r1.z + (double)r2.z; cn[2] += (double)r1.w + (double)r2.w;
            if (k + 1u < tiles) {
                a[0] += t0.x; a[1] += t0.y; a[2] += t0.z; a[3] += t0.w; a[4] += t1.x; a[8] += t2.x;
                cn[0] += (double)t1.y + (double)t2.y; cn[1] += (double)t1.z + (double)t2.z; cn[2] += (double)t1.w + (double)t2.w;
            }
            if (k + 2u < tiles) {
                a[0] += u0.x; a[1] += u0.y; a[2] += u0.z; a[3] += u0.w; a[4] += u1.x; a[8] += u2.x;
                cn[0] += (double)u1.y + (double)u2.y; cn[1] += (double)u1.z + (double)u2.z; cn[2] += (double)u1.w + (double)u2.w;
            }
            if (k + 3u < tiles) {
                a[0] += w0.x; a[1] += w0.y; a[2] += w0.z; a[3] += w0.w; a[4] += w1.x; a[8] += w2.x;
                cn[0] += (double)w1.y + (double)w2.y; cn[1] += (double)w1.z + (double)w2.z; cn[2] += (double)w1.w + (double)w2.w;
            }
        }
    }
    a[5] = (float)cn[0]; a[6] = (float)cn[1]; a[7] = (float)cn[2];
}

// computeColorFromSH backward (backward.cu:20-139) of one Gaussian in one view: from the summed colour gradient rgb[0..2] and the clamp
// bits, dRGB (the gradient that reaches the SH row), coef[k] (dL_dsh[k][c] = coef[k] * dRGB[c]) and the view-direction path's share of
// dL_dmean3D, ADDED to dmean.  sh(i): coefficient i of the Gaussian's [16][3] row.
template <typename ShRow>
__device__ __forceinline__ void sh_backward_terms(int D, const ShRow& sh, uint32_t cl, float rgb0, float rgb1, float rgb2, float mx, float my, float mz,
                                          float camx, float camy, float camz, float (&coef)[16], float (&dRGB)[3], float (&dmean)[3])
{
    dRGB[0] = rgb0 * ((cl & 1u) ? 0.f : 1.f);
    dRGB[1] = rgb1 * ((cl & 2u) ? 0.f : 1.f);
    dRGB[2] = rgb2 * ((cl & 4u) ? 0.f : 1.f);
    const float ox = mx - camx, oy = my - camy, oz = mz - camz;
    const float len = sqrtf(ox * ox + oy * oy + oz * oz);
    const float x = ox / len, y = oy / len, z = oz / len;
    float gxv[3] = {0.f, 0.f, 0.f}, gyv[3] = {0.f, 0.f, 0.f}, gzv[3] = {0.f, 0.f, 0.f};
#define SH(k) sh(3 * (k) + c)
    coef[0] = SH_C0;
    if (D > 0) {
        coef[1] = -SH_C1 * y; coef[2] = SH_C1 * z; coef[3] = -SH_C1 * x;
#pragma unroll
        for (int c = 0; c < 3; c++) { gxv[c] = -SH_C1 * SH(3); gyv[c] = -SH_C1 * SH(1); gzv[c] = SH_C1 * SH(2); }
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            coef[4] = SH_C2_0 * xy; coef[5] = SH_C2_1 * yz; coef[6] = SH_C2_2 * (2.f * zz - xx - yy); coef[7] = SH_C2_3 * xz; coef[8] = SH_C2_4 * (xx - yy);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                gxv[c] += SH_C2_0 * y * SH(4) + SH_C2_2 * 2.f * -x * SH(6) + SH_C2_3 * z * SH(7) + SH_C2_4 * 2.f * x * SH(8);
                gyv[c] += SH_C2_0 * x * SH(4) + SH_C2_1 * z * SH(5) + SH_C2_2 * 2.f * -y * SH(6) + SH_C2_4 * 2.f * -y * SH(8);
                gzv[c] += SH_C2_1 * y * SH(5) + SH_C2_2 * 2.f * 2.f * z * SH(6) + SH_C2_3 * x * SH(7);
            }
            if (D > 2) {
                coef[9] = SH_C3_0 * y * (3.f * xx - yy); coef[10] = SH_C3_1 * xy * z; coef[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
                coef[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy); coef[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
                coef[14] = SH_C3_5 * z * (xx - yy); coef[15] = SH_C3_6 * x * (xx - 3.f * yy);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    gxv[c] += (SH_C3_0 * SH(9) * 3.f * 2.f * xy + SH_C3_1 * SH(10) * yz + SH_C3_2 * SH(11) * -2.f * xy +
                               SH_C3_3 * SH(12) * -3.f * 2.f * xz + SH_C3_4 * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                               SH_C3_5 * SH(14) * 2.f * xz + SH_C3_6 * SH(15) * 3.f * (xx - yy));
                    gyv[c] += (SH_C3_0 * SH(9) * 3.f * (xx - yy) + SH_C3_1 * SH(10) * xz + SH_C3_2 * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                               SH_C3_3 * SH(12) * -3.f * 2.f * yz + SH_C3_4 * SH(13) * -2.f * xy + SH_C3_5 * SH(14) * -2.f * yz +
                               SH_C3_6 * SH(15) * -3.f * 2.f * xy);
                    gzv[c] += (SH_C3_1 * SH(10) * xy + SH_C3_2 * SH(11) * 4.f * 2.f * yz + SH_C3_3 * SH(12) * 3.f * (2.f * zz - xx - yy) +
                               SH_C3_4 * SH(13) * 4.f * 2.f * xz + SH_C3_5 * SH(14) * (xx - yy));
                }
            }
        }
    }
#undef SH
    const float ddx = gxv[0] * dRGB[0] + gxv[1] * dRGB[1] + gxv[2] * dRGB[2];
    const float ddy = gyv[0] * dRGB[0] + gyv[1] * dRGB[1] + gyv[2] * dRGB[2];
    const float ddz = gzv[0] * dRGB[0] + gzv[1] * dRGB[1] + gzv[2] * dRGB[2];
    // dnormvdv (auxiliary.h:107-117)
    const float sum2 = ox * ox + oy * oy + oz * oz;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean[0] += ((+sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
    dmean[1] += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
    dmean[2] += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
}

// ---------------------------------------------------------------------------------------------
// The per-Gaussian chain dL_dconic, dL_dmean2D -> dL_dcov3D, dL_dmean3D, dL_dscale, dL_drot (computeCov2DCUDA backward.cu:144-274, the
// projection part of preprocessCUDA :369-387, computeCov3D :278-341) in DOUBLE (round 4).
// These are the formulas whose fp32 evaluation cancels -- (denom - a c), T T^T differences, the quaternion derivative of a needle -- and
// the three-way adjudication of round 4 (tests/adjudicate.py: HIP / fp32 oracle / the oracle's text in double) showed the fp32 version of
// this chain further from exact arithmetic than the reference's on 3 of 896 fuzz scenes (one needle-shaped splat each, e.g. dL_dcov3D
// 2.3e-4 against the reference's 2.8e-5).  gfx950 issues f64 FMA at half the f32 rate and the kernels around this are HBM-bound: ~400
// double operations per Gaussian and view are invisible in time (measured: DESIGN.md section 3), and the result is the reference's
// FUNCTION evaluated on the fp32 inputs to < 1e-12, rounded once.
// ---------------------------------------------------------------------------------------------

// (1) computeCov2DCUDA + the projection part: per view.  dc: dL_dcov3D of this view in double (reference layout: off-diagonals doubled).
// ACCUMULATE: dc += this view's share (the batch kernels' running sum over the views) instead of dc = it.
template <bool ACCUMULATE>
__device__ __forceinline__ void cov2d_chain_bwd_f64(float mxf, float myf, float mzf, const float (&cov3d)[6], const CamParams& cam, const ViewMat& V, const ViewMat& PM,
                                                    const double (&dLconic)[3], float g2xf, float g2yf, float (&dmean)[3], double (&dc)[6])
{
    typedef double R;
    const R mx = mxf, my = myf, mz = mzf;
    const float* vm = V.m;
    R tx = (R)vm[0] * mx + (R)vm[4] * my + (R)vm[8] * mz + (R)vm[12];
    R ty = (R)vm[1] * mx + (R)vm[5] * my + (R)vm[9] * mz + (R)vm[13];
    const R tz = (R)vm[2] * mx + (R)vm[6] * my + (R)vm[10] * mz + (R)vm[14];
    const R limx = (R)(1.3f * cam.tan_fovx), limy = (R)(1.3f * cam.tan_fovy);
    // 1 / tz once (v_rcp_f64 + two Newton steps: full double precision for the finite tz > 0.2 of a splat that was not culled), not three IEEE
    // divisions of ~10 double instructions each: this chain is the long pole of the batch pass's geometry half (354 double instructions per view)
    R iz = __builtin_amdgcn_rcp(tz);
    iz = fma(fma(-tz, iz, 1.0), iz, iz);
    iz = fma(fma(-tz, iz, 1.0), iz, iz);
    const R txtz = tx * iz, tytz = ty * iz;
    tx = fmin(limx, fmax(-limx, txtz)) * tz;
    ty = fmin(limy, fmax(-limy, tytz)) * tz;
    const R xg = (txtz < -limx || txtz > limx) ? 0.0 : 1.0, yg = (tytz < -limy || tytz > limy) ? 0.0 : 1.0;
    const R fx = cam.focal_x, fy = cam.focal_y;
    const R iz2 = iz * iz, iz3 = iz2 * iz;
    // J (rows): [fx/tz, 0, -fx tx/tz^2], [0, fy/tz, -fy ty/tz^2];  T = W J in GLM terms: T.m[c][r] of the fp32 code.  Written out:
    // t0[k] = T.m[0][k], t1[k] = T.m[1][k]  (k = 0..2): the two rows of J applied to the columns of the view rotation
    const R j00 = fx * iz, j02 = -(fx * tx) * iz2, j11 = fy * iz, j12 = -(fy * ty) * iz2;
#define w_(c_, r_) ((R)vm[(c_) + 4 * (r_)])                // W.m[c][r] of compute_cov2d = vm[c + 4 r]; converted at each use (uniform: one v_cvt)
    R t0[3], t1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { t0[k] = w_(0, k) * j00 + w_(2, k) * j02; t1[k] = w_(1, k) * j11 + w_(2, k) * j12; }
    const R v[3][3] = {{cov3d[0], cov3d[1], cov3d[2]}, {cov3d[1], cov3d[3], cov3d[4]}, {cov3d[2], cov3d[4], cov3d[5]}};
    R r0[3], r1[3];                                        // V t0, V t1
#pragma unroll
    for (int k = 0; k < 3; k++) { r0[k] = v[k][0] * t0[0] + v[k][1] * t0[1] + v[k][2] * t0[2]; r1[k] = v[k][0] * t1[0] + v[k][1] * t1[1] + v[k][2] * t1[2]; }
    const R ca = t0[0] * r0[0] + t0[1] * r0[1] + t0[2] * r0[2] + (R)0.3f;
    const R cb = t0[0] * r1[0] + t0[1] * r1[1] + t0[2] * r1[2];
    const R cc = t1[0] * r1[0] + t1[1] * r1[1] + t1[2] * r1[2] + (R)0.3f;
    const R g0 = dLconic[0], g1 = dLconic[1], g2 = dLconic[2];
    const R denom = ca * cc - cb * cb;
    const R d2i = 1.0 / (denom * denom + (R)0.0000001f);
    R dLa = 0, dLb = 0, dLc = 0;
    if (!ACCUMULATE) {
#pragma unroll
        for (int k = 0; k < 6; k++) dc[k] = 0;
    }
    if (d2i != 0) {
        dLa = d2i * (-cc * cc * g0 + 2 * cb * cc * g1 + (denom - ca * cc) * g2);
        dLc = d2i * (-ca * ca * g2 + 2 * ca * cb * g1 + (denom - ca * cc) * g0);
        dLb = d2i * 2 * (cb * cc * g0 - (denom + 2 * cb * cb) * g1 + ca * cb * g2);
        // dL_dcov3D = T^T dSig T with dSig = [[dLa, dLb / 2], [dLb / 2, dLc]] (backward.cu:215-229 written out), formed through the two rows
        // P = dSig T -- 27 double operations instead of 48; entry kl is t0[k] P0[l] + t1[k] P1[l], twice that off the diagonal
        const R hb = 0.5 * dLb;
        R P0[3], P1[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { P0[k] = dLa * t0[k] + hb * t1[k]; P1[k] = hb * t0[k] + dLc * t1[k]; }
        dc[0] += t0[0] * P0[0] + t1[0] * P1[0];
        dc[3] += t0[1] * P0[1] + t1[1] * P1[1];
        dc[5] += t0[2] * P0[2] + t1[2] * P1[2];
        dc[1] += 2 * (t0[0] * P0[1] + t1[0] * P1[1]);
        dc[2] += 2 * (t0[0] * P0[2] + t1[0] * P1[2]);
        dc[4] += 2 * (t0[1] * P0[2] + t1[1] * P1[2]);
    }
    // dL_dT (backward.cu:231-242): dT0[k] = 2 (V t0)[k] dLa + (V t1)[k] dLb,  dT1[k] = 2 (V t1)[k] dLc + (V t0)[k] dLb
    R dT0[3], dT1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { dT0[k] = 2 * r0[k] * dLa + r1[k] * dLb; dT1[k] = 2 * r1[k] * dLc + r0[k] * dLb; }
    const R dJ00 = w_(0, 0) * dT0[0] + w_(0, 1) * dT0[1] + w_(0, 2) * dT0[2];
    const R dJ02 = w_(2, 0) * dT0[0] + w_(2, 1) * dT0[1] + w_(2, 2) * dT0[2];
    const R dJ11 = w_(1, 0) * dT1[0] + w_(1, 1) * dT1[1] + w_(1, 2) * dT1[2];
    const R dJ12 = w_(2, 0) * dT1[0] + w_(2, 1) * dT1[1] + w_(2, 2) * dT1[2];
    const R dtx = xg * -fx * iz2 * dJ02, dty = yg * -fy * iz2 * dJ12;
    const R dtz = -fx * iz2 * dJ00 - fy * iz2 * dJ11 + (2 * fx * tx) * iz3 * dJ02 + (2 * fy * ty) * iz3 * dJ12;
    // the view-space part rounded once; the projection part (backward.cu:369-387) is well-conditioned: fp32, in the reference's operation order
    const float dm0 = (float)(vm[0] * dtx + vm[1] * dty + vm[2] * dtz), dm1 = (float)(vm[4] * dtx + vm[5] * dty + vm[6] * dtz), dm2 = (float)(vm[8] * dtx + vm[9] * dty + vm[10] * dtz);
    const float* pj = PM.m;
    const float m_w = 1.0f / ((pj[3] * mxf + pj[7] * myf + pj[11] * mzf + pj[15]) + 0.0000001f);
    const float mul1 = (pj[0] * mxf + pj[4] * myf + pj[8] * mzf + pj[12]) * m_w * m_w, mul2 = (pj[1] * mxf + pj[5] * myf + pj[9] * mzf + pj[13]) * m_w * m_w;
    dmean[0] = dm0 + ((pj[0] * m_w - pj[3] * mul1) * g2xf + (pj[1] * m_w - pj[3] * mul2) * g2yf);
    dmean[1] = dm1 + ((pj[4] * m_w - pj[7] * mul1) * g2xf + (pj[5] * m_w - pj[7] * mul2) * g2yf);
    dmean[2] = dm2 + ((pj[8] * m_w - pj[11] * mul1) * g2xf + (pj[9] * m_w - pj[11] * mul2) * g2yf);
#undef w_
}

// (2) computeCov3D backward (backward.cu:278-341): M = S R, dM = (2 M) dSig, dMt = dM^T in GLM's column-major products.  LINEAR in dc and
// independent of the view: the batch kernels run it ONCE on the sum of the views' dc (the reference runs it per view and lets autograd add
// the results: the same sum up to fp32 rounding of the addends).
__device__ __forceinline__ void cov3d_bwd_f64(const double (&dc)[6], float scale_modifier, const float* __restrict__ scales3, const float* __restrict__ rot4,
                                              float (&dscale)[3], float (&drot)[4])
{
    typedef double R;
    {
        const R q0 = rot4[0], qx = rot4[1], qy = rot4[2], qz = rot4[3];
        const R Rg[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - q0 * qz), 2 * (qx * qz + q0 * qy)},
                            {2 * (qx * qy + q0 * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - q0 * qx)},
                            {2 * (qx * qz - q0 * qy), 2 * (qy * qz + q0 * qx), 1 - 2 * (qx * qx + qy * qy)}};      // Rg[c][w]: GLM column c, row w
        const R sc[3] = {(R)scale_modifier * (R)scales3[0], (R)scale_modifier * (R)scales3[1], (R)scale_modifier * (R)scales3[2]};
        const R dS[3][3] = {{dc[0], 0.5 * dc[1], 0.5 * dc[2]}, {0.5 * dc[1], dc[3], 0.5 * dc[4]}, {0.5 * dc[2], 0.5 * dc[4], dc[5]}};
        // GLM product (a * b).m[c][w] = sum_k a.m[k][w] b.m[c][k]; S is diagonal, so Mx = S * R has Mx.m[c][w] = s_w Rg[c][w]
        // dMt.m[c][w] = dM.m[w][c] = sum_k 2 s_c Rg[k][c] dS[w][k], formed column by column; dscale[c] needs column c before, the
        // quaternion derivative (linear in the entries A(c, w) = s_c dMt.m[c][w]) is accumulated as the entries appear: 3 live doubles, not 9
        R dq[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            R col[3];
#pragma unroll
            for (int wi = 0; wi < 3; wi++) col[wi] = 2 * sc[c] * (Rg[0][c] * dS[wi][0] + Rg[1][c] * dS[wi][1] + Rg[2][c] * dS[wi][2]);
            dscale[c] = (float)(Rg[0][c] * col[0] + Rg[1][c] * col[1] + Rg[2][c] * col[2]);
            const R A0 = col[0] * sc[c], A1 = col[1] * sc[c], A2 = col[2] * sc[c];         // A(c, 0..2)
            if (c == 0) {        // A(0,0) A(0,1) A(0,2)
                dq[0] += 2 * qz * A1 - 2 * qy * A2;
                dq[1] += 2 * qy * A1 + 2 * qz * A2;
                dq[2] += 2 * qx * A1 - 2 * q0 * A2 - 4 * qy * A0;
                dq[3] += 2 * q0 * A1 + 2 * qx * A2 - 4 * qz * A0;
            } else if (c == 1) { // A(1,0) A(1,1) A(1,2)
                dq[0] += -2 * qz * A0 + 2 * qx * A2;
                dq[1] += 2 * qy * A0 + 2 * q0 * A2 - 4 * qx * A1;
                dq[2] += 2 * qx * A0 + 2 * qz * A2;
                dq[3] += -2 * q0 * A0 + 2 * qy * A2 - 4 * qz * A1;
            } else {             // A(2,0) A(2,1) A(2,2)
                dq[0] += 2 * qy * A0 - 2 * qx * A1;
                dq[1] += 2 * qz * A0 - 2 * q0 * A1 - 4 * qx * A2;
                dq[2] += 2 * q0 * A0 + 2 * qz * A1 - 4 * qy * A2;
                dq[3] += 2 * qx * A0 + 2 * qy * A1;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) drot[k] = (float)dq[k];
    }
}

// the Gaussian's 3D covariance as the forward used it: evaluated again from scale and rotation (compute_cov3d, bit-identical), or the caller's
template <bool HAS_SCALE_ROT>
__device__ __forceinline__ void load_cov3d(const BwdIn& in, float scale_modifier, int idx, float (&cov3d)[6])
{
    if (HAS_SCALE_ROT) {
        const size_t i3 = 3 * (size_t)idx;
        compute_cov3d(scale_modifier, in.scales[i3], in.scales[i3 + 1], in.scales[i3 + 2], reinterpret_cast<const float4*>(in.rotations)[idx], cov3d);
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) cov3d[i] = in.cov3D_precomp[6 * (size_t)idx + i];
    }
}

struct GaussTerms {
    float a[NACC];                 // sums of the tile partials: colour rgb, mean2D xy, conic xx xy yy, opacity
    float dmean[3], dcov[6], dscale[3], drot[4];

    float coef[16], dRGB[3];       // dL_dsh[k][c] = coef[k] * dRGB[c]
};

// One view's terms of a Gaussian in the batch kernels.  dL_dcov3D of the view is ADDED to dcacc (double; the caller's running sum over the
// views) and the scale / rotation gradients are left to the caller (finish_cov3d, once behind the view loop): t.dcov / t.dscale / t.drot stay zero
template <bool HAS_SH, typename ShRow>
__device__ __forceinline__ void pergauss_terms(int idx, bool live, int D, const ShRow& sh, const CamParams& cam, const ViewMat& V, const ViewMat& PM,
                                               float camx, float camy, float camz, const GeomState& g, const BinState& b, uint32_t tiles_in, uint32_t off_in,
                                               float mx, float my, float mz, const float (&cov3d)[6], GaussTerms& t, double (&dcacc)[6])
{
    float (&a)[NACC] = t.a;
    float (&dmean)[3] = t.dmean; float (&dcov)[6] = t.dcov; float (&dscale)[3] = t.dscale; float (&drot)[4] = t.drot;
    float (&coef)[16] = t.coef; float (&dRGB)[3] = t.dRGB;
    double cn[3];
    slab_sum(live, tiles_in, off_in, b, a, cn);     // convergent: the wave helps its splats that touch many tiles
#pragma unroll
    for (int k = 0; k < 3; k++) { dmean[k] = 0.f; dscale[k] = 0.f; dRGB[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 6; k++) dcov[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) drot[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) coef[k] = 0.f;
    if (!live) return;
    cov2d_chain_bwd_f64<true>(mx, my, mz, cov3d, cam, V, PM, cn, a[3], a[4], dmean, dcacc);
    if (HAS_SH) {
            asm volatile("" ::: "memory");            // keep the 48 SH reads below from being hoisted over the covariance math (VGPR pressure)
            sh_backward_terms(D, sh, g.clamped[idx], a[0], a[1], a[2], mx, my, mz, camx, camy, camz, coef, dRGB, dmean);
    }
}

// behind the view loop of a batch kernel: dL_dcov3D = the double sum rounded once, and the scale / rotation gradients from that sum
template <bool HAS_SCALE_ROT>
__device__ __forceinline__ void finish_cov3d(const BwdIn& in, float scale_modifier, int idx, bool in_range, const double (&dcsum)[6], float (&dcov)[6], float (&dscale)[3],
                                             float (&drot)[4])
{
#pragma unroll
    for (int k = 0; k < 6; k++) dcov[k] = (float)dcsum[k];
    if (HAS_SCALE_ROT && in_range) cov3d_bwd_f64(dcsum, scale_modifier, in.scales + 3 * (size_t)idx, in.rotations + 4 * (size_t)idx, dscale, drot);
}

// ---------------------------------------------------------------------------------------------
// k_preprocess_bwd: one view per launch
// ---------------------------------------------------------------------------------------------
// the parameter gradients of one Gaussian: stored, or added to what the buffers hold (accumulate)
__device__ __forceinline__ void store_param_grads(const BwdIn& in, int idx, float dopacity, const float (&dcolor)[3], const float (&dmean)[3], const float (&dcov)[6],
                                                  const float (&dscale)[3], const float (&drot)[4])
{
    const size_t i3 = 3 * (size_t)idx;
    if (in.accumulate) {
        // fused gradient accumulation of a multi-view batch (youreditableavatar_amd/multiview.py): += on the parameter gradients
        in.dL_dopacity[idx] += dopacity;
        if (in.dL_dcolor) { in.dL_dcolor[i3] += dcolor[0]; in.dL_dcolor[i3 + 1] += dcolor[1]; in.dL_dcolor[i3 + 2] += dcolor[2]; }   // NULL: intermediate on the SH path
        in.dL_dmean3D[i3] += dmean[0]; in.dL_dmean3D[i3 + 1] += dmean[1]; in.dL_dmean3D[i3 + 2] += dmean[2];
        if (in.dL_dcov3D) {                                                                                         // NULL: intermediate on the scale/rot path
#pragma unroll
            for (int i = 0; i < 6; i++) in.dL_dcov3D[6 * (size_t)idx + i] += dcov[i];
        }
        if (in.dL_dscale) { in.dL_dscale[i3] += dscale[0]; in.dL_dscale[i3 + 1] += dscale[1]; in.dL_dscale[i3 + 2] += dscale[2]; }
        if (in.dL_drot) { float4 p = reinterpret_cast<float4*>(in.dL_drot)[idx]; p.x += drot[0]; p.y += drot[1]; p.z += drot[2]; p.w += drot[3]; reinterpret_cast<float4*>(in.dL_drot)[idx] = p; }
        return;
    }
    in.dL_dopacity[idx] = dopacity;
    if (in.dL_dcolor) { in.dL_dcolor[i3] = dcolor[0]; in.dL_dcolor[i3 + 1] = dcolor[1]; in.dL_dcolor[i3 + 2] = dcolor[2]; }
    in.dL_dmean3D[i3] = dmean[0]; in.dL_dmean3D[i3 + 1] = dmean[1]; in.dL_dmean3D[i3 + 2] = dmean[2];
    if (in.dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) in.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    }
    if (in.dL_dscale) { in.dL_dscale[i3] = dscale[0]; in.dL_dscale[i3 + 1] = dscale[1]; in.dL_dscale[i3 + 2] = dscale[2]; }
    if (in.dL_drot) reinterpret_cast<float4*>(in.dL_drot)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
}

// dL_dsh rows of the workgroup, 48 values per thread in `o48` (M == 16): through LDS so that global memory sees 16 B per lane
template <typename Row48>
__device__ __forceinline__ void store_sh_rows_staged(const BwdIn& in, float4* sh_lds, const Row48& o48, uint32_t blk)
{
    const size_t base4 = (size_t)blk * PRE_BLOCK * 12, total4 = (size_t)in.P * 12;
    __syncthreads();                                       // every thread has consumed its SH row
#pragma unroll
    for (int q = 0; q < 12; q++) sh_lds[threadIdx.x * 12 + q] = make_float4(o48(4 * q), o48(4 * q + 1), o48(4 * q + 2), o48(4 * q + 3));
    __syncthreads();
    float4* d4 = reinterpret_cast<float4*>(in.dL_dsh);
    float4 prev[12];
    if (in.accumulate) {                                   // (uniform) the twelve reads in flight together
#pragma unroll
        for (int q = 0; q < 12; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; prev[q] = d4[i < total4 ? i : total4 - 1]; }
    } else {
#pragma unroll
        for (int q = 0; q < 12; q++) prev[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 12; q++) {
        const size_t i = base4 + q * PRE_BLOCK + threadIdx.x;
        if (i < total4) {
            float4 o = sh_lds[q * PRE_BLOCK + threadIdx.x];
            if (in.accumulate) { o.x += prev[q].x; o.y += prev[q].y; o.z += prev[q].z; o.w += prev[q].w; }
            d4[i] = o;
        }
    }
}
__device__ __forceinline__ void load_sh_rows_staged(const BwdIn& in, float4* sh_lds, uint32_t blk)
{
    const size_t base4 = (size_t)blk * PRE_BLOCK * 12, total4 = (size_t)in.P * 12;
    const float4* s4 = reinterpret_cast<const float4*>(in.shs);
    // into registers first, then into LDS: written as `if (i < total4) sh_lds[..] = s4[i]` every load is waited for on its own before its
    // LDS write -- twelve memory latencies in a row at the head of the workgroup
    float4 r[12];
#pragma unroll
    for (int q = 0; q < 12; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; r[q] = s4[i < total4 ? i : total4 - 1]; }
#pragma unroll
    for (int q = 0; q < 12; q++) sh_lds[q * PRE_BLOCK + threadIdx.x] = r[q];
    __syncthreads();
}

// (The one-view kernel keeps its own copy of the math of pergauss_terms: routed through the shared function hipcc
// allocates 172 VGPRs instead of 132 -- 2 resident waves per SIMD instead of 3 -- and the kernel takes 120 us instead
// of 98.  tests/test_gpu_api.py::test_batched_backward_equals_per_view_backward keeps the two in step.)
template <bool HAS_SH, bool HAS_SCALE_ROT>
__global__ __launch_bounds__(PRE_BLOCK) void k_preprocess_bwd(const BwdIn in, const CamParams cam, const GeomState g, const BinState b)
{
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    if (in.meta && (in.meta->error & META_ERR_CAPACITY)) {              // rejected frame (tgs_forward_async): contributes nothing
        if (idx < in.P) {
            const size_t j3 = 3 * (size_t)idx;
            in.dL_dmean2D[j3] = 0.f; in.dL_dmean2D[j3 + 1] = 0.f; in.dL_dmean2D[j3 + 2] = 0.f;
            if (!in.accumulate) {                                       // tgs_backward promises that every output element is written: zeros
                if (in.dL_dconic) reinterpret_cast<float4*>(in.dL_dconic)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
                in.dL_dopacity[idx] = 0.f;
                if (in.dL_dcolor) { in.dL_dcolor[j3] = 0.f; in.dL_dcolor[j3 + 1] = 0.f; in.dL_dcolor[j3 + 2] = 0.f; }
                in.dL_dmean3D[j3] = 0.f; in.dL_dmean3D[j3 + 1] = 0.f; in.dL_dmean3D[j3 + 2] = 0.f;
                if (in.dL_dcov3D) for (int i = 0; i < 6; i++) in.dL_dcov3D[6 * (size_t)idx + i] = 0.f;
                if (in.dL_dscale) { in.dL_dscale[j3] = 0.f; in.dL_dscale[j3 + 1] = 0.f; in.dL_dscale[j3 + 2] = 0.f; }
                if (in.dL_drot) reinterpret_cast<float4*>(in.dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (HAS_SH && in.dL_dsh) for (int i = 0; i < 3 * in.M; i++) in.dL_dsh[(size_t)idx * in.M * 3 + i] = 0.f;
            }
        }
        return;
    }
    const ViewMat V = load_mat(cam.view), PM = load_mat(cam.proj);
    const float camx = cam.campos[0], camy = cam.campos[1], camz = cam.campos[2];
    // SH rows in, dL_dsh rows out: staged through LDS so that global memory sees 16 B per lane, fully coalesced
    // (a thread's 12 cells are 48 banks apart: every fourth lane of a 16-B access meets the same banks -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.61.
    // Rows padded to 13 cells are conflict-free and measured the same time (round 6, profiles/r06_tuning.txt): the conflicts are not what this kernel
    // waits for -- SQ_WAIT_INST_LDS is 3.6 % of its wave cycles.)
    __shared__ float4 sh_lds[HAS_SH ? PRE_BLOCK * 12 : 1];
    const bool sh_staged = HAS_SH && in.M == 16;
    const size_t base4 = (size_t)blockIdx.x * PRE_BLOCK * 12, total4 = (size_t)in.P * 12;
    const bool in_range = idx < in.P;
    const size_t ic = in_range ? (size_t)idx : 0, i3 = 3 * ic;
    // Every load that depends on nothing but the index is issued here, together, before the first wait: radius / tiles_touched / offset of the
    // view, the mean, scale and rotation (or the caller's covariance) and the twelve pieces of the SH rows.  Read where they are used they
    // were six memory round trips one after the other: rows, radius, tiles and offset, [slab rows], mean / scale / rotation, scale / rotation again.
    const int rad = in.radii[ic];
    const uint32_t tl = g.tiles_touched[ic], of = g.offsets[ic];
    const float mx = in.means3D[i3], my = in.means3D[i3 + 1], mz = in.means3D[i3 + 2];
    float scv[3] = {0.f, 0.f, 0.f}, cpre[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    tgs_v4f rq = {0.f, 0.f, 0.f, 0.f};
    if (HAS_SCALE_ROT) {
#pragma unroll
        for (int k = 0; k < 3; k++) scv[k] = in.scales[i3 + k];
        rq = reinterpret_cast<const tgs_v4f*>(in.rotations)[ic];
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) cpre[k] = in.cov3D_precomp[6 * ic + k];
    }
    if (sh_staged) {
        const tgs_v4f* s4 = reinterpret_cast<const tgs_v4f*>(in.shs);
        tgs_v4f r[12];
#pragma unroll
        for (int q = 0; q < 12; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; r[q] = s4[i < total4 ? i : total4 - 1]; }
        asm volatile("" ::: "memory");                      // (all of the above is asked for before the first wait)
#pragma unroll
        for (int q = 0; q < 12; q++) sh_lds[q * PRE_BLOCK + threadIdx.x] = make_float4(r[q].x, r[q].y, r[q].z, r[q].w);
        __syncthreads();
    }
    float a[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) a[k] = 0.f;
    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    const bool live = in_range && rad > 0;                  // backward.cu:156,367
    float dRGB[3] = {0.f, 0.f, 0.f};
    double cn[3];
    slab_sum(live, live ? tl : 0u, live ? of : 0u, b, a, cn);   // sum of this Gaussian's tile partials (wave-cooperative for big splats)
    if (live) {
        {
            float cov3d[6];
            const float rqv[4] = {rq.x, rq.y, rq.z, rq.w};
            if (HAS_SCALE_ROT) compute_cov3d(cam.scale_modifier, scv[0], scv[1], scv[2], make_float4(rq.x, rq.y, rq.z, rq.w), cov3d);
            else {
#pragma unroll
                for (int k = 0; k < 6; k++) cov3d[k] = cpre[k];
            }
            double dc64[6];
            cov2d_chain_bwd_f64<false>(mx, my, mz, cov3d, cam, V, PM, cn, a[3], a[4], dmean, dc64);
#pragma unroll
            for (int k = 0; k < 6; k++) dcov[k] = (float)dc64[k];
            if (HAS_SCALE_ROT) cov3d_bwd_f64(dc64, cam.scale_modifier, scv, rqv, dscale, drot);
        }
}

    if (HAS_SH) {
        // computeColorFromSH backward (backward.cu:20-139); culled Gaussians write zeros.
        // dL_dsh[k][c] = basis_k * dL_dRGB[c]; M = 16 rows (192 B, 16-B aligned) move as float4.
        float* dsh = in.dL_dsh + (size_t)idx * in.M * 3;
        const int ncoef = (in.D + 1) * (in.D + 1);
        float coef[16];
#pragma unroll
        for (int k = 0; k < 16; k++) coef[k] = 0.f;
        if (live) {
            float shv[48];
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
            const uint32_t cl = g.clamped[idx];
            dRGB[0] = a[0] * ((cl & 1u) ? 0.f : 1.f);
            dRGB[1] = a[1] * ((cl & 2u) ? 0.f : 1.f);
            dRGB[2] = a[2] * ((cl & 4u) ? 0.f : 1.f);
            const float ox = mx - camx, oy = my - camy, oz = mz - camz;
            const float len = sqrtf(ox * ox + oy * oy + oz * oz);
            const float x = ox / len, y = oy / len, z = oz / len;
            float gxv[3] = {0.f, 0.f, 0.f}, gyv[3] = {0.f, 0.f, 0.f}, gzv[3] = {0.f, 0.f, 0.f};
#define SH(k) shv[3 * (k) + c]
            coef[0] = SH_C0;
            if (in.D > 0) {
                coef[1] = -SH_C1 * y; coef[2] = SH_C1 * z; coef[3] = -SH_C1 * x;
#pragma unroll
                for (int c = 0; c < 3; c++) { gxv[c] = -SH_C1 * SH(3); gyv[c] = -SH_C1 * SH(1); gzv[c] = SH_C1 * SH(2); }
                if (in.D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    coef[4] = SH_C2_0 * xy; coef[5] = SH_C2_1 * yz; coef[6] = SH_C2_2 * (2.f * zz - xx - yy); coef[7] = SH_C2_3 * xz; coef[8] = SH_C2_4 * (xx - yy);
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        gxv[c] += SH_C2_0 * y * SH(4) + SH_C2_2 * 2.f * -x * SH(6) + SH_C2_3 * z * SH(7) + SH_C2_4 * 2.f * x * SH(8);
                        gyv[c] += SH_C2_0 * x * SH(4) + SH_C2_1 * z * SH(5) + SH_C2_2 * 2.f * -y * SH(6) + SH_C2_4 * 2.f * -y * SH(8);
                        gzv[c] += SH_C2_1 * y * SH(5) + SH_C2_2 * 2.f * 2.f * z * SH(6) + SH_C2_3 * x * SH(7);
                    }
                    if (in.D > 2) {
                        coef[9] = SH_C3_0 * y * (3.f * xx - yy); coef[10] = SH_C3_1 * xy * z; coef[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
                        coef[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy); coef[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
                        coef[14] = SH_C3_5 * z * (xx - yy); coef[15] = SH_C3_6 * x * (xx - 3.f * yy);
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            gxv[c] += (SH_C3_0 * SH(9) * 3.f * 2.f * xy + SH_C3_1 * SH(10) * yz + SH_C3_2 * SH(11) * -2.f * xy +
                                       SH_C3_3 * SH(12) * -3.f * 2.f * xz + SH_C3_4 * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                       SH_C3_5 * SH(14) * 2.f * xz + SH_C3_6 * SH(15) * 3.f * (xx - yy));
                            gyv[c] += (SH_C3_0 * SH(9) * 3.f * (xx - yy) + SH_C3_1 * SH(10) * xz + SH_C3_2 * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                                       SH_C3_3 * SH(12) * -3.f * 2.f * yz + SH_C3_4 * SH(13) * -2.f * xy + SH_C3_5 * SH(14) * -2.f * yz +
                                       SH_C3_6 * SH(15) * -3.f * 2.f * xy);
                            gzv[c] += (SH_C3_1 * SH(10) * xy + SH_C3_2 * SH(11) * 4.f * 2.f * yz + SH_C3_3 * SH(12) * 3.f * (2.f * zz - xx - yy) +
                                       SH_C3_4 * SH(13) * 4.f * 2.f * xz + SH_C3_5 * SH(14) * (xx - yy));
                        }
                    }
                }
            }
#undef SH
            const float ddx = gxv[0] * dRGB[0] + gxv[1] * dRGB[1] + gxv[2] * dRGB[2];
            const float ddy = gyv[0] * dRGB[0] + gyv[1] * dRGB[1] + gyv[2] * dRGB[2];
            const float ddz = gzv[0] * dRGB[0] + gzv[1] * dRGB[1] + gzv[2] * dRGB[2];
            // dnormvdv (auxiliary.h:107-117)
            const float sum2 = ox * ox + oy * oy + oz * oz;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((+sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
            dmean[1] += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
            dmean[2] += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
        }
        // coefficients above the active degree (and culled Gaussians) keep the reference's zeros (torch::zeros, rasterize_points.cu:157)
        if (sh_staged) {
            __syncthreads();                                   // every thread has consumed its SH row
#pragma unroll
            for (int q = 0; q < 12; q++) {
                float o[4];
#pragma unroll
                for (int t = 0; t < 4; t++) { const int i = 4 * q + t; o[t] = coef[i / 3] * dRGB[i % 3]; }
                sh_lds[threadIdx.x * 12 + q] = make_float4(o[0], o[1], o[2], o[3]);
            }
            __syncthreads();
            float4* d4 = reinterpret_cast<float4*>(in.dL_dsh);
            float4 prev[12];
            if (in.accumulate) {                               // (uniform) the twelve reads in flight together
#pragma unroll
                for (int q = 0; q < 12; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; prev[q] = d4[i < total4 ? i : total4 - 1]; }
            } else {
#pragma unroll
                for (int q = 0; q < 12; q++) prev[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 12; q++) {
                const size_t i = base4 + q * PRE_BLOCK + threadIdx.x;
                if (i < total4) {
                    float4 o = sh_lds[q * PRE_BLOCK + threadIdx.x];
                    if (in.accumulate) { o.x += prev[q].x; o.y += prev[q].y; o.z += prev[q].z; o.w += prev[q].w; }
                    d4[i] = o;
                }
            }
        } else if (in_range) {
            for (int k = 0; k < in.M; k++) {
                const float ck = k < 16 ? coef[k < 16 ? k : 0] : 0.f;
                const float o0 = in.accumulate ? dsh[3 * k] : 0.f, o1 = in.accumulate ? dsh[3 * k + 1] : 0.f, o2 = in.accumulate ? dsh[3 * k + 2] : 0.f;
                dsh[3 * k] = o0 + ck * dRGB[0]; dsh[3 * k + 1] = o1 + ck * dRGB[1]; dsh[3 * k + 2] = o2 + ck * dRGB[2];
            }
        }
    }

    // every output element is written (zeros for culled Gaussians)
    if (!in_range) return;
    in.dL_dmean2D[i3] = a[3]; in.dL_dmean2D[i3 + 1] = a[4]; in.dL_dmean2D[i3 + 2] = 0.f;   // .z never written: backward.cu:545-546
    // (dL_dconic, dL_dcolor on the SH path and dL_dcov3D on the scale / rotation path are intermediates of the reference's two-kernel backward that its
    // callers discard: a caller of tgs_backward_opt may pass NULL for them -- 52 of the ~300 B this kernel writes per Gaussian)
    if (in.dL_dconic) reinterpret_cast<float4*>(in.dL_dconic)[idx] = make_float4(a[5], a[6], 0.f, a[7]);   // .z never written: backward.cu:549-551
    if (in.accumulate) {
        // fused gradient accumulation of a multi-view batch (youreditableavatar_amd/multiview.py): += on the parameter gradients
        in.dL_dopacity[idx] += a[8];
        if (in.dL_dcolor) { in.dL_dcolor[i3] += a[0]; in.dL_dcolor[i3 + 1] += a[1]; in.dL_dcolor[i3 + 2] += a[2]; }   // NULL: intermediate on the SH path
        in.dL_dmean3D[i3] += dmean[0]; in.dL_dmean3D[i3 + 1] += dmean[1]; in.dL_dmean3D[i3 + 2] += dmean[2];
        if (in.dL_dcov3D) {                                                                                         // NULL: intermediate on the scale/rot path
#pragma unroll
            for (int i = 0; i < 6; i++) in.dL_dcov3D[6 * (size_t)idx + i] += dcov[i];
        }
        if (in.dL_dscale) { in.dL_dscale[i3] += dscale[0]; in.dL_dscale[i3 + 1] += dscale[1]; in.dL_dscale[i3 + 2] += dscale[2]; }
        if (in.dL_drot) { float4 p = reinterpret_cast<float4*>(in.dL_drot)[idx]; p.x += drot[0]; p.y += drot[1]; p.z += drot[2]; p.w += drot[3]; reinterpret_cast<float4*>(in.dL_drot)[idx] = p; }
        return;
    }
    in.dL_dopacity[idx] = a[8];
    if (in.dL_dcolor) { in.dL_dcolor[i3] = a[0]; in.dL_dcolor[i3 + 1] = a[1]; in.dL_dcolor[i3 + 2] = a[2]; }
    in.dL_dmean3D[i3] = dmean[0]; in.dL_dmean3D[i3 + 1] = dmean[1]; in.dL_dmean3D[i3 + 2] = dmean[2];
    if (in.dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) in.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    }
    if (in.dL_dscale) { in.dL_dscale[i3] = dscale[0]; in.dL_dscale[i3 + 1] = dscale[1]; in.dL_dscale[i3 + 2] = dscale[2]; }
    if (in.dL_drot) reinterpret_cast<float4*>(in.dL_drot)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
}

// ---------------------------------------------------------------------------------------------
// k_preprocess_bwd_batch: the per-Gaussian half for ALL views of a batch in one launch.  The view-independent inputs
// (mean, scale, rotation, the 192-B SH row) are read once and the parameter gradients -- dL_dsh above all: 192 B read +
// 192 B written per Gaussian per view in the one-view kernel's += mode -- are accumulated in registers over the views
// and stored (or added) once.  Per view only the view's own state is touched: pack line, tile partials, radii, flags.
// ---------------------------------------------------------------------------------------------
template <bool HAS_SH, bool HAS_SCALE_ROT>
__global__ __launch_bounds__(PRE_BLOCK, 2) void k_preprocess_bwd_batch(const BwdIn in, const BatchViews views)
{
    const uint32_t blk = (uint32_t)in.block0 + blockIdx.x;  // a launch may cover a range of Gaussians only (tgs_backward_batch_range)
    const int idx = (int)(blk * PRE_BLOCK + threadIdx.x);
    __shared__ float4 sh_lds[HAS_SH ? PRE_BLOCK * 12 : 1];
    const bool sh_staged = HAS_SH && in.M == 16;
    if (sh_staged) load_sh_rows_staged(in, sh_lds, blk);
    const bool in_range = idx < in.P;
    const float* sh_row = sh_staged ? reinterpret_cast<const float*>(&sh_lds[threadIdx.x * 12]) : (HAS_SH ? in.shs + (size_t)(in_range ? idx : 0) * in.M * 3 : nullptr);
    // radii / tiles_touched / offsets of this Gaussian in every view, fetched together up front (one memory round trip instead of two per
    // view in front of the view's slab rows: at 2 waves per SIMD the pass is bound by the latency of such chains) and parked in LDS --
    // each thread reads back only what it wrote, so no barrier
    __shared__ uint32_t pv_lds[BATCH_VIEWS][3][PRE_BLOCK];
    {
        uint32_t pr[BATCH_VIEWS], pt[BATCH_VIEWS], po[BATCH_VIEWS];
#pragma unroll
        for (int v = 0; v < BATCH_VIEWS; v++) {
            pr[v] = pt[v] = po[v] = 0u;
            if (v < views.n && in_range) { pr[v] = (uint32_t)views.v[v].radii[idx]; pt[v] = views.v[v].g.tiles_touched[idx]; po[v] = views.v[v].g.offsets[idx]; }
        }
        // a frame tgs_forward_async could not fit contributes nothing: its radii count as 0.  Read here, with the loads above, and not at the
        // head of every trip of the view loop -- there it was one more memory round trip in front of the trip's slab rows.
        uint32_t rej[BATCH_VIEWS];
#pragma unroll
        for (int v = 0; v < BATCH_VIEWS; v++) rej[v] = v < views.n ? (views.v[v].meta->error & META_ERR_CAPACITY) : 0u;
#pragma unroll
        for (int v = 0; v < BATCH_VIEWS; v++) { pv_lds[v][0][threadIdx.x] = rej[v] ? 0u : pr[v]; pv_lds[v][1][threadIdx.x] = pt[v]; pv_lds[v][2][threadIdx.x] = po[v]; }
    }
    float o48[48];                                         // dL_dsh row accumulated over the views (dead code without SH)
#pragma unroll
    for (int i = 0; i < 48; i++) o48[i] = 0.f;
    float dopacity = 0.f, dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    double dcsum[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // dL_dcov3D summed over the views in double (cov3d_bwd_f64 runs once, behind the loop)
    // view-independent inputs of the geometry chain, once: the mean and the 3D covariance (evaluated as the forward did, compute_cov3d)
    float gmx = 0.f, gmy = 0.f, gmz = 0.f, cov3d[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (in_range) {
        gmx = in.means3D[3 * (size_t)idx]; gmy = in.means3D[3 * (size_t)idx + 1]; gmz = in.means3D[3 * (size_t)idx + 2];
        load_cov3d<HAS_SCALE_ROT>(in, views.v[0].cam.scale_modifier, idx, cov3d);
    }
#pragma unroll 1
    for (int v = 0; v < views.n; v++) {
        const BatchView& vw = views.v[v];
        const bool live = in_range && (int)pv_lds[v][0][threadIdx.x] > 0;        // (0 for every Gaussian of a rejected frame)
        GaussTerms t;
        if (__builtin_amdgcn_ballot_w64(live) != 0) {
            const ViewMat V = load_mat(vw.cam.view), PM = load_mat(vw.cam.proj);
            pergauss_terms<HAS_SH>(idx, live, in.D, [&](int i) { return sh_row[i]; }, vw.cam, V, PM, vw.cam.campos[0], vw.cam.campos[1],
                                   vw.cam.campos[2], vw.g, vw.b, pv_lds[v][1][threadIdx.x], pv_lds[v][2][threadIdx.x], gmx, gmy, gmz, cov3d, t, dcsum);
        } else {
            t.a[0] = t.a[1] = t.a[2] = t.a[3] = t.a[4] = 0.f;
        }
        if (in_range) {
            const size_t i3 = 3 * (size_t)idx;
            vw.dL_dmean2D[i3] = live ? t.a[3] : 0.f; vw.dL_dmean2D[i3 + 1] = live ? t.a[4] : 0.f; vw.dL_dmean2D[i3 + 2] = 0.f;
            if (!HAS_SH && vw.dL_dcolor) { vw.dL_dcolor[i3] = live ? t.a[0] : 0.f; vw.dL_dcolor[i3 + 1] = live ? t.a[1] : 0.f; vw.dL_dcolor[i3 + 2] = live ? t.a[2] : 0.f; }
        }
        if (live) {
            dopacity += t.a[8];
#pragma unroll
            for (int k = 0; k < 3; k++) dmean[k] += t.dmean[k];
            if (HAS_SH) {
#pragma unroll
                for (int i = 0; i < 48; i++) o48[i] += t.coef[i / 3] * t.dRGB[i % 3];
            }
        }
    }
    finish_cov3d<HAS_SCALE_ROT>(in, views.v[0].cam.scale_modifier, idx, in_range, dcsum, dcov, dscale, drot);
    if (HAS_SH) {
        if (sh_staged) store_sh_rows_staged(in, sh_lds, [&](int i) { return o48[i]; }, blk);
        else if (in_range) {
            float* dsh = in.dL_dsh + (size_t)idx * in.M * 3;
            for (int i = 0; i < in.M * 3; i++) {
                float o = 0.f;
#pragma unroll
                for (int j = 0; j < 48; j++) o = (i == j) ? o48[j] : o;           // no dynamic indexing: keeps the accumulators in registers
                dsh[i] = (in.accumulate ? dsh[i] : 0.f) + o;
            }
        }
    }
    if (!in_range) return;
    const float nocolor[3] = {0.f, 0.f, 0.f};
    BwdIn shared = in;
    shared.dL_dcolor = nullptr;                              // per-view colours have per-view gradients (BatchView::dL_dcolor)
    store_param_grads(shared, idx, dopacity, nocolor, dmean, dcov, dscale, drot);
}

// The same pass with TWO threads per Gaussian (SH path, M == 16): a workgroup takes 128 Gaussians, its waves 0-1 do the geometry half
// of every view (slab sums, cov2D / projection / cov3D backward: the 17 accumulators of pergauss_terms without its SH block), waves 2-3 the
// colour half (the colour sums of the slab rows, sh_backward_terms, the 48 dL_dsh accumulators).  The one-thread kernel needs 256 VGPRs --
// 2 waves per SIMD -- and is bound by the latency of its own dependent arithmetic and loads at that occupancy (278 us per 8 views, alone
// on the GPU at the end of the step); split, the kernel needs 168 and three waves per SIMD are resident, each with half the work: 239 us
// (forced into 128 VGPRs for four waves it spills 144 B and takes 307).  The halves meet once, at the end: the colour half's share of
// dL_dmean3D goes through LDS to the geometry half, which stores the parameter gradients.
// The colour sums of the Gaussian's slab rows (a[0..2] of slab_sum: the rows' first 16 bytes), four rows per trip with their loads issued
// together: row after row a lane pays one memory latency per tile of its splat and the wave waits for its lane with the most (a 3 x 3
// rectangle: nine in a row, per view).  The rows past the last are the last one again (a valid address, so that no load sits under a branch
// of its own: a load under `if` is waited for inside it) and are not added.  Same order of the additions as slab_sum.
// The splat's first four rows come in registers (q0..q3: asked for one view ahead by the split pass).
__device__ __forceinline__ void slab_sum_rgb_pre(bool live, uint32_t tiles_in, uint32_t off_in, const BinState& b, tgs_v4f q0, tgs_v4f q1, tgs_v4f q2, tgs_v4f q3, float (&rgb)[3])
{
    const int lane = threadIdx.x & 63;
    const uint32_t tiles = live ? tiles_in : 0u, off = live ? off_in : 0u;
    rgb[0] = rgb[1] = rgb[2] = 0.f;
    const uint32_t coop = slab_coop_threshold(tiles);
    unsigned long long big = __builtin_amdgcn_ballot_w64(tiles >= coop);
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)tiles, src), o = (uint32_t)__builtin_amdgcn_readlane((int)off, src);
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
        for (uint32_t k = lane; k < n; k += 64) { const float4 r0 = b.slab[(size_t)(o + k) * SLAB_ROW]; p0 += r0.x; p1 += r0.y; p2 += r0.z; }
        const float t0 = wave_sum(p0), t1 = wave_sum(p1), t2 = wave_sum(p2);
        if (lane == src) { rgb[0] = t0; rgb[1] = t1; rgb[2] = t2; }
    }
    if (tiles > 0u && tiles < coop) {
        rgb[0] += q0.x; rgb[1] += q0.y; rgb[2] += q0.z;
        if (1u < tiles) { rgb[0] += q1.x; rgb[1] += q1.y; rgb[2] += q1.z; }
        if (2u < tiles) { rgb[0] += q2.x; rgb[1] += q2.y; rgb[2] += q2.z; }
        if (3u < tiles) { rgb[0] += q3.x; rgb[1] += q3.y; rgb[2] += q3.z; }
        const tgs_v4f* row = reinterpret_cast<const tgs_v4f*>(b.slab) + (size_t)off * SLAB_ROW;
        for (uint32_t k = 4; k < tiles; k += 4) {
            const uint32_t last = tiles - 1u;
            const tgs_v4f r0 = row[(size_t)k * SLAB_ROW], r1 = row[(size_t)min(k + 1u, last) * SLAB_ROW], r2 = row[(size_t)min(k + 2u, last) * SLAB_ROW],
                          r3 = row[(size_t)min(k + 3u, last) * SLAB_ROW];
            asm volatile("" ::: "memory");
            rgb[0] += r0.x; rgb[1] += r0.y; rgb[2] += r0.z;
            if (k + 1u < tiles) { rgb[0] += r1.x; rgb[1] += r1.y; rgb[2] += r1.z; }
            if (k + 2u < tiles) { rgb[0] += r2.x; rgb[1] += r2.y; rgb[2] += r2.z; }
            if (k + 3u < tiles) { rgb[0] += r3.x; rgb[1] += r3.y; rgb[2] += r3.z; }
        }
    }
}

constexpr int SPLIT_G = PRE_BLOCK / 2;                      // Gaussians per workgroup of the split pass
#ifndef TGS_SPLIT_WAVES
#define TGS_SPLIT_WAVES 3
#endif
template <bool HAS_SCALE_ROT>
__global__ __launch_bounds__(PRE_BLOCK, TGS_SPLIT_WAVES) void k_preprocess_bwd_batch_split(const BwdIn in, const BatchViews views)
{
    __shared__ float4 sh_lds[SPLIT_G * 12];                 // the SH rows of the 128 Gaussians in, their dL_dsh rows out
    __shared__ uint32_t pv_lds[BATCH_VIEWS][3][SPLIT_G];    // radii / tiles_touched / offsets of every Gaussian in every view (as in the one-thread kernel), shared by its two threads
    __shared__ float dm_lds[3][SPLIT_G];                    // the colour half's share of dL_dmean3D
    __shared__ float gc_lds[9][SPLIT_G];                    // geometry half: mean and 3D covariance of its Gaussian
    const bool colour = threadIdx.x >= SPLIT_G;             // wave-uniform  (odd workgroups with the roles of their wave pairs swapped -- in case a
                                                            // workgroup's wave i always lands on SIMD i -- measured no different: 0.1962 / 0.1965 ms)
    const int gl = threadIdx.x & (SPLIT_G - 1);             // Gaussian of the workgroup
    const size_t gbase = ((size_t)in.block0 * 2 + blockIdx.x) * SPLIT_G;
    const int idx = (int)(gbase + gl);
    const bool in_range = idx < in.P;
    // Prologue: EVERY load of it is issued before the first of them is waited for -- the six 16-B pieces of the SH rows (128 x 12 float4,
    // coalesced), radii / tiles_touched / offsets and the rejected flag of half of the views (waves 0-1 take views [0, 4), waves 2-3 views
    // [4, 8) of their Gaussian), the mean, and (geometry half) scale and rotation.  Indices are clamped instead of the loads predicated: a
    // load under `if (i < n)` whose value goes to LDS is waited for inside its branch, one memory latency after the other.
    constexpr int HV = BATCH_VIEWS / 2;
    const int ic = in_range ? idx : 0;
    tgs_v4f shr[6];                                        // (first-class vectors, not float4 structs: those went through scratch around the asm statement below)
    {
        const tgs_v4f* s4 = reinterpret_cast<const tgs_v4f*>(in.shs);
        const size_t base4 = gbase * 12, total4 = (size_t)in.P * 12;
#pragma unroll
        for (int q = 0; q < 6; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; shr[q] = s4[i < total4 ? i : total4 - 1]; }
    }
    uint32_t pr[HV], pt[HV], po[HV], rej[HV];
    const int vfirst = __builtin_amdgcn_readfirstlane(colour ? HV : 0);      // (wave-uniform: the views' pointers come by scalar loads)
#pragma unroll
    for (int j = 0; j < HV; j++) {
        const int v = vfirst + j, vc = v < views.n ? v : 0;
        pr[j] = (uint32_t)views.v[vc].radii[ic]; pt[j] = views.v[vc].g.tiles_touched[ic]; po[j] = views.v[vc].g.offsets[ic];
        rej[j] = views.v[vc].meta->error & META_ERR_CAPACITY;            // a frame tgs_forward_async could not fit contributes nothing: its radii count as 0
    }
    const float mx = in.means3D[3 * (size_t)ic], my = in.means3D[3 * (size_t)ic + 1], mz = in.means3D[3 * (size_t)ic + 2];
    float sc[3] = {0.f, 0.f, 0.f}, cpre[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    tgs_v4f rq = {0.f, 0.f, 0.f, 0.f};
    if (HAS_SCALE_ROT) {
#pragma unroll
        for (int k = 0; k < 3; k++) sc[k] = in.scales[3 * (size_t)ic + k];
        rq = reinterpret_cast<const tgs_v4f*>(in.rotations)[ic];
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) cpre[k] = in.cov3D_precomp[6 * (size_t)ic + k];
    }
    asm volatile("" ::: "memory");                         // (every load above is issued before the first wait: the compiler sinks those the covariance block uses into it)
#pragma unroll
    for (int q = 0; q < 6; q++) sh_lds[q * PRE_BLOCK + threadIdx.x] = make_float4(shr[q].x, shr[q].y, shr[q].z, shr[q].w);
#pragma unroll
    for (int j = 0; j < HV; j++) {
        const int v = vfirst + j;
        const bool on = in_range && v < views.n && !rej[j];
        pv_lds[v][0][gl] = on ? pr[j] : 0u; pv_lds[v][1][gl] = on ? pt[j] : 0u; pv_lds[v][2][gl] = on ? po[j] : 0u;
    }
    if (!colour) {
        // view-independent inputs of the geometry chain, once: the mean and the 3D covariance (compute_cov3d, as the forward evaluated it).
        // Parked in LDS, each thread its own nine words: held in registers across the view loop they push the kernel over the 168 VGPRs of
        // three waves per SIMD.
        float cov3d[6];
        if (HAS_SCALE_ROT) compute_cov3d(views.v[0].cam.scale_modifier, sc[0], sc[1], sc[2], make_float4(rq.x, rq.y, rq.z, rq.w), cov3d);
        else {
#pragma unroll
            for (int k = 0; k < 6; k++) cov3d[k] = cpre[k];
        }
        gc_lds[0][gl] = mx; gc_lds[1][gl] = my; gc_lds[2][gl] = mz;
#pragma unroll
        for (int k = 0; k < 6; k++) gc_lds[3 + k][gl] = cov3d[k];
    }
    __syncthreads();
    if (colour) {
        const float* sh_row = reinterpret_cast<const float*>(&sh_lds[gl * 12]);
        float o48[48];
#pragma unroll
        for (int i = 0; i < 48; i++) o48[i] = 0.f;
        float dmean[3] = {0.f, 0.f, 0.f};
        // One view AHEAD: the first four slab rows of the Gaussian, its clamp bits and the camera position of view v + 1 are asked for before
        // view v is evaluated -- a trip of this loop was [rows' memory round trip] + [~300 instructions], one after the other, at three waves
        // per SIMD; now the round trip of the next view runs under the arithmetic of this one.
        tgs_v4f nq0 = {0.f, 0.f, 0.f, 0.f}, nq1 = nq0, nq2 = nq0, nq3 = nq0;
        uint32_t ncl = 0u;
        float ncx = 0.f, ncy = 0.f, ncz = 0.f;
        auto ask = [&](int v) {
            const BatchView& w = views.v[v];
            const uint32_t tl = pv_lds[v][1][gl], of = pv_lds[v][2][gl];
            ncl = w.g.clamped[ic];
            ncx = w.cam.campos[0]; ncy = w.cam.campos[1]; ncz = w.cam.campos[2];
            if ((int)pv_lds[v][0][gl] > 0 && tl > 0u && tl < SLAB_COOP) {      // (0 radii outside the range and in rejected frames: no row is touched)
                const tgs_v4f* row = reinterpret_cast<const tgs_v4f*>(w.b.slab) + (size_t)of * SLAB_ROW;
                const uint32_t last = tl - 1u;
                nq0 = row[0]; nq1 = row[(size_t)min(1u, last) * SLAB_ROW]; nq2 = row[(size_t)min(2u, last) * SLAB_ROW]; nq3 = row[(size_t)min(3u, last) * SLAB_ROW];
            }
        };
        const int nv = views.n;
        if (nv > 0) ask(0);
#pragma unroll 1
        for (int v = 0; v < nv; v++) {
            const BatchView& vw = views.v[v];
            const bool live = in_range && (int)pv_lds[v][0][gl] > 0;
            const tgs_v4f q0 = nq0, q1 = nq1, q2 = nq2, q3 = nq3;
            const uint32_t cl = ncl;
            const float cpx = ncx, cpy = ncy, cpz = ncz;
            if (v + 1 < nv) ask(v + 1);
            asm volatile("" ::: "memory");                 // (the next view's loads stay here)
            if (__builtin_amdgcn_ballot_w64(live) == 0) continue;
            float rgb[3];
            slab_sum_rgb_pre(live, pv_lds[v][1][gl], pv_lds[v][2][gl], vw.b, q0, q1, q2, q3, rgb);
            if (live) {
                float coef[16], dRGB[3];
#pragma unroll
                for (int k = 0; k < 16; k++) coef[k] = 0.f;
                sh_backward_terms(in.D, [&](int i) { return sh_row[i]; }, cl, rgb[0], rgb[1], rgb[2], mx, my, mz, cpx, cpy, cpz, coef, dRGB, dmean);
#pragma unroll
                for (int i = 0; i < 48; i++) o48[i] += coef[i / 3] * dRGB[i % 3];
            }
        }
        dm_lds[0][gl] = dmean[0]; dm_lds[1][gl] = dmean[1]; dm_lds[2][gl] = dmean[2];
        __syncthreads();                                   // (A) the geometry half has the colour half's dL_dmean3D; every SH row has been consumed
        if (in.dsh_plane == 0) {
#pragma unroll
            for (int q = 0; q < 12; q++) sh_lds[gl * 12 + q] = make_float4(o48[4 * q], o48[4 * q + 1], o48[4 * q + 2], o48[4 * q + 3]);
        } else {                                           // level-major output: plane k of the workgroup = 128 x 3 consecutive floats (lane stride 3 words: no bank conflict)
            float* lf = reinterpret_cast<float*>(sh_lds);
#pragma unroll
            for (int i = 0; i < 48; i++) lf[(i / 3) * (3 * SPLIT_G) + 3 * gl + (i % 3)] = o48[i];
        }
        __syncthreads();                                   // (B) the dL_dsh rows are staged
    } else {
        float dopacity = 0.f, dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
        double dcsum[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int v = 0; v < views.n; v++) {
            const BatchView& vw = views.v[v];
            const bool live = in_range && (int)pv_lds[v][0][gl] > 0;
            GaussTerms t;
            if (__builtin_amdgcn_ballot_w64(live) != 0) {
                const ViewMat V = load_mat(vw.cam.view), PM = load_mat(vw.cam.proj);
                asm volatile("" ::: "memory");             // (read the nine words here, not in front of the loop)
                float cov3d[6];
#pragma unroll
                for (int k = 0; k < 6; k++) cov3d[k] = gc_lds[3 + k][gl];
                pergauss_terms<false>(idx, live, in.D, [&](int) { return 0.f; }, vw.cam, V, PM, vw.cam.campos[0], vw.cam.campos[1],
                                      vw.cam.campos[2], vw.g, vw.b, pv_lds[v][1][gl], pv_lds[v][2][gl], gc_lds[0][gl], gc_lds[1][gl], gc_lds[2][gl], cov3d, t, dcsum);
            } else {
                t.a[3] = t.a[4] = 0.f;
            }
            if (in_range) {
                const size_t i3 = 3 * (size_t)idx;
                vw.dL_dmean2D[i3] = live ? t.a[3] : 0.f; vw.dL_dmean2D[i3 + 1] = live ? t.a[4] : 0.f; vw.dL_dmean2D[i3 + 2] = 0.f;
            }
            if (live) {                                    // (scale / rotation gradients: once, from dcsum, behind the loop)
                dopacity += t.a[8];
#pragma unroll
                for (int k = 0; k < 3; k++) dmean[k] += t.dmean[k];
            }
        }
        finish_cov3d<HAS_SCALE_ROT>(in, views.v[0].cam.scale_modifier, idx, in_range, dcsum, dcov, dscale, drot);
        __syncthreads();                                   // (A)
        if (in_range) {
            dmean[0] += dm_lds[0][gl]; dmean[1] += dm_lds[1][gl]; dmean[2] += dm_lds[2][gl];
            const float nocolor[3] = {0.f, 0.f, 0.f};
            BwdIn shared = in;
            shared.dL_dcolor = nullptr;
            store_param_grads(shared, idx, dopacity, nocolor, dmean, dcov, dscale, drot);
        }
        __syncthreads();                                   // (B)
    }
    if (in.dsh_plane != 0) {
        // LEVEL-MAJOR dL_dsh (round 6): coefficient k of all Gaussians is one plane, so the (D + 1)^2 LIVE coefficients of a step rendered below the
        // stored degree are ONE contiguous piece of the gradient buffer -- a data-parallel step hands that piece to the collective as it is
        // (FlatGradients(level_major=True)), where the row-major layout needed a strided pack and unpack of 28 MB around it (round 5: 0.128 ms per
        // step).  The workgroup's share of plane k: 384 consecutive floats = 96 float4; 16 planes x 96 = the same 6 coalesced 16-B stores per thread.
        float* dsh = in.dL_dsh;
        const long long left = (long long)in.P - (long long)gbase;                                          // (<= 0: a workgroup behind the last Gaussian -- the grid covers whole 256-blocks)
        const uint32_t nval = 3u * (uint32_t)(left <= 0 ? 0 : (left < (long long)SPLIT_G ? left : (long long)SPLIT_G));   // valid floats of this workgroup per plane
        float* dst[6]; uint32_t f0[6];
        float4 prev[6];
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const uint32_t j = (uint32_t)(q * PRE_BLOCK) + threadIdx.x, k = (j * 683u) >> 16;                  // j / 96 for j < 1536
            f0[q] = 4u * (j - 96u * k);
            dst[q] = dsh + (size_t)k * (size_t)in.dsh_plane + 3 * gbase + f0[q];
            prev[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (in.accumulate) {
#pragma unroll
            for (int q = 0; q < 6; q++) if (f0[q] + 3u < nval) prev[q] = *reinterpret_cast<const float4*>(dst[q]);
        }
#pragma unroll
        for (int q = 0; q < 6; q++) {
            float4 o = sh_lds[q * PRE_BLOCK + threadIdx.x];
            if (f0[q] + 3u < nval) {
                o.x += prev[q].x; o.y += prev[q].y; o.z += prev[q].z; o.w += prev[q].w;
                *reinterpret_cast<float4*>(dst[q]) = o;
            } else {                                        // the last workgroup's ragged end: float by float
                const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int i = 0; i < 4; i++) if (f0[q] + (uint32_t)i < nval) dst[q][i] = (in.accumulate ? dst[q][i] : 0.f) + ov[i];
            }
        }
    } else {   // dL_dsh rows out: 6 coalesced 16-B stores per thread
        float4* d4 = reinterpret_cast<float4*>(in.dL_dsh);
        const size_t base4 = gbase * 12, total4 = (size_t)in.P * 12;
        float4 prev[6];
        if (in.accumulate) {                               // (uniform) the six reads in flight together
#pragma unroll
            for (int q = 0; q < 6; q++) { const size_t i = base4 + q * PRE_BLOCK + threadIdx.x; prev[q] = d4[i < total4 ? i : total4 - 1]; }
        } else {
#pragma unroll
            for (int q = 0; q < 6; q++) prev[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const size_t i = base4 + q * PRE_BLOCK + threadIdx.x;
            if (i < total4) {
                float4 o = sh_lds[q * PRE_BLOCK + threadIdx.x];
                if (in.accumulate) { o.x += prev[q].x; o.y += prev[q].y; o.z += prev[q].z; o.w += prev[q].w; }
                d4[i] = o;
            }
        }
    }
}

// hardware self-test of wave_reduce36: in[64][36] -> out[4][9] (row e, component k)
__global__ void k_selftest_reduce36(const float* in, float* out)
{
    float v[36], r[9];
#pragma unroll
    for (int i = 0; i < 36; i++) v[i] = in[threadIdx.x * 36 + i];
    wave_reduce36(v, r);
    if ((threadIdx.x & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) out[(threadIdx.x >> 4) * 9 + k] = r[k];
    }
}
void launch_selftest_reduce36(hipStream_t st, const float* in, float* out)
{
    hipLaunchKernelGGL(k_selftest_reduce36, dim3(1), dim3(64), 0, st, in, out);
}

// `tiles`: leading entries of tile_order to visit -- all of them, or the caller's bound on the tiles with instances (the rest is empty)
// tiles: leading entries of tile_order that can hold instances (all, or the caller's bound); mid_tiles (light != 0): how many of them can hold
// >= LIGHT_MAX instances -- the caller's bound, or `tiles`.  T: tiles of the image.
void launch_render_bwd(hipStream_t st, const ImgState& s, const BinState& b, int W, int H, uint32_t gx, uint32_t tiles, const float* bg, const float* dL_dpix,
                       bool deterministic, uint32_t mid_tiles, int light, uint32_t T)
{
    // (-DTGS_FAST_MATH=0 builds always take the fixed-order kernel: it evaluates exp / the divisions in their accurate forms)
    if (deterministic || !TGS_FAST_MATH) { hipLaunchKernelGGL(k_render_bwd_det, dim3(tiles), dim3(256), 0, st, s, b, W, H, gx, bg, dL_dpix); return; }
    if (!light) { hipLaunchKernelGGL(k_render_bwd, dim3(tiles), dim3(BWD_THREADS), 0, st, s, b, W, H, gx, bg, dL_dpix, 0, T); return; }
    const uint32_t heavy = mid_tiles < tiles ? mid_tiles : tiles;
    // one-tile workgroups for the (bound on the) tiles with >= LIGHT_MAX instances + light groups for the rest, three tiles each.  With exact
    // counts (heavy = n_mid, tiles = n_nonempty) that is n_mid + ceil((n_nonempty - n_mid) / 3); with bounds it is an upper bound of it.
    const uint32_t grid = heavy + (tiles - heavy + BWD_LIGHT_PER_WG - 1) / BWD_LIGHT_PER_WG;      // (largest when n_mid reaches its bound)
    hipLaunchKernelGGL(k_render_bwd, dim3(grid > 0 ? grid : 1u), dim3(BWD_THREADS), 0, st, s, b, W, H, gx, bg, dL_dpix, 1, T);
}
void launch_preprocess_bwd(hipStream_t st, const BwdIn& in, const CamParams& cam, const GeomState& g, const BinState& b)
{
    const dim3 grid((unsigned)n_blocks(in.P)), blk(PRE_BLOCK);
    const bool sh = in.shs != nullptr, sr = in.scales != nullptr;
    if (sh && sr) hipLaunchKernelGGL((k_preprocess_bwd<true, true>), grid, blk, 0, st, in, cam, g, b);
    else if (sh) hipLaunchKernelGGL((k_preprocess_bwd<true, false>), grid, blk, 0, st, in, cam, g, b);
    else if (sr) hipLaunchKernelGGL((k_preprocess_bwd<false, true>), grid, blk, 0, st, in, cam, g, b);
    else hipLaunchKernelGGL((k_preprocess_bwd<false, false>), grid, blk, 0, st, in, cam, g, b);
}

void launch_preprocess_bwd_batch(hipStream_t st, const BwdIn& in, const BatchViews& views)
{
    const dim3 grid((unsigned)(in.nblocks > 0 ? in.nblocks : n_blocks(in.P))), blk(PRE_BLOCK);
    const bool sh = in.shs != nullptr, sr = in.scales != nullptr;
    if (sh && in.M == 16) {                                 // two threads per Gaussian: 128 Gaussians per workgroup (the one-thread kernel below: per-view colours, other M; tgs_api refuses dsh_plane != 0 there)
        const dim3 grid2(2 * grid.x);
        if (sr) hipLaunchKernelGGL((k_preprocess_bwd_batch_split<true>), grid2, blk, 0, st, in, views);
        else hipLaunchKernelGGL((k_preprocess_bwd_batch_split<false>), grid2, blk, 0, st, in, views);
        return;
    }
    if (sh && sr) hipLaunchKernelGGL((k_preprocess_bwd_batch<true, true>), grid, blk, 0, st, in, views);
    else if (sh) hipLaunchKernelGGL((k_preprocess_bwd_batch<true, false>), grid, blk, 0, st, in, views);
    else if (sr) hipLaunchKernelGGL((k_preprocess_bwd_batch<false, true>), grid, blk, 0, st, in, views);
    else hipLaunchKernelGGL((k_preprocess_bwd_batch<false, false>), grid, blk, 0, st, in, views);
}

}  // namespace tgs
