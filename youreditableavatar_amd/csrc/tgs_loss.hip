// tgs_loss.hip -- the trainers' photometric loss, value and gradient, in two passes over the image ("next" row 2).
//
//   loss = (1 - f) * mean|pred - gt| + f * (1 - mean(ssim_map(pred, gt)))          f = dssim_factor
// (Edit_core/utils/loss_utils.py:17-18, :39-63 composed as in tetgs_texture/refine.py:245-247).  The reference builds
// the SSIM statistics with five zero-padded depthwise 11x11 convolutions plus ~15 element-wise kernels and lets
// autograd run them backwards; here
//   k_ssim_stats_stream : a wave walks down a vertical strip of one plane (lane = column): separable 11-tap Gaussian of
//                  (x, y, x^2 + y^2, xy) -> ssim_map, its three partial derivatives with respect to the windowed statistics, and the
//                  wave's partial sums of ssim_map and |x - y|;
//   k_loss_reduce: fixed-order sum of the partials -> loss, ssim, l1 (no float atomics: reproducible);
//   k_ssim_grad_stream : the adjoint of the same separable window applied to the three derivative maps
//                  d loss / d x(p) = gl * sign(x - y) + gs * (conv(dM1)(p) + 2 x(p) conv(dX2)(p) + y(p) conv(dXY)(p)).
// No LDS, no barrier.  The statistics pass is bound by vector issue (53 % of its wave cycles issuing, 37 % waiting for an issue slot, 9 % for
// memory: profiles/r05_ssim_counters.txt), the gradient pass by HBM (5 planes read with a 64 / 54 overlap + 1 written: ~5 TB/s).
#include "tgs_device.hpp"
#include <cmath>
#include <cstdlib>

namespace tgs {

constexpr int LR = 5;                                   // window radius (window_size 11, loss_utils.py:39)
constexpr int NTAP = 2 * LR + 1;

struct LossWin { float w[NTAP]; };

// out[0] = loss, out[1] = ssim, out[2] = l1; sums in double, fixed order
__global__ __launch_bounds__(1024) void k_loss_reduce(int n, const float2* __restrict__ partial, double inv_count, float f, float* __restrict__ out)
{
    __shared__ double rm[16], rl[16];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) { const float2 p = partial[i]; a += (double)p.x; b += (double)p.y; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { rm[threadIdx.x >> 6] = a; rl[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sm = 0.0, sl = 0.0;
        for (int i = 0; i < 16; i++) { sm += rm[i]; sl += rl[i]; }
        const double ssim = sm * inv_count, l1 = sl * inv_count;
        out[0] = (float)((1.0 - (double)f) * l1 + (double)f * (1.0 - ssim));
        out[1] = (float)ssim;
        out[2] = (float)l1;
    }
}

// STREAMING kernels without LDS and without barriers (round 3).  A wave owns a vertical strip of the plane -- lane = column, 64 input
// columns = 54 output columns + the window's 5-column halo on either side -- and walks down SS_SEG output rows:
//   per input row: one coalesced 4-B load per lane and map; the horizontal 11-tap window across the lanes (below); the vertical window as 11
//   running sums per lane and map (a row adds w[k] * H to the 11 output rows it belongs to; the row loop is unrolled by 11 so that the slot
//   of a running sum is a compile-time register); the output row 5 rows behind is finished: SSIM map and its derivatives (stats pass) / the
//   gradient (gradient pass), one store per map.
// Cost of the scheme: 64 / 54 of the input loads (neighbouring strips overlap by 10 columns, L2 hits) and 10 warm-up rows per segment.
// (The tiled LDS kernels of rounds 1-2 -- 32x32 tiles with halo, three barriers per tile: 1.9 / 2.7 TB/s of their algorithmic bytes -- left the
//  source in round 5; DESIGN_HISTORY.md keeps their numbers.)
// Round 4: the row loads are unconditional (clamped into the image, times 0 or 1) and the row loop walks whole groups of 11 rows: under `if`
// each pair of loads was waited for with vmcnt(0) -- one memory round trip per row instead of three rows in flight: 104 + 78 -> 96.5 + 76 us.
// Measured and not kept: TWO columns per lane (round 4; 125 / 106 VGPRs, half as many waves: 113 + 95 us instead of 104 + 78 at 3 x 2048 x 2048);
// round 5: the window's neighbours read from a wave-private LDS line instead of DPP shifts (stats 111.6 -> 99.4 us but 108 VGPRs, the gradient
// pass slower: 60.5 -> 68.8 us), and segments short enough to fill six workgroup slots per CU in one round (43 rows: 109.6 -> 115.4 us -- the
// ten warm-up rows per segment cost more than the occupancy returns).
// Round 5: THE SUMS TRAVEL, not the inputs (stats_row): 109.6 -> 94 us for the statistics pass at 3 x 2048 x 2048; row addresses are a scalar
// base + one 32-bit lane offset (at_b).  L1 + SSIM value and gradient 173.9 -> 160-163 us.
// =====================================================================================================================================
constexpr int SS_OUT = WAVE - 2 * LR;       // 54 output columns per wave
#ifndef TGS_SS_SEG
#define TGS_SS_SEG 64
#endif
constexpr int SS_SEG = TGS_SS_SEG;          // output rows per wave

__device__ __forceinline__ float wave_shr1(float v)     // lane i <- lane i - 1 (lane 0 gets 0: never used)
{
    // (bound_ctrl: the lane without a source reads 0, so the compiler need not preset the destination -- one instruction, not two)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

struct StripGeom { int col_in, col_out, y0, y1; bool in_x, out_ok, own_col; size_t plane; uint32_t out_b; };
// Element of a row whose base is the same for the whole wave, at a per-lane BYTE offset kept in 32 bits: the address is a scalar base plus a
// vector offset (global_load_dword v, v_off, s[base:base+1]) -- no 64-bit vector add per access, one offset register for every map of the pass.
__device__ __forceinline__ const float& at_b(const float* row, uint32_t byte_off) { return *(const float*)((const char*)row + byte_off); }
__device__ __forceinline__ float& at_b(float* row, uint32_t byte_off) { return *(float*)((char*)row + byte_off); }
__device__ __forceinline__ bool strip_geom(StripGeom& g, int H, int W, int nstrips)
{
    const int lane = threadIdx.x & 63, strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (strip >= nstrips) return false;
    g.col_in = strip * SS_OUT - LR + lane;
    g.col_out = g.col_in - LR;
    g.in_x = g.col_in >= 0 && g.col_in < W;
    g.out_ok = lane >= 2 * LR && g.col_out < W;
    g.own_col = lane >= LR && lane < LR + SS_OUT && g.in_x;        // the columns whose |x - y| this wave adds up
    g.y0 = blockIdx.y * SS_SEG;
    g.y1 = min(g.y0 + SS_SEG, H);
    g.plane = (size_t)blockIdx.z * H * W;
    g.out_b = (uint32_t)max(g.col_out, 0) * 4u;
    return true;
}

// one input row of the stats pass in phase P (= row counter mod 11): horizontal window of the five maps, scatter into the running sums,
// finish output row r - 5
// Round 4: FOUR maps, not five.  The SSIM map needs E[x^2] and E[y^2] only through sigma1^2 + sigma2^2 = E[x^2] + E[y^2] - mu1^2 - mu2^2
// (loss_utils.py:53-58: D2 = sigma1_sq + sigma2_sq + C2), so the two windows are one window over x^2 + y^2: one map less in the vertical
// scatter (11 FMAs per row) and 11 running sums less per lane (86 -> 75 VGPRs: six waves per SIMD instead of five).
template <int P>
__device__ __forceinline__ void stats_row(float x, float y, int r, const StripGeom& g, const LossWin& win, float (&V)[4][NTAP], int W,
                                          float* __restrict__ dM1, float* __restrict__ dX2, float* __restrict__ dXY, float& s_map)
{
    // Horizontal window, round 5: the SUMS travel, not the inputs.  h = w[10] x; ten times h = shr1(h) + w[j] x (j = 9 .. 0) leaves
    // sum_j w[j] x(c - j) in lane c -- the window of output column c - 5 -- with ONE instruction per tap and map (v_add_f32_dpp; the shift
    // rides on the add) instead of a shift and a multiply-add, and the products w[j] x are six per map, not eleven: the window is symmetric
    // (w[j] = w[10 - j], loss_utils.py:23-25).
    const float ss = x * x + y * y, xy = x * y;
    float tx[LR + 1], ty[LR + 1], ts[LR + 1], tq[LR + 1];
#pragma unroll
    for (int j = 0; j <= LR; j++) { const float w = win.w[j]; tx[j] = w * x; ty[j] = w * y; ts[j] = w * ss; tq[j] = w * xy; }
    float hx = tx[0], hy = ty[0], hss = ts[0], hxy = tq[0];                    // (w[10] = w[0])
    // ONE asm block, the four chains interleaved: a DPP read is three instructions behind the write it depends on (the hazard asks for two wait
    // states) and nothing of the compiler's can come between them -- its hazard recogniser does not look into inline asm, so the block opens
    // with the s_nop that covers whatever was written just in front of it.  (One piece per tap with its own s_nop measured + 2 us.)
#define TGS_TAP4(K) "v_add_f32_dpp %[hx], %[hx], %[x" #K "] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
                    "v_add_f32_dpp %[hy], %[hy], %[y" #K "] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
                    "v_add_f32_dpp %[hs], %[hs], %[s" #K "] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
                    "v_add_f32_dpp %[hq], %[hq], %[q" #K "] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("s_nop 1\n\t" TGS_TAP4(1) TGS_TAP4(2) TGS_TAP4(3) TGS_TAP4(4) TGS_TAP4(5) TGS_TAP4(4) TGS_TAP4(3) TGS_TAP4(2) TGS_TAP4(1) TGS_TAP4(0)
        : [hx] "+&v"(hx), [hy] "+&v"(hy), [hs] "+&v"(hss), [hq] "+&v"(hxy)
        : [x0] "v"(tx[0]), [x1] "v"(tx[1]), [x2] "v"(tx[2]), [x3] "v"(tx[3]), [x4] "v"(tx[4]), [x5] "v"(tx[5]),
          [y0] "v"(ty[0]), [y1] "v"(ty[1]), [y2] "v"(ty[2]), [y3] "v"(ty[3]), [y4] "v"(ty[4]), [y5] "v"(ty[5]),
          [s0] "v"(ts[0]), [s1] "v"(ts[1]), [s2] "v"(ts[2]), [s3] "v"(ts[3]), [s4] "v"(ts[4]), [s5] "v"(ts[5]),
          [q0] "v"(tq[0]), [q1] "v"(tq[1]), [q2] "v"(tq[2]), [q3] "v"(tq[3]), [q4] "v"(tq[4]), [q5] "v"(tq[5]));
#undef TGS_TAP4
    {   // tap 0 opens the running sums of output row r + 5 (slot (P + 10) % 11, finished and read 11 rows ago): an assignment, no zeroing pass
        const float w = win.w[0];
        const int sl = (P + NTAP - 1) % NTAP;
        V[0][sl] = w * hx; V[1][sl] = w * hy; V[2][sl] = w * hss; V[3][sl] = w * hxy;
    }
#pragma unroll
    for (int k = 1; k < NTAP; k++) {                        // this row is tap k of output row r + 5 - k, whose running sums sit in slot (P + 10 - k) % 11
        const float w = win.w[k];
        const int sl = (P + NTAP - 1 - k) % NTAP;
        V[0][sl] += w * hx; V[1][sl] += w * hy; V[2][sl] += w * hss; V[3][sl] += w * hxy;
    }
    const int ro = r - LR;
    if (ro >= g.y0 && ro < g.y1) {                          // (uniform over the wave)
        const float m1 = V[0][P], m2 = V[1][P], SS = V[2][P], XY = V[3][P];
        // loss_utils.py:49-58
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
        const float s12 = XY - m12;
        const float N1 = 2.f * m12 + C1, N2 = 2.f * s12 + C2, D1 = m11 + m22 + C1, D2 = (SS - m11 - m22) + C2;      // sigma1^2 + sigma2^2 + C2
        const float iD1 = __builtin_amdgcn_rcpf(D1), iD2 = __builtin_amdgcn_rcpf(D2), q = iD1 * iD2;     // v_rcp_f32 (1 ulp); D1, D2 >= C1, C2 > 0
        const float map = N1 * N2 * q;
        if (g.out_ok) {
            const size_t row = g.plane + (size_t)ro * W;          // (the same for every lane)
            at_b(dM1 + row, g.out_b) = 2.f * q * (m2 * (N2 - N1) - m1 * map * (D2 - D1));
            at_b(dX2 + row, g.out_b) = -map * iD2;
            at_b(dXY + row, g.out_b) = 2.f * N1 * q;
            s_map += map;
        }
    }
}

__global__ __launch_bounds__(256) void k_ssim_stats_stream(int H, int W, int nstrips, const float* __restrict__ img, const float* __restrict__ gt, LossWin win,
                                                          float* __restrict__ dM1, float* __restrict__ dX2, float* __restrict__ dXY, float2* __restrict__ partial)
{
    StripGeom g;
    const size_t pidx = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * (gridDim.x * 4) + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (!strip_geom(g, H, W, nstrips)) { if ((threadIdx.x & 63) == 0) partial[pidx] = make_float2(0.f, 0.f); return; }
    float V[4][NTAP];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int k = 0; k < NTAP; k++) V[m][k] = 0.f;
    float s_map = 0.f, s_l1 = 0.f;
    // Loads are UNCONDITIONAL (row and column clamped into the image, the value replaced by the padding's zero afterwards) and the row loop
    // walks whole groups of 11 rows without a test per row: with the load under `if (inside)` -- or the step under `if (r <= r_last)` --
    // the compiler waits with vmcnt(0) behind every pair of loads, i.e. for the row it has just asked for, and the three rows "in flight"
    // were one memory round trip per row.  Rows past the segment's last (at most 10, in the last group) are evaluated and never stored.
    const uint32_t in_b = (uint32_t)min(max(g.col_in, 0), W - 1) * 4u;
    const float* px = img + g.plane;
    const float* py = gt + g.plane;
    // (times 0 or 1, not a select: a select is turned back into a load under a branch, waited for inside it)
    auto ld = [&](const float* p, int r) { return at_b(p + (size_t)min(max(r, 0), H - 1) * W, in_b) * ((g.in_x && r >= 0 && r < H) ? 1.f : 0.f); };
    const int r_first = g.y0 - LR, r_last = g.y1 + LR - 1;  // input rows this segment needs (zero padding outside the image)
    float xa = ld(px, r_first), ya = ld(py, r_first), xb = ld(px, r_first + 1), yb = ld(py, r_first + 1), xc = ld(px, r_first + 2), yc = ld(py, r_first + 2);
#define TGS_STATS_STEP(P)                                                                                                   \
    {                                                                                                                       \
        const int rr = r + P;                                                                                               \
        const float x = xa, y = ya;                                                                                         \
        xa = xb; ya = yb; xb = xc; yb = yc; xc = ld(px, rr + 3); yc = ld(py, rr + 3);   /* three rows in flight */          \
        s_l1 += (g.own_col && rr >= g.y0 && rr < g.y1) ? fabsf(x - y) : 0.f;                                                \
        stats_row<P>(x, y, rr, g, win, V, W, dM1, dX2, dXY, s_map);                                                         \
    }
    for (int r = r_first; r <= r_last; r += NTAP) {
        TGS_STATS_STEP(0) TGS_STATS_STEP(1) TGS_STATS_STEP(2) TGS_STATS_STEP(3) TGS_STATS_STEP(4) TGS_STATS_STEP(5)
        TGS_STATS_STEP(6) TGS_STATS_STEP(7) TGS_STATS_STEP(8) TGS_STATS_STEP(9) TGS_STATS_STEP(10)
    }
#undef TGS_STATS_STEP
    s_map = wave_sum(s_map); s_l1 = wave_sum(s_l1);
    if ((threadIdx.x & 63) == 0) partial[pidx] = make_float2(s_map, s_l1);
}

template <int P>
__device__ __forceinline__ void grad_row(float a, float b, float c, float xo, float yo, int r, const StripGeom& g, const LossWin& win, float (&V)[3][NTAP], int W,
                                         float gs, float gl, float* __restrict__ grad)
{
    // (the sums travel, the six products per map are shared by the symmetric taps: see stats_row)
    float ta[LR + 1], tb[LR + 1], tc[LR + 1];
#pragma unroll
    for (int j = 0; j <= LR; j++) { const float w = win.w[j]; ta[j] = w * a; tb[j] = w * b; tc[j] = w * c; }
    float h0 = ta[0], h1 = tb[0], h2 = tc[0];
#define TGS_TAP3(K) "v_add_f32_dpp %[h0], %[h0], %[a" #K "] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
                    "v_add_f32_dpp %[h1], %[h1], %[b" #K "] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
                    "v_add_f32_dpp %[h2], %[h2], %[c" #K "] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("s_nop 1\n\t" TGS_TAP3(1) TGS_TAP3(2) TGS_TAP3(3) TGS_TAP3(4) TGS_TAP3(5) TGS_TAP3(4) TGS_TAP3(3) TGS_TAP3(2) TGS_TAP3(1) TGS_TAP3(0)
        : [h0] "+&v"(h0), [h1] "+&v"(h1), [h2] "+&v"(h2)
        : [a0] "v"(ta[0]), [a1] "v"(ta[1]), [a2] "v"(ta[2]), [a3] "v"(ta[3]), [a4] "v"(ta[4]), [a5] "v"(ta[5]),
          [b0] "v"(tb[0]), [b1] "v"(tb[1]), [b2] "v"(tb[2]), [b3] "v"(tb[3]), [b4] "v"(tb[4]), [b5] "v"(tb[5]),
          [c0] "v"(tc[0]), [c1] "v"(tc[1]), [c2] "v"(tc[2]), [c3] "v"(tc[3]), [c4] "v"(tc[4]), [c5] "v"(tc[5]));
#undef TGS_TAP3
    {
        const float w = win.w[0];
        const int sl = (P + NTAP - 1) % NTAP;
        V[0][sl] = w * h0; V[1][sl] = w * h1; V[2][sl] = w * h2;
    }
#pragma unroll
    for (int k = 1; k < NTAP; k++) {
        const float w = win.w[k];
        const int sl = (P + NTAP - 1 - k) % NTAP;
        V[0][sl] += w * h0; V[1][sl] += w * h1; V[2][sl] += w * h2;
    }
    const int ro = r - LR;
    if (ro >= g.y0 && ro < g.y1 && g.out_ok) {
        const float d = xo - yo;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);             // torch's abs backward: 0 at 0
        at_b(grad + g.plane + (size_t)ro * W, g.out_b) = gl * sgn + gs * (V[0][P] + 2.f * xo * V[1][P] + yo * V[2][P]);
    }
}

__global__ __launch_bounds__(256) void k_ssim_grad_stream(int H, int W, int nstrips, const float* __restrict__ img, const float* __restrict__ gt, LossWin win,
                                                         const float* __restrict__ dM1, const float* __restrict__ dX2, const float* __restrict__ dXY,
                                                         float gs, float gl, const float* __restrict__ upstream, float* __restrict__ grad)
{
    StripGeom g;
    if (!strip_geom(g, H, W, nstrips)) return;
    if (upstream) { const float u = upstream[0]; gs *= u; gl *= u; }      // d(outer)/d(loss), a device scalar: the chain rule costs no pass of its own
    float V[3][NTAP];
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int k = 0; k < NTAP; k++) V[m][k] = 0.f;
    const uint32_t in_b = (uint32_t)min(max(g.col_in, 0), W - 1) * 4u;
    const float* p0 = dM1 + g.plane;
    const float* p1 = dX2 + g.plane;
    const float* p2 = dXY + g.plane;
    const bool oc = g.col_out >= 0 && g.col_out < W;
    const uint32_t oq_b = oc ? g.out_b : 0u;
    const float* qx = img + g.plane;
    const float* qy = gt + g.plane;
    // (unconditional clamped loads, whole groups of 11 rows: see k_ssim_stats_stream)
    auto ld = [&](const float* p, int r) { return at_b(p + (size_t)min(max(r, 0), H - 1) * W, in_b) * ((g.in_x && r >= 0 && r < H) ? 1.f : 0.f); };
    auto ldo = [&](const float* p, int r) { return at_b(p + (size_t)min(max(r, 0), H - 1) * W, oq_b) * ((oc && r >= 0 && r < H) ? 1.f : 0.f); };       // the image at the OUTPUT pixel of row r - 5
    const int r_first = g.y0 - LR, r_last = g.y1 + LR - 1;
    float a0 = ld(p0, r_first), b0 = ld(p1, r_first), c0 = ld(p2, r_first), x0 = ldo(qx, r_first - LR), y0v = ldo(qy, r_first - LR);
    float a1 = ld(p0, r_first + 1), b1 = ld(p1, r_first + 1), c1 = ld(p2, r_first + 1), x1 = ldo(qx, r_first + 1 - LR), y1v = ldo(qy, r_first + 1 - LR);
#define TGS_GRAD_STEP(P)                                                                                                    \
    {                                                                                                                       \
        const int rr = r + P;                                                                                               \
        const float a = a0, b = b0, c = c0, xo = x0, yo = y0v;                                                              \
        a0 = a1; b0 = b1; c0 = c1; x0 = x1; y0v = y1v;                                                                      \
        a1 = ld(p0, rr + 2); b1 = ld(p1, rr + 2); c1 = ld(p2, rr + 2); x1 = ldo(qx, rr + 2 - LR); y1v = ldo(qy, rr + 2 - LR); \
        grad_row<P>(a, b, c, xo, yo, rr, g, win, V, W, gs, gl, grad);                                                        \
    }
    for (int r = r_first; r <= r_last; r += NTAP) {
        TGS_GRAD_STEP(0) TGS_GRAD_STEP(1) TGS_GRAD_STEP(2) TGS_GRAD_STEP(3) TGS_GRAD_STEP(4) TGS_GRAD_STEP(5)
        TGS_GRAD_STEP(6) TGS_GRAD_STEP(7) TGS_GRAD_STEP(8) TGS_GRAD_STEP(9) TGS_GRAD_STEP(10)
    }
#undef TGS_GRAD_STEP
}

static dim3 stream_grid(int planes, int height, int width, int& nstrips)
{
    nstrips = (width + SS_OUT - 1) / SS_OUT;
    return dim3((unsigned)((nstrips + 3) / 4), (unsigned)((height + SS_SEG - 1) / SS_SEG), (unsigned)planes);
}

static LossWin make_window()
{
    // loss_utils.py:23-25: exp in double, stored fp32, normalised in fp32
    LossWin w;
    float sum = 0.f;
    for (int i = 0; i <= 2 * LR; i++) { w.w[i] = (float)std::exp(-(double)((i - LR) * (i - LR)) / (2.0 * 1.5 * 1.5)); sum += w.w[i]; }
    for (int i = 0; i <= 2 * LR; i++) w.w[i] /= sum;
    return w;
}

}  // namespace tgs

extern "C" {
#include "../../include/tgs_raster.h"

size_t tgs_l1_ssim_workspace_bytes(int planes, int height, int width)
{
    if (planes <= 0 || height <= 0 || width <= 0) return 0;
    const size_t n = (size_t)planes * height * width;
    const size_t strips = (size_t)((width + tgs::SS_OUT - 1) / tgs::SS_OUT + 3) / 4 * 4;                       // streaming kernels: one partial per wave
    const size_t waves = (size_t)planes * ((height + tgs::SS_SEG - 1) / tgs::SS_SEG) * strips;
    return 3 * n * sizeof(float) + waves * sizeof(float2) + 1024;
}

int tgs_l1_ssim(void* stream, int planes, int height, int width, const float* img, const float* gt, float dssim_factor, float* out3,
                float* dL_dimg, void* workspace, size_t workspace_bytes)
{
    using namespace tgs;
    hipStream_t st = (hipStream_t)stream;
    if (planes <= 0 || height <= 0 || width <= 0 || !img || !gt || !out3 || !workspace)
        return set_error(TGS_ERR_INVALID, "tgs_l1_ssim: positive sizes and non-NULL img / gt / out3 / workspace required");
    if (workspace_bytes < tgs_l1_ssim_workspace_bytes(planes, height, width))
        return set_error(TGS_ERR_INVALID, "tgs_l1_ssim: workspace smaller than tgs_l1_ssim_workspace_bytes()");
    const size_t n = (size_t)planes * height * width;
    if (planes > 65535) return set_error(TGS_ERR_INVALID, "tgs_l1_ssim: image too large");
    float* dM1 = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* dX2 = dM1 + n;
    float* dXY = dX2 + n;
    float2* partial = (float2*)(dXY + n);
    const LossWin win = make_window();
    int nstrips = 0;
    const dim3 sgrid = stream_grid(planes, height, width, nstrips);
    if (sgrid.y > 65535u) return set_error(TGS_ERR_INVALID, "tgs_l1_ssim: image too large");
    const int nblk = (int)(sgrid.x * 4 * sgrid.y * sgrid.z);
    hipLaunchKernelGGL(k_ssim_stats_stream, sgrid, dim3(256), 0, st, height, width, nstrips, img, gt, win, dM1, dX2, dXY, partial);
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(1024), 0, st, nblk, partial, 1.0 / (double)n, dssim_factor, out3);
    if (dL_dimg) {
        const float gs = (float)(-(double)dssim_factor / (double)n), gl = (float)((1.0 - (double)dssim_factor) / (double)n);
        hipLaunchKernelGGL(k_ssim_grad_stream, sgrid, dim3(256), 0, st, height, width, nstrips, img, gt, win, dM1, dX2, dXY, gs, gl, (const float*)nullptr, dL_dimg);
    }
    return hip_status("tgs_l1_ssim");
}

int tgs_l1_ssim_backward(void* stream, int planes, int height, int width, const float* img, const float* gt, float dssim_factor,
                         const float* upstream, float* dL_dimg, const void* workspace, size_t workspace_bytes)
{
    using namespace tgs;
    hipStream_t st = (hipStream_t)stream;
    if (planes <= 0 || height <= 0 || width <= 0 || !img || !gt || !dL_dimg || !workspace)
        return set_error(TGS_ERR_INVALID, "tgs_l1_ssim_backward: positive sizes and non-NULL img / gt / dL_dimg / workspace required");
    if (workspace_bytes < tgs_l1_ssim_workspace_bytes(planes, height, width))
        return set_error(TGS_ERR_INVALID, "tgs_l1_ssim_backward: workspace smaller than tgs_l1_ssim_workspace_bytes()");
    const size_t n = (size_t)planes * height * width;
    if (planes > 65535) return set_error(TGS_ERR_INVALID, "tgs_l1_ssim_backward: image too large");
    const float* dM1 = (const float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const float* dX2 = dM1 + n;
    const float* dXY = dX2 + n;
    const LossWin win = make_window();
    const float gs = (float)(-(double)dssim_factor / (double)n), gl = (float)((1.0 - (double)dssim_factor) / (double)n);
    int nstrips = 0;
    const dim3 sgrid = stream_grid(planes, height, width, nstrips);
    if (sgrid.y > 65535u) return set_error(TGS_ERR_INVALID, "tgs_l1_ssim_backward: image too large");
    hipLaunchKernelGGL(k_ssim_grad_stream, sgrid, dim3(256), 0, st, height, width, nstrips, img, gt, win, dM1, dX2, dXY, gs, gl, upstream, dL_dimg);
    return hip_status("tgs_l1_ssim_backward");
}
}
