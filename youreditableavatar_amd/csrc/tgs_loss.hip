// tgs_loss.hip -- the trainers' photometric loss, value and gradient, in two passes over the image ("next" row 2).
//
//   loss = (1 - f) * mean|pred - gt| + f * (1 - mean(ssim_map(pred, gt)))          f = dssim_factor
// (Edit_core/utils/loss_utils.py:17-18, :39-63 composed as in tetgs_texture/refine.py:245-247).  The reference builds
// the SSIM statistics with five zero-padded depthwise 11x11 convolutions plus ~15 element-wise kernels and lets
// autograd run them backwards; here
//   k_ssim_stats : a workgroup walks 4 vertically adjacent 32x32 tiles of one image plane (the next tile's loads in flight under
//                  the current tile's arithmetic); pred and gt tiles (+5 halo, zeros outside) in LDS,
//                  separable 11-tap Gaussian of (x, y, x^2, y^2, xy) -> ssim_map, its three partial derivatives with
//                  respect to the windowed statistics, and the workgroup's partial sums of ssim_map and |x - y|;
//   k_loss_reduce: fixed-order sum of the partials -> loss, ssim, l1 (no float atomics: reproducible);
//   k_ssim_grad  : the adjoint of the same separable window applied to the three derivative maps
//                  d loss / d x(p) = gl * sign(x - y) + gs * (conv(dM1)(p) + 2 x(p) conv(dX2)(p) + y(p) conv(dXY)(p)).
// Both image passes are bound by HBM: 2 planes read + 3 written, then 5 read + 1 written (4 B each).
#include "tgs_device.hpp"
#include <cmath>
#include <cstdlib>

namespace tgs {

constexpr int LW = 32, LH = 32, LR = 5;                 // tile width / height, window radius
constexpr int LTW = LW + 2 * LR, LTH = LH + 2 * LR;     // 42 x 42 with halo
constexpr int NTAP = 2 * LR + 1;
constexpr int STRIP = 4;                                // outputs per thread and pass: 14 inputs serve 4 outputs

struct LossWin { float w[NTAP]; };

// global [planes, H, W] plane -> LDS tile with halo, zeros outside the image (conv2d padding, loss_utils.py:46), in two
// steps so that the loads of the NEXT tile are in flight while the current one is convolved: fetch into registers
// (7 values per thread and map), commit to LDS one iteration later.
constexpr int LPT = (LTW * LTH + 255) / 256;            // tile elements per thread
// The (row, column) of a thread's LPT tile elements never change while a workgroup walks down its column of tiles, so the
// index arithmetic (a division by 42 per element) is done once: global offset inside the tile's rows, LDS offset, and whether
// the element exists / lies inside the image horizontally.
struct TileSlots {
    int goff[LPT];      // r * W + (x0 + c - LR), or -1 when there is no element or it is outside the image in x
    int row[LPT];       // r
    int lds[LPT];       // r * (LTW + 1) + c, or -1 when there is no element
};
__device__ __forceinline__ TileSlots make_slots(int W, int x0)
{
    TileSlots t;
#pragma unroll
    for (int k = 0; k < LPT; k++) {
        const int i = threadIdx.x + 256 * k, r = i / LTW, c = i - r * LTW, gx = x0 + c - LR;
        const bool exists = i < LTW * LTH;
        t.row[k] = r;
        t.lds[k] = exists ? r * (LTW + 1) + c : -1;
        t.goff[k] = (exists && gx >= 0 && gx < W) ? r * W + gx : -1;
    }
    return t;
}
__device__ __forceinline__ void fetch_tile(float (&v)[LPT], const TileSlots& t, const float* __restrict__ src, int H, int W, int y0)
{
    const float* base = src + (ptrdiff_t)(y0 - LR) * W;
#pragma unroll
    for (int k = 0; k < LPT; k++) {
        const int gy = y0 + t.row[k] - LR;
        v[k] = (t.goff[k] >= 0 && gy >= 0 && gy < H) ? base[t.goff[k]] : 0.f;
    }
}
__device__ __forceinline__ void commit_tile(float (*dst)[LTW + 1], const TileSlots& t, const float (&v)[LPT])
{
    float* d = &dst[0][0];
#pragma unroll
    for (int k = 0; k < LPT; k++)
        if (t.lds[k] >= 0) d[t.lds[k]] = v[k];
}
constexpr int LTY = 4;                                   // tiles a workgroup walks down a column of the image

// Horizontal pass of NMAP maps: work item = (row, strip of 4 columns); the 14 inputs of a strip are read once.
// `load(r, c, v)` fills v[NMAP] with the map values at tile position (r, c).
template <int NMAP, typename Load>
__device__ __forceinline__ void hpass(float (*h)[LTH][LW + 1], const LossWin& win, Load load)
{
    for (int i = threadIdx.x; i < LTH * (LW / STRIP); i += 256) {
        const int r = i / (LW / STRIP), c0 = (i % (LW / STRIP)) * STRIP;
        float acc[STRIP][NMAP];
#pragma unroll
        for (int o = 0; o < STRIP; o++)
#pragma unroll
            for (int m = 0; m < NMAP; m++) acc[o][m] = 0.f;
#pragma unroll
        for (int k = 0; k < NTAP + STRIP - 1; k++) {
            float v[NMAP];
            load(r, c0 + k, v);
#pragma unroll
            for (int o = 0; o < STRIP; o++) {
                if (k - o >= 0 && k - o < NTAP) {
                    const float w = win.w[k - o];
#pragma unroll
                    for (int m = 0; m < NMAP; m++) acc[o][m] += w * v[m];
                }
            }
        }
#pragma unroll
        for (int o = 0; o < STRIP; o++)
#pragma unroll
            for (int m = 0; m < NMAP; m++) h[m][r][c0 + o] = acc[o][m];
    }
}
// Vertical pass: thread (tx, ty) produces rows 4 ty .. 4 ty + 3 of column tx from 14 rows of h
template <int NMAP>
__device__ __forceinline__ void vpass(const float (*h)[LTH][LW + 1], const LossWin& win, int tx, int ty, float (&out)[STRIP][NMAP])
{
#pragma unroll
    for (int o = 0; o < STRIP; o++)
#pragma unroll
        for (int m = 0; m < NMAP; m++) out[o][m] = 0.f;
#pragma unroll
    for (int k = 0; k < NTAP + STRIP - 1; k++) {
        float v[NMAP];
#pragma unroll
        for (int m = 0; m < NMAP; m++) v[m] = h[m][STRIP * ty + k][tx];
#pragma unroll
        for (int o = 0; o < STRIP; o++) {
            if (k - o >= 0 && k - o < NTAP) {
                const float w = win.w[k - o];
#pragma unroll
                for (int m = 0; m < NMAP; m++) out[o][m] += w * v[m];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_ssim_stats(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, LossWin win,
                                                   float* __restrict__ dM1, float* __restrict__ dX2, float* __restrict__ dXY, float2* __restrict__ partial)
{
    __shared__ float sx[LTH][LTW + 1], sy[LTH][LTW + 1];
    __shared__ float h[5][LTH][LW + 1];
    __shared__ float2 red[4];
    const int x0 = blockIdx.x * LW;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float s_map = 0.f, s_l1 = 0.f;
    float vx[LPT], vy[LPT];
    const TileSlots slots = make_slots(W, x0);
    fetch_tile(vx, slots, img + plane, H, W, blockIdx.y * LTY * LH);
    fetch_tile(vy, slots, gt + plane, H, W, blockIdx.y * LTY * LH);
    for (int it = 0; it < LTY; it++) {
        const int y0 = (blockIdx.y * LTY + it) * LH;
        if (y0 >= H) break;                                  // uniform
        commit_tile(sx, slots, vx);
        commit_tile(sy, slots, vy);
        __syncthreads();
        if (it + 1 < LTY && y0 + LH < H) {                   // next tile's loads fly under this tile's arithmetic
            fetch_tile(vx, slots, img + plane, H, W, y0 + LH);
            fetch_tile(vy, slots, gt + plane, H, W, y0 + LH);
        }
        hpass<5>(h, win, [&](int r, int c, float (&v)[5]) { const float x = sx[r][c], y = sy[r][c]; v[0] = x; v[1] = y; v[2] = x * x; v[3] = y * y; v[4] = x * y; });
        __syncthreads();
        float st[STRIP][5];
        vpass<5>(h, win, tx, ty, st);
#pragma unroll
        for (int o = 0; o < STRIP; o++) {
            const int row = STRIP * ty + o, gx = x0 + tx, gy = y0 + row;
            if (gx < W && gy < H) {
                // loss_utils.py:49-58
                const float m1 = st[o][0], m2 = st[o][1], X2 = st[o][2], Y2 = st[o][3], XY = st[o][4];
                const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
                const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
                const float s1 = X2 - m11, s2 = Y2 - m22, s12 = XY - m12;
                const float N1 = 2.f * m12 + C1, N2 = 2.f * s12 + C2, D1 = m11 + m22 + C1, D2 = s1 + s2 + C2;
                const float iD1 = 1.f / D1, iD2 = 1.f / D2, q = iD1 * iD2;
                const float map = N1 * N2 * q;
                // map as a function of the windowed statistics (m1, X2, XY); sigma1_sq = X2 - m1^2, sigma12 = XY - m1 m2
                const size_t at = plane + (size_t)gy * W + gx;
                dM1[at] = 2.f * q * (m2 * (N2 - N1) - m1 * map * (D2 - D1));
                dX2[at] = -map * iD2;
                dXY[at] = 2.f * N1 * q;
                s_map += map;
                s_l1 += fabsf(sx[row + LR][tx + LR] - sy[row + LR][tx + LR]);
            }
        }
        __syncthreads();                                     // sx / sy / h are rewritten by the next tile
    }
    s_map = wave_sum(s_map); s_l1 = wave_sum(s_l1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(s_map, s_l1);
    __syncthreads();
    if (threadIdx.x == 0)
        partial[(size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] =
            make_float2((red[0].x + red[1].x) + (red[2].x + red[3].x), (red[0].y + red[1].y) + (red[2].y + red[3].y));
}

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

__global__ __launch_bounds__(256) void k_ssim_grad(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, LossWin win,
                                                  const float* __restrict__ dM1, const float* __restrict__ dX2, const float* __restrict__ dXY,
                                                  float gs, float gl, const float* __restrict__ upstream, float* __restrict__ grad)
{
    __shared__ float t[3][LTH][LTW + 1];
    __shared__ float h[3][LTH][LW + 1];
    if (upstream) { const float u = upstream[0]; gs *= u; gl *= u; }      // d(outer)/d(loss), a device scalar: the chain rule costs no pass of its own
    const int x0 = blockIdx.x * LW;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float v0[LPT], v1[LPT], v2[LPT];
    const TileSlots slots = make_slots(W, x0);
    fetch_tile(v0, slots, dM1 + plane, H, W, blockIdx.y * LTY * LH);
    fetch_tile(v1, slots, dX2 + plane, H, W, blockIdx.y * LTY * LH);
    fetch_tile(v2, slots, dXY + plane, H, W, blockIdx.y * LTY * LH);
    for (int it = 0; it < LTY; it++) {
        const int y0 = (blockIdx.y * LTY + it) * LH;
        if (y0 >= H) break;                                  // uniform
        commit_tile(t[0], slots, v0);
        commit_tile(t[1], slots, v1);
        commit_tile(t[2], slots, v2);
        __syncthreads();
        if (it + 1 < LTY && y0 + LH < H) {
            fetch_tile(v0, slots, dM1 + plane, H, W, y0 + LH);
            fetch_tile(v1, slots, dX2 + plane, H, W, y0 + LH);
            fetch_tile(v2, slots, dXY + plane, H, W, y0 + LH);
        }
        hpass<3>(h, win, [&](int r, int c, float (&v)[3]) { v[0] = t[0][r][c]; v[1] = t[1][r][c]; v[2] = t[2][r][c]; });
        __syncthreads();
        float cv[STRIP][3];
        vpass<3>(h, win, tx, ty, cv);
#pragma unroll
        for (int o = 0; o < STRIP; o++) {
            const int gx = x0 + tx, gy = y0 + STRIP * ty + o;
            if (gx >= W || gy >= H) continue;
            const size_t at = plane + (size_t)gy * W + gx;
            const float x = img[at], y = gt[at], d = x - y;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);         // torch's abs backward: 0 at 0
            grad[at] = gl * sgn + gs * (cv[o][0] + 2.f * x * cv[o][1] + y * cv[o][2]);
        }
        __syncthreads();
    }
}

// =====================================================================================================================================
// Round 3: the same two passes as STREAMING kernels without LDS and without barriers.  A wave owns a vertical strip of the plane -- lane =
// column, 64 input columns = 54 output columns + the window's 5-column halo on either side -- and walks down SS_SEG output rows:
//   per input row: one coalesced 4-B load per lane and map; the horizontal 11-tap window from the neighbouring lanes by ten full-wave DPP
//   shifts (wave_shr:1) per input map; the vertical window as 11 running sums per lane and map (a row adds w[k] * H to the 11 output rows
//   it belongs to; the row loop is unrolled by 11 so that the slot of a running sum is a compile-time register);
//   the output row 5 rows behind is finished: SSIM map and its derivatives (stats pass) / the gradient (gradient pass), one store per map.
// No tile staging, no shared memory, no workgroup barrier: the tiled kernels above spent their time waiting on exactly those (3 workgroups
// per CU at 42 KB of LDS each, three barriers per tile; 1.9 / 2.7 TB/s of their algorithmic bytes at 3 x 2048 x 2048).  Cost of the scheme:
// 64 / 54 of the input loads (neighbouring strips overlap by 10 columns, L2 hits) and 10 warm-up rows per segment.
// The tiled kernels remain for A/B runs (TGS_LOSS_TILED=1).
// Round 4: the row loads are unconditional (clamped into the image, times 0 or 1) and the row loop walks whole groups of 11 rows: under `if`
// each pair of loads was waited for with vmcnt(0) -- one memory round trip per row instead of three rows in flight: 104 + 78 -> 96.5 + 76 us.
// Measured and not kept (round 4): TWO columns per lane (one wave shift per map and tap instead of two, halo 10 of 128 columns instead of 10 of
// 64; bit-equal): 125 / 106 VGPRs instead of 75 / 68, half as many waves -- 113 + 95 us instead of 104 + 78 at 3 x 2048 x 2048.
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

struct StripGeom { int col_in, col_out, y0, y1; bool in_x, out_ok, own_col; size_t plane; };
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
    float hx = 0.f, hy = 0.f, hss = 0.f, hxy = 0.f;
    float xs = x, ys = y;
#pragma unroll
    for (int j = 0; j < NTAP; j++) {                        // lane L holds column c; after j shifts xs = x(c - j): the window of output column c - 5
        const float w = win.w[j];
        const float wx = w * xs, wy = w * ys;
        hx += wx; hy += wy; hss += wx * xs; hss += wy * ys; hxy += wx * ys;
        if (j + 1 < NTAP) { xs = wave_shr1(xs); ys = wave_shr1(ys); }
    }
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
            const size_t at = g.plane + (size_t)ro * W + g.col_out;
            dM1[at] = 2.f * q * (m2 * (N2 - N1) - m1 * map * (D2 - D1));
            dX2[at] = -map * iD2;
            dXY[at] = 2.f * N1 * q;
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
    const int colc = min(max(g.col_in, 0), W - 1);
    const float* px = img + g.plane + colc;
    const float* py = gt + g.plane + colc;
    // (times 0 or 1, not a select: a select is turned back into a load under a branch, waited for inside it)
    auto ld = [&](const float* p, int r) { return p[(size_t)min(max(r, 0), H - 1) * W] * ((g.in_x && r >= 0 && r < H) ? 1.f : 0.f); };
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
    float h0 = 0.f, h1 = 0.f, h2 = 0.f;
    float as = a, bs = b, cs = c;
#pragma unroll
    for (int j = 0; j < NTAP; j++) {
        const float w = win.w[j];
        h0 += w * as; h1 += w * bs; h2 += w * cs;
        if (j + 1 < NTAP) { as = wave_shr1(as); bs = wave_shr1(bs); cs = wave_shr1(cs); }
    }
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
        grad[g.plane + (size_t)ro * W + g.col_out] = gl * sgn + gs * (V[0][P] + 2.f * xo * V[1][P] + yo * V[2][P]);
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
    const int colc = min(max(g.col_in, 0), W - 1);
    const float* p0 = dM1 + g.plane + colc;
    const float* p1 = dX2 + g.plane + colc;
    const float* p2 = dXY + g.plane + colc;
    const bool oc = g.col_out >= 0 && g.col_out < W;
    const float* qx = img + g.plane + (oc ? g.col_out : 0);
    const float* qy = gt + g.plane + (oc ? g.col_out : 0);
    // (unconditional clamped loads, whole groups of 11 rows: see k_ssim_stats_stream)
    auto ld = [&](const float* p, int r) { return p[(size_t)min(max(r, 0), H - 1) * W] * ((g.in_x && r >= 0 && r < H) ? 1.f : 0.f); };
    auto ldo = [&](const float* p, int r) { return p[(size_t)min(max(r, 0), H - 1) * W] * ((oc && r >= 0 && r < H) ? 1.f : 0.f); };       // the image at the OUTPUT pixel of row r - 5
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

static bool loss_tiled()
{
    static const bool v = [] { const char* e = getenv("TGS_LOSS_TILED"); return e && atoi(e) != 0; }();      // A/B knob, read once
    return v;
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
    const size_t blocks = (size_t)planes * ((height + tgs::LH * tgs::LTY - 1) / (tgs::LH * tgs::LTY)) * ((width + tgs::LW - 1) / tgs::LW);
    const size_t strips = (size_t)((width + tgs::SS_OUT - 1) / tgs::SS_OUT + 3) / 4 * 4;                       // streaming kernels: one partial per wave
    const size_t waves = (size_t)planes * ((height + tgs::SS_SEG - 1) / tgs::SS_SEG) * strips;
    return 3 * n * sizeof(float) + (blocks > waves ? blocks : waves) * sizeof(float2) + 1024;
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
    const dim3 grid((width + LW - 1) / LW, (height + LH * LTY - 1) / (LH * LTY), planes);
    if (grid.y > 65535u || grid.z > 65535u) return set_error(TGS_ERR_INVALID, "tgs_l1_ssim: image too large");
    float* dM1 = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* dX2 = dM1 + n;
    float* dXY = dX2 + n;
    float2* partial = (float2*)(dXY + n);
    const LossWin win = make_window();
    int nstrips = 0;
    const dim3 sgrid = stream_grid(planes, height, width, nstrips);
    if (sgrid.y > 65535u) return set_error(TGS_ERR_INVALID, "tgs_l1_ssim: image too large");
    const bool tiled = loss_tiled();
    const int nblk = tiled ? (int)(grid.x * grid.y * grid.z) : (int)(sgrid.x * 4 * sgrid.y * sgrid.z);
    if (tiled) hipLaunchKernelGGL(k_ssim_stats, grid, dim3(256), 0, st, height, width, img, gt, win, dM1, dX2, dXY, partial);
    else hipLaunchKernelGGL(k_ssim_stats_stream, sgrid, dim3(256), 0, st, height, width, nstrips, img, gt, win, dM1, dX2, dXY, partial);
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(1024), 0, st, nblk, partial, 1.0 / (double)n, dssim_factor, out3);
    if (dL_dimg) {
        const float gs = (float)(-(double)dssim_factor / (double)n), gl = (float)((1.0 - (double)dssim_factor) / (double)n);
        if (tiled) hipLaunchKernelGGL(k_ssim_grad, grid, dim3(256), 0, st, height, width, img, gt, win, dM1, dX2, dXY, gs, gl, (const float*)nullptr, dL_dimg);
        else hipLaunchKernelGGL(k_ssim_grad_stream, sgrid, dim3(256), 0, st, height, width, nstrips, img, gt, win, dM1, dX2, dXY, gs, gl, (const float*)nullptr, dL_dimg);
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
    const dim3 grid((width + LW - 1) / LW, (height + LH * LTY - 1) / (LH * LTY), planes);
    if (grid.y > 65535u || grid.z > 65535u) return set_error(TGS_ERR_INVALID, "tgs_l1_ssim_backward: image too large");
    const float* dM1 = (const float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const float* dX2 = dM1 + n;
    const float* dXY = dX2 + n;
    const LossWin win = make_window();
    const float gs = (float)(-(double)dssim_factor / (double)n), gl = (float)((1.0 - (double)dssim_factor) / (double)n);
    int nstrips = 0;
    const dim3 sgrid = stream_grid(planes, height, width, nstrips);
    if (loss_tiled()) hipLaunchKernelGGL(k_ssim_grad, grid, dim3(256), 0, st, height, width, img, gt, win, dM1, dX2, dXY, gs, gl, upstream, dL_dimg);
    else hipLaunchKernelGGL(k_ssim_grad_stream, sgrid, dim3(256), 0, st, height, width, nstrips, img, gt, win, dM1, dX2, dXY, gs, gl, upstream, dL_dimg);
    return hip_status("tgs_l1_ssim_backward");
}
}
