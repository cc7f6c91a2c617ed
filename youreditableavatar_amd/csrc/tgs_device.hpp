// tgs_device.hpp -- shared device helpers and the private layout of the three state buffers.
// gfx950 (CDNA4) only: wave64, DPP row operations, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stddef.h>
#include <stdint.h>

namespace tgs {

constexpr int TILE = 16;            // the reference's BLOCK_X = BLOCK_Y (cuda_rasterizer/config.h:16-17)
constexpr int TILE_PIX = TILE * TILE;
constexpr int WAVE = 64;
constexpr int PRE_BLOCK = 256;      // Gaussians per workgroup in the per-Gaussian kernels
constexpr uint32_t SORT_LDS_CAP = 8192;   // longest tile list sorted inside LDS (64 KiB of u64 keys)
constexpr int COOP_TILES = 64;       // rectangles up to this size carry a 64-bit mask of their live tiles; their (splat, tile) pairs are spread over a wave
constexpr int RANK_TILES = 4;        // rectangles up to this size (almost all) are binned by the splat's own lane
constexpr int SLAB_ROW = 3;          // float4 per gradient-slab row: 9 sums padded to 48 B so rows move as three 16-B accesses
// Binning (k_bin_count / k_bin_colscan / k_scatter): the Gaussians are cut into BIN_WGS_MAX (or fewer) contiguous chunks, one fat
// workgroup each, which count and later emit their instances through a per-tile table in LDS -- no global atomics.
#ifndef TGS_BIN_WGS_MAX
#define TGS_BIN_WGS_MAX 256
#endif
constexpr int BIN_WGS_MAX = TGS_BIN_WGS_MAX;   // rows of the per-view table (bin_table: BIN_WGS_MAX x T words); TGS_BIN_WGS (environment) selects fewer
constexpr int BIN_THREADS = 1024;
constexpr uint32_t BIN_LDS_TILES = 24576; // tiles per pass of the LDS table (96 KB); larger tile grids are walked in bands

// ---------------------------------------------------------------------------------------------
// state buffers (opaque to callers; the reference's equivalents: rasterizer_impl.h:29-65)
// ---------------------------------------------------------------------------------------------
struct Meta {                 // lives at the start of the image buffer
    unsigned long long R;     // number of tile instances ("num_rendered")
    uint32_t max_count;       // longest tile list
    uint32_t n_overflow;      // tiles whose list is longer than SORT_LDS_CAP
    uint32_t error;           // bit0: prefiltered Gaussian culled; bit1 (META_ERR_CAPACITY): frame rejected by tgs_forward_async; bit2: META_ERR_TILE_BOUND
    uint32_t n_nonempty;      // tiles with at least one instance (they come first in tile_order)
    uint32_t n_heavy;         // tiles with >= 1024 instances (first in tile_order): sorted by 1024-thread workgroups
    uint32_t n_mid;           // tiles with >= 128 instances (heavy ones included); the rest are sorted one wave per tile
    uint32_t pad[8];          // [0]: a tile bound was exceeded (k_scan, mirrored to host_meta)
};

constexpr uint32_t META_ERR_CAPACITY = 2u;
constexpr uint32_t META_ERR_TILE_BOUND = 4u;   // a per-pixel backward was launched over fewer tiles than hold instances (caller's tile bound too small): TGS_FRAME_TILE_BOUND

// scratch of the binning chain's scan (round 5), cleared with Meta by block 0 of the per-Gaussian forward stage
struct ScanAux {
    uint32_t hist[40];        // tiles per list-length bucket (0: empty, b: 2^(b-1) <= n < 2^b), added up by k_bin_colscan's workgroups
    uint32_t cursor[40];      // k_scan: positions handed out inside each bucket's stretch of tile_order
    uint32_t done;            // k_scan: tile workgroups that have finished (the last one copies the frame's Meta to the host)
    uint32_t pad[47];
};
static_assert(sizeof(ScanAux) == 512, "ScanAux: 128 words, cleared by the first 128 threads of a block");

struct GeomState {
    // One 64-byte line per Gaussian with everything the per-tile gather needs (geomState.means2D,
    // .conic_opacity, .rgb of the reference, plus the tile rectangle and the slab offset):
    //   pack[4g+0] = (x, y, conic.x, conic.y)   pack[4g+1] = (conic.z, opacity, r, g)
    //   pack[4g+2] = (b, bits(minx | miny<<16), bits(maxx | maxy<<16), bits(offset))
    //   pack[4g+3] = for a rectangle of <= COOP_TILES tiles the 64-bit mask of its LIVE tiles, row-major (.x low word, .y high word):
    //                a tile of the rectangle the splat cannot reach with alpha >= 1/255 gets no instance
    float4* pack;
    uint2* live;              // the same mask once more, compact (the binning kernels read 20 B per Gaussian instead of its pack line)
    float* depth;             // view-space z                       (geomState.depths)
    // (geomState.cov3D has no counterpart: the backward evaluates compute_cov3d again)
    uint8_t* clamped;         // 3 clamp bits per Gaussian          (geomState.clamped)
    ushort4* rect;            // tile rectangle (minx, miny, maxx, maxy)
    uint32_t* tiles_touched;  //                                    (geomState.tiles_touched)
    uint32_t* offsets;        // EXCLUSIVE scan of tiles_touched    (geomState.point_offsets is inclusive)
    uint32_t* block_sums;     // per PRE_BLOCK partial sums / their exclusive scan
    uint32_t* block_flags;    // per PRE_BLOCK: 1 if a Gaussian of the block was culled although `prefiltered` is set (auxiliary.h:156-160); every
                              // block of k_preprocess_fwd* writes its own word and k_scan ORs them into Meta::error -- no atomic on Meta
                              // before the scan, so the preprocess kernel itself can clear Meta (no memset launch in front of a frame)
};
struct ImgState {
    Meta* meta;
    ScanAux* aux;             // (behind Meta: both are cleared in front of a frame)
    uint2* ranges;            // per tile [start, end)              (imgState.ranges)
    uint32_t* tile_count;     // per tile: number of instances (k_bin_colscan)
    uint32_t* bin_table;      // [BIN_WGS_MAX][T]: instances of binning chunk w in tile t (k_bin_count), then their first position
                              // inside the tile's list (exclusive scan over w, k_bin_colscan)
    uint32_t* ovf_tiles;      // list of overflow tiles
    uint32_t* tile_order;     // tiles by descending list length: render kernels start the long lists first
    uint4* tile_desc;         // the same order with the range inlined: (tile, start, end, w); w = the deepest list position any pixel of the tile blended (max n_contrib), written by k_render_fwd -- one load instead of a dependent chain
    uint4* light_desc;        // the descriptors of the LIGHT tiles once more (tile_desc[n_mid + i], i < n_nonempty - n_mid: fewer than LIGHT_MAX instances),
                              // indexed from 0: the light render kernels fetch descriptor and frame counts side by side instead of one behind the other
    float* final_T;           //                                    (imgState.accum_alpha)
    uint32_t* n_contrib;      //                                    (imgState.n_contrib)
    unsigned long long* stamps; // diagnostic builds (-DTGS_STAMPS=1): per tile {fwd start, fwd end, bwd start, bwd end, fwd sum / max of the waves' busy time, bwd sum / max}, 100 MHz ticks
};
struct BinState {
    unsigned long long* keys; // (depth bits << 32 | gaussian idx), per tile segment, sorted after k_tile_sort
    uint32_t* tile_of;        // tile of list position p (written by k_tile_sort with the sorted keys: k_finalize is one thread per position)
    float4* recA;             // sorted per-instance records: xy.x, xy.y, conic.x, conic.y
    float4* recB;             //                              conic.z, opacity, r, g
    float2* recC;             //                              b, bits(gaussian idx)
    uint32_t* slot;           // offsets[g] + ordinal of this tile in g's rectangle (gradient slab row)
    uint2* qmask;             // 64-bit mask of the tile's 8x8 grid of 2x2-pixel quadrants the splat reaches: bit 8*row + column
    float4* slab;             // backward scratch: SLAB_ROW float4 (9 sums + padding) per instance, Gaussian-major rows
};

template <typename T>
__host__ __device__ inline void carve(char*& p, T*& out, size_t count)
{
    uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255);
    out = reinterpret_cast<T*>(a);
    p = reinterpret_cast<char*>(out + count);
}
__host__ __device__ inline size_t n_blocks(size_t P) { return (P + PRE_BLOCK - 1) / PRE_BLOCK; }

__host__ __device__ inline size_t geom_carve(GeomState& g, char* base, size_t P, bool has_sh, bool has_scale_rot)
{
    char* p = base;
    (void)has_sh; (void)has_scale_rot;
    carve(p, g.pack, 4 * P); carve(p, g.live, P); carve(p, g.depth, P);
    carve(p, g.clamped, P); carve(p, g.rect, P); carve(p, g.tiles_touched, P); carve(p, g.offsets, P);
    carve(p, g.block_sums, n_blocks(P) + 1); carve(p, g.block_flags, n_blocks(P) + 1);
    return (size_t)(p - base) + 256;
}
__host__ __device__ inline size_t img_carve(ImgState& s, char* base, size_t N, size_t T)
{
    char* p = base;
    carve(p, s.meta, 1); carve(p, s.aux, 1); carve(p, s.ranges, T); carve(p, s.tile_count, T); carve(p, s.bin_table, (size_t)BIN_WGS_MAX * T);
    carve(p, s.ovf_tiles, T); carve(p, s.tile_order, T); carve(p, s.tile_desc, T); carve(p, s.light_desc, T);
    carve(p, s.final_T, N); carve(p, s.n_contrib, N);
    carve(p, s.stamps, 8 * T);
    return (size_t)(p - base) + 256;
}
__host__ __device__ inline size_t bin_carve(BinState& b, char* base, size_t R)
{
    char* p = base;
    carve(p, b.keys, R); carve(p, b.tile_of, R); carve(p, b.recA, R); carve(p, b.recB, R); carve(p, b.recC, R); carve(p, b.slot, R); carve(p, b.qmask, R); carve(p, b.slab, (size_t)SLAB_ROW * R);
    return (size_t)(p - base) + 256;
}

// host side (tgs_api.hip): record the message tgs_last_error() returns
int set_error(int code, const char* msg);
int hip_status(const char* what);            // hipGetLastError() -> TGS_OK / TGS_ERR_HIP (+ message)

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------
// wave64 primitives
// ---------------------------------------------------------------------------------------------
#define TGS_DPP_ADD(v, ctrl, rowmask)                                                                      \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rowmask, 0xf, false))

// Sum over the 64 lanes of a wave; the total is returned in every lane (via readlane 63).
// 6 v_add_f32 with DPP modifiers: quad swaps, row_half_mirror, row_mirror, row_bcast15, row_bcast31.
__device__ __forceinline__ float wave_sum(float v)
{
    TGS_DPP_ADD(v, 0xB1, 0xf);    // quad_perm [1,0,3,2]
    TGS_DPP_ADD(v, 0x4E, 0xf);    // quad_perm [2,3,0,1]
    TGS_DPP_ADD(v, 0x141, 0xf);   // row_half_mirror
    TGS_DPP_ADD(v, 0x140, 0xf);   // row_mirror
    TGS_DPP_ADD(v, 0x142, 0xa);   // row_bcast15 -> rows 1,3
    TGS_DPP_ADD(v, 0x143, 0xc);   // row_bcast31 -> rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// max over the wave, returned in every lane: 6 DPP steps like wave_sum (a __shfl_xor butterfly is 6 ds_bpermute round trips through LDS)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#define TGS_DPP_MAX(ctrl, rowmask) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rowmask, 0xf, false); v = o_ > v ? o_ : v; }
    TGS_DPP_MAX(0xB1, 0xf)     // quad_perm [1,0,3,2]
    TGS_DPP_MAX(0x4E, 0xf)     // quad_perm [2,3,0,1]
    TGS_DPP_MAX(0x141, 0xf)    // row_half_mirror
    TGS_DPP_MAX(0x140, 0xf)    // row_mirror
    TGS_DPP_MAX(0x142, 0xa)    // row_bcast15 -> rows 1,3
    TGS_DPP_MAX(0x143, 0xc)    // row_bcast31 -> rows 2,3
#undef TGS_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// inclusive scan over the wave
__device__ __forceinline__ uint32_t wave_iscan_u32(uint32_t v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    return v;
}
__device__ __forceinline__ unsigned long long wave_iscan_u64(unsigned long long v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { unsigned long long t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    return v;
}

// ---------------------------------------------------------------------------------------------
// 3x3 helpers in GLM's column-major convention, m[c][r] = column c, row r, product order of
// third_party/glm/glm/detail/type_mat3x3.inl:486-518 -- so that the fp32 operation order of the
// reference's glm expressions is kept.
// ---------------------------------------------------------------------------------------------
struct mat3 { float m[3][3]; };
__device__ __forceinline__ mat3 m3mul(const mat3& a, const mat3& b)
{
#pragma clang fp contract(off)      // un-fused like the oracle (and the same in every kernel this is inlined into)
    mat3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int w = 0; w < 3; w++)
            r.m[c][w] = a.m[0][w] * b.m[c][0] + a.m[1][w] * b.m[c][1] + a.m[2][w] * b.m[c][2];
    return r;
}
__device__ __forceinline__ mat3 m3t(const mat3& a)
{
    mat3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int w = 0; w < 3; w++) r.m[c][w] = a.m[w][c];
    return r;
}
__device__ __forceinline__ mat3 m3make(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    mat3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c; r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}

// ---------------------------------------------------------------------------------------------
// render-kernel staging shared by forward and backward
// ---------------------------------------------------------------------------------------------
constexpr int RCHUNK = 256;            // list entries staged per round (one per thread)
constexpr int RNULL = RCHUNK;          // LDS slot of a null record (opacity 0: never contributes)
constexpr int RUNROLL = 4;             // entries evaluated together by a wave
constexpr int QL_STRIDE = 64 + RUNROLL;

// Per round: for every pixel quadrant (= compute wave) and every staging wave, the compacted list of
// staged slots whose splat can reach the quadrant (quadrant_mask() bit), padded with RNULL to a
// multiple of RUNROLL.  A compute wave walks its 4 sub-lists with plain LDS reads -- no scalar
// bit-twiddling in the hot loop.
struct alignas(16) QuadLists {       // rows are read 8 bytes at a time: QL_STRIDE*2 is a multiple of 8
    unsigned short idx[4][4][QL_STRIDE];   // [quadrant][staging wave][k]
    uint32_t cnt[4][4];
};
__device__ __forceinline__ void build_quad_lists(QuadLists& L, uint32_t qm, int wv, int lane)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const bool on = (qm >> q) & 1u;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(on);
        const uint32_t n = (uint32_t)__builtin_popcountll(bal);
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (on) L.idx[q][wv][pos] = (unsigned short)threadIdx.x;
        if (lane < RUNROLL) L.idx[q][wv][n + lane] = (unsigned short)RNULL;
        if (lane == 0) L.cnt[q][wv] = n;
    }
}

// ---- quadrant culling of the default render kernels -------------------------------------------------------------------
// A wave owns a 4x4-pixel block; each of its four 16-lane DPP rows owns one 2x2-pixel quadrant of the block (4 pixels x 4
// entry slots) and walks ITS OWN list: the entries whose splat reaches alpha >= 1/255 on one of the quadrant's pixels
// (BinState::qmask).  Two levels: build_own_list_q compacts the staged round into the block's list (entries that reach any
// quadrant, each with its 4-bit quadrant nibble), then every chunk of 64 block entries is split into the four quadrant lists
// (build_chunk_quadrant_lists) and the wave runs max_q ceil(n_q / 4) passes over them.
constexpr int QCH = 64;                    // block-list entries split per chunk
constexpr int QL_ROW = QCH + 8;            // one quadrant list: <= 64 entries + null padding up to the longest list of the chunk

// ---- list building, round 5 -------------------------------------------------------------------------------------------
// Cost decomposition on the MI355X (profiles/r05_render_decomposition.txt): building the lists twice costs k_render_fwd +12.4 us of 56
// and k_render_bwd +8 us of 97 -- a fifth of the forward is ordered compaction.  The compiler's code for `ballot; if (on) list[base +
// mbcnt] = x; base += popcount` was 9-11 vector instructions per list and 64 entries (two compares of the same bit, address
// arithmetic in three steps) plus 5 per list for the null padding; written out it is 5 + 1 LDS write, and the padding is ONE 16-byte
// store per lane in front of the appends (LDS operations of a wave execute in order, so the appends overwrite what they need).
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }   // byte offset of a __shared__ object (low word of its flat address)

// Appends `val` (low 16 bits) of every lane whose `ent` has a bit of BITS set to the u16 list at LDS byte address `base` (wave-uniform),
// in lane order; returns how many (wave-uniform, in an SGPR).  The ranks are only needed by the lanes that write: mbcnt runs under their exec.
// HAZARDS the hand-written blocks rely on (LLVM's hazard recogniser does not look inside inline asm; ADVICE of round 5):
//   * gfx940 family: a VALU write of an SGPR / VCC needs 2 wait states before a VALU instruction reads it as a constant.  v_cmp writes vcc, and
//     v_mbcnt_lo reads vcc_lo: the TWO scalar instructions between them (s_and_saveexec, s_bcnt1) are those wait states -- with no margin.
//     Whoever removes or moves one of them must put an s_nop in its place.  (The scalar unit's own read of vcc is interlocked.)
//   * the 16-byte null fill of ql_fill_null and the b16 appends go to the same LDS words: a wave's LDS operations execute in order (one
//     in-order queue per wave, lgkmcnt counts them in order), so the appends land behind the fill without a wait between them.
// -DTGS_QL_APPEND_C=1 builds the same three functions from ballot / mbcnt builtins (the compiler's 9-11 instructions per list): the A/B partner
// for the parity suite after a toolchain update (TGS_LIB_NAME=libtgs_raster_qlc.so TGS_DEFINES=-DTGS_QL_APPEND_C=1 python -m youreditableavatar_amd.build --force).
#ifndef TGS_QL_APPEND_C
#define TGS_QL_APPEND_C 0
#endif
#if TGS_QL_APPEND_C
template <uint32_t BITS>
__device__ __forceinline__ uint32_t ql_append(uint32_t ent, uint32_t val, uint32_t base)
{
    const bool on = (ent & BITS) != 0u;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(on);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (on) *reinterpret_cast<__attribute__((address_space(3))) unsigned short*>((uintptr_t)(base + 2u * pos)) = (unsigned short)val;
    return (uint32_t)__builtin_popcountll(bal);
}
template <uint32_t BITS>
__device__ __forceinline__ uint32_t ql_append_above(uint32_t ent, uint32_t key, uint32_t val, uint32_t base, int thr)
{
    const bool on = (ent & BITS) != 0u && (int)key > thr;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(on);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (on) *reinterpret_cast<__attribute__((address_space(3))) unsigned short*>((uintptr_t)(base + 2u * pos)) = (unsigned short)val;
    return (uint32_t)__builtin_popcountll(bal);
}
#else
template <uint32_t BITS>
__device__ __forceinline__ uint32_t ql_append(uint32_t ent, uint32_t val, uint32_t base)
{
    uint32_t t, r, n;
    unsigned long long sv;
    asm volatile(
        "v_and_b32 %[t], %[bits], %[ent]\n\t"
        "v_cmp_ne_u32 vcc, 0, %[t]\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "v_mbcnt_lo_u32_b32 %[r], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[r], vcc_hi, %[r]\n\t"
        "v_lshl_add_u32 %[r], %[r], 1, %[base]\n\t"
        "ds_write_b16 %[r], %[val]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [t] "=&v"(t), [r] "=&v"(r), [sv] "=&s"(sv), [n] "=&s"(n)
        : [ent] "v"(ent), [val] "v"(val), [base] "s"(base), [bits] "i"(BITS)
        : "vcc", "scc", "memory");
    return n;
}
// ... and only if (int)key > thr (the backward's bound: list position in front of the quadrant's deepest last contributor)
template <uint32_t BITS>
__device__ __forceinline__ uint32_t ql_append_above(uint32_t ent, uint32_t key, uint32_t val, uint32_t base, int thr)
{
    uint32_t t, r, n;
    unsigned long long sv, m;
    asm volatile(
        "v_and_b32 %[t], %[bits], %[ent]\n\t"
        "v_cmp_ne_u32 vcc, 0, %[t]\n\t"
        "v_cmp_gt_i32 %[m], %[key], %[thr]\n\t"
        "s_and_b64 vcc, vcc, %[m]\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "v_mbcnt_lo_u32_b32 %[r], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[r], vcc_hi, %[r]\n\t"
        "v_lshl_add_u32 %[r], %[r], 1, %[base]\n\t"
        "ds_write_b16 %[r], %[val]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [t] "=&v"(t), [r] "=&v"(r), [sv] "=&s"(sv), [m] "=&s"(m), [n] "=&s"(n)
        : [ent] "v"(ent), [key] "v"(key), [val] "v"(val), [base] "s"(base), [thr] "s"(thr), [bits] "i"(BITS)
        : "vcc", "scc", "memory");
    return n;
}
#endif
// the wave's four quadrant lists <- the null slot everywhere (ROW u16 each, contiguous, 16-B aligned)
template <int ROW>
__device__ __forceinline__ void ql_fill_null(unsigned short (*ql)[ROW], int lane, int null_slot)
{
    static_assert((4 * ROW * 2) % 16 == 0 && 4 * ROW * 2 / 16 <= 128, "quadrant lists: whole 16-byte pieces, at most two per lane");
    constexpr int N16 = 4 * ROW * 2 / 16;
    const uint32_t nn = (uint32_t)null_slot * 0x10001u;
    uint4* p = reinterpret_cast<uint4*>(&ql[0][0]);
    if (N16 >= 64 || lane < N16) p[lane] = make_uint4(nn, nn, nn, nn);
    if (N16 > 64 && lane < N16 - 64) p[64 + lane] = make_uint4(nn, nn, nn, nn);
}

// The block's list of a staged round: entries [0, cnt) whose 64-bit quadrant mask names one of the block's four quadrants, in slot
// order, each as slot | nibble << 10.  CH: capacity of the staged arrays (a multiple of 64: a lane past cnt reads a stale mask and drops it).
template <int CH>
__device__ __forceinline__ uint32_t build_own_list_q(unsigned short* list, const uint2* qmasks, uint32_t cnt_in, int blk_in, int lane)
{
    static_assert(CH % 64 == 0, "whole 64-entry groups");
    const uint32_t blk = (uint32_t)__builtin_amdgcn_readfirstlane(blk_in), cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt_in);
    // block (bx, by) = blk: quadrant rows 2 by, 2 by + 1 and columns 2 bx, 2 bx + 1 of the 8x8 grid -> nibble bit 2*(lower) + (right)
    const uint32_t* half = reinterpret_cast<const uint32_t*>(qmasks) + (blk >> 3);
    const uint32_t sh = 16u * ((blk >> 2) & 1u) + 2u * (blk & 3u);
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(list));
    // all groups' masks are asked for at once: one LDS round trip for the round instead of one per 64 entries (a wave builds its lists
    // right behind the staging barrier, when every other wave of the workgroup does the same: nobody covers anybody's latency)
    constexpr int NG = CH / 64;
    uint32_t w[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) w[g] = half[2 * (64 * g + lane)];
    uint32_t n = 0;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (64u * g < cnt) {                                // (wave-uniform)
            const uint32_t slot = 64u * g + lane;
            const uint32_t wg = slot < cnt ? w[g] : 0u;
            const uint32_t ent = slot | (__builtin_amdgcn_ubfe(wg, sh, 2u) << 10) | (__builtin_amdgcn_ubfe(wg, sh + 8u, 2u) << 12);
            n += ql_append<0x3c00u>(ent, ent, lb + 2u * n);
        }
    }
    return n;
}
// quadrant lists of block-list entries [c0, min(c0 + 64, n)); returns the longest list's length.  BOUNDED (backward): an entry whose
// list position top - slot is not in front of the quadrant's deepest last contributor (bound[q]) is left out -- no pixel of the
// quadrant blended it (backward.cu:487).
// The lists hold slot << SHIFT (null_slot << SHIFT behind the end): the backward stores byte offsets into its 16-byte record arrays (SHIFT = 4),
// so a pass addresses its records without a shift.
template <bool BOUNDED = false, int SHIFT = 0>
__device__ __forceinline__ uint32_t build_chunk_quadrant_lists(unsigned short (*ql)[QL_ROW], const unsigned short* list, uint32_t c0, uint32_t n, int lane,
                                                              int null_slot, uint32_t top = 0u, const uint32_t* bound = nullptr)
{
    const uint32_t qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(&ql[0][0]));
    ql_fill_null<QL_ROW>(ql, lane, null_slot << SHIFT);
    const uint32_t ent = c0 + lane < n ? list[c0 + lane] : 0u;
    const uint32_t slot = ent & 1023u, val = slot << SHIFT;
    uint32_t c[4];
    if (BOUNDED) {      // top - slot < bound[q]  <=>  slot > top - bound[q]   (slot <= top < 2^31)
        c[0] = ql_append_above<0x0400u>(ent, slot, val, qb, (int)top - (int)bound[0]);
        c[1] = ql_append_above<0x0800u>(ent, slot, val, qb + 2u * QL_ROW, (int)top - (int)bound[1]);
        c[2] = ql_append_above<0x1000u>(ent, slot, val, qb + 4u * QL_ROW, (int)top - (int)bound[2]);
        c[3] = ql_append_above<0x2000u>(ent, slot, val, qb + 6u * QL_ROW, (int)top - (int)bound[3]);
    } else {
        c[0] = ql_append<0x0400u>(ent, val, qb);
        c[1] = ql_append<0x0800u>(ent, val, qb + 2u * QL_ROW);
        c[2] = ql_append<0x1000u>(ent, val, qb + 4u * QL_ROW);
        c[3] = ql_append<0x2000u>(ent, val, qb + 6u * QL_ROW);
    }
    return max(max(c[0], c[1]), max(c[2], c[3]));
}

// The forward's variant: chunks of 128 block-list entries (two appends per quadrant).  Longer chunks even out the four rows of a wave --
// tools/culling_potential.py at config 3: 325 k wave passes with 64-entry chunks, 304 k with 128 -- and halve the number of list builds;
// the forward has the LDS for the longer rows (the backward, at 2 x 75.6 KB per CU, does not).
constexpr int QCH_F = 128;
constexpr int QL_ROW_F = QCH_F + 8;
template <int SHIFT = 0>
__device__ __forceinline__ uint32_t build_chunk_quadrant_lists_128(unsigned short (*ql)[QL_ROW_F], const unsigned short* list, uint32_t c0, uint32_t n, int lane,
                                                                  int null_slot)
{
    const uint32_t qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(&ql[0][0]));
    ql_fill_null<QL_ROW_F>(ql, lane, null_slot << SHIFT);
    uint32_t c[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t i = c0 + 64u * h + lane;
        const uint32_t ent = i < n ? list[i] : 0u;
        const uint32_t val = (ent & 1023u) << SHIFT;
        c[0] += ql_append<0x0400u>(ent, val, qb + 2u * c[0]);
        c[1] += ql_append<0x0800u>(ent, val, qb + 2u * QL_ROW_F + 2u * c[1]);
        c[2] += ql_append<0x1000u>(ent, val, qb + 4u * QL_ROW_F + 2u * c[2]);
        c[3] += ql_append<0x2000u>(ent, val, qb + 6u * QL_ROW_F + 2u * c[3]);
    }
    return max(max(c[0], c[1]), max(c[2], c[3]));
}

// 16-bit block mask -> 4-bit quadrant mask (quadrant q: bit0 = right half, bit1 = lower half)
__device__ __forceinline__ uint32_t block_to_quadrant_mask(uint32_t m)
{
    return ((m & 0x0033u) ? 1u : 0u) | ((m & 0x00CCu) ? 2u : 0u) | ((m & 0x3300u) ? 4u : 0u) | ((m & 0xCC00u) ? 8u : 0u);
}
// 16-B loads / stores of data that is touched once per launch (SH rows, gradient rows): the non-temporal hint keeps them from
// displacing what later kernels gather from L2 (pack lines, counters, records)
typedef float tgs_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p)
{
    const tgs_v4f t = __builtin_nontemporal_load(reinterpret_cast<const tgs_v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store4(float4* p, float4 v)
{
    tgs_v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<tgs_v4f*>(p));
}

// Records staged into LDS by the render kernels carry the conic pre-scaled for one v_exp_f32:
//   log2(e) * power = (A dx + B dy) dx + (C dy) dy   with A = -0.5 log2e conic.x, B = -log2e conic.y, C = -0.5 log2e conic.z
// (forward.cu:336 `power = -0.5f * (con.x*d.x*d.x + con.z*d.y*d.y) - con.y*d.x*d.y`, d = mean - pixel as in this code).
constexpr float LOG2E = 1.4426950408889634f;
constexpr float UNSCALE_CONIC = -2.0f / LOG2E, UNSCALE_CONIC_XY = -1.0f / LOG2E;
__device__ __forceinline__ void stage_conic_a(float4& ra) { ra.z *= -0.5f * LOG2E; ra.w *= -LOG2E; }
__device__ __forceinline__ void stage_conic_b(float4& rb) { rb.x *= -0.5f * LOG2E; }
// log2(e) * power of one (pixel, entry) pair from the staged conic -- ONE instruction sequence for k_render_fwd and k_render_bwd (round 6).  The backward
// replays the forward's decisions (alpha >= 1/255, forward.cu:340-343 / backward.cu:500-504) from its own evaluation of alpha; written as a plain
// expression in both kernels the compiler was free to contract it differently in each, and on one pair in ~10^9 the two evaluations fell on different
// sides of the cut-off: the forward skipped the entry, the backward blended it, and every entry in front of it at that pixel saw T off by 1 / 255 (fuzz
// seed 104 scene 25: one pixel, dL_dconic of an image-filling splat 2.6e-4 from the oracle; profiles/r06_fuzz_soak_b.txt).  Explicit fused
// multiply-adds, contraction off: both kernels round alike, so they decide alike.
__device__ __forceinline__ float pair_power2(float axx, float axy, float ayy, float dx, float dy)
{
#pragma clang fp contract(off)
    return __builtin_fmaf(__builtin_fmaf(axx, dx, axy * dy), dx, (ayy * dy) * dy);
}

// ---------------------------------------------------------------------------------------------
// Hand-scheduled DPP sequences.  hipcc keeps `v_mov_b32 tmp, 0; v_mov_b32_dpp tmp, x` in front of every consumer
// that is not a plain VOP2 with the DPP value in src0 (selects, fma, fmac: 3 instructions where 1 is enough), and
// both render kernels are bound by VALU issue, so the per-group chains are written out as ISA.  Hazards handled
// by ordering / s_nop inside the blocks: a VGPR written by VALU needs 2 wait states before a DPP read, a v_rcp/v_exp
// result 1 before any use (gfx940+).
// ---------------------------------------------------------------------------------------------
#define TGS_QP(E) " quad_perm:[" #E "," #E "," #E "," #E "] row_mask:0xf bank_mask:0xf\n\t"
constexpr unsigned long long QUAD_LT1 = 0x1111111111111111ull;   // lanes with (lane & 3) < 1
constexpr unsigned long long QUAD_LT2 = 0x3333333333333333ull;
constexpr unsigned long long QUAD_LT3 = 0x7777777777777777ull;
// The three lane masks as OPAQUE values in SGPR pairs, made once per kernel (round 4): handed to the chains as literal constants, the
// compiler re-assembled each 64-bit mask from its 32-bit half inside the render loops -- three s_mov_b32 per pass beside the three
// s_mov_b64 vcc the chain itself needs (k_render_fwd: 25 scalar instructions beside 33 vector ones per pass, VERDICT of round 3).
struct QuadMasks { unsigned long long lt1, lt2, lt3; };
__device__ __forceinline__ QuadMasks quad_masks()
{
    QuadMasks m{QUAD_LT1, QUAD_LT2, QUAD_LT3};
    asm volatile("" : "+s"(m.lt1), "+s"(m.lt2), "+s"(m.lt3));
    return m;
}

// Back-to-front walk of one pixel through the 4 entries held by the lanes of its quad (backward.cu:505-531), for the ONE scalar
// the backward needs from accum_rec: dL_dalpha = sum_ch (c_ch - accum_rec_ch) * dL_dpixel_ch (backward.cu:520-527) = s - A with
// s = dL_dpixel . c and A = dL_dpixel . accum_rec, and A obeys the recurrence of every channel (it is linear with the same
// coefficients): A <- a_k s_k + (1 - a_k) A.  One chain instead of one per channel: 21 VALU, not 39; against the oracle the
// gradients moved by < 1e-7 rel-L2 (config 3: dL_dmeans2D 2.45e-7 -> 2.51e-7).
//   in : a = alpha of this lane's entry (0 for a skipped one), s = dL_dpixel . colour of that entry, T / A = state in front of the group
//   out: Town = T after this lane's entry, inv = 1/(1-a), Aown = A seen by this lane's entry; T / A advanced
// Lane e applies entries 0..e-1 in order (one rounding per step); lanes with e <= k take the identity (1, 0) for step k through
// v_cndmask_b32_dpp.
__device__ __forceinline__ void bwd_chain4s(float a, float s, float& T, float& A, float& Town, float& inv, float& Aown, float one, float zero, const QuadMasks& qm)
{
    float om, m, so, t, q;
#define TGS_STEP(K, LT, SRC, SO, QSTEP)                                                       \
        "s_mov_b64 vcc, %[" #LT "]\n\t"                                                       \
        "v_cndmask_b32_dpp %[" #SO "], %[om], %[one], vcc" TGS_QP(K)                           \
        "v_cndmask_b32_dpp %[t], %[m], %[zero], vcc" TGS_QP(K)                                \
        QSTEP                                                                                 \
        "v_fma_f32 %[o], %[" #SRC "], %[" #SO "], %[t]\n\t"
    asm volatile(
        "v_sub_f32 %[om], 1.0, %[a]\n\t"
        "v_mul_f32 %[m], %[a], %[s]\n\t"
        TGS_STEP(0, lt1, A, q, "")                        // (step 0's factor IS the running product: written into q directly, round 5)
        TGS_STEP(1, lt2, o, so, "v_mul_f32 %[q], %[q], %[so]\n\t")
        TGS_STEP(2, lt3, o, so, "v_mul_f32 %[q], %[q], %[so]\n\t")
        "v_mul_f32 %[so], %[q], %[om]\n\t"               // prod_{k<=e} (1 - a_k)
        "v_rcp_f32 %[so], %[so]\n\t"
        "v_fma_f32 %[t], %[o], %[om], %[m]\n\t"          // state behind this lane's own entry
        "v_mul_f32 %[Town], %[T], %[so]\n\t"
        "v_mul_f32 %[inv], %[q], %[so]\n\t"
        "v_mov_b32_dpp %[A], %[t]" TGS_QP(3)
        "v_mov_b32_dpp %[T], %[Town]" TGS_QP(3)
        : [om] "=&v"(om), [m] "=&v"(m), [so] "=&v"(so), [t] "=&v"(t), [q] "=&v"(q), [Town] "=&v"(Town), [inv] "=&v"(inv), [o] "=&v"(Aown),
          [T] "+v"(T), [A] "+v"(A)
        : [a] "v"(a), [s] "v"(s), [one] "v"(one), [zero] "v"(zero), [lt1] "s"(qm.lt1), [lt2] "s"(qm.lt2), [lt3] "s"(qm.lt3)
        : "vcc");
#undef TGS_STEP
}

// Front-to-back walk of one pixel through the 4 entries of its quad (forward.cu:344-357): p = 1 - alpha of this lane's
// entry (1 for a skipped one), T = transmittance in front of the group (uniform over the quad).
//   y = T * p_0 * ... * p_{e-1}  (transmittance in front of this lane's entry, the reference's left-to-right products),
//   x = y * p_e                   (the reference's test_T).
__device__ __forceinline__ void fwd_chain4(float p, float T, float& y, float& x, float one, const QuadMasks& qm)
{
    asm volatile(
        "s_mov_b64 vcc, %[lt1]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_dpp %[y], %[p], %[one], vcc" TGS_QP(0)
        "s_mov_b64 vcc, %[lt2]\n\t"
        "v_cndmask_b32_dpp %[x], %[p], %[one], vcc" TGS_QP(1)
        "v_mul_f32 %[y], %[T], %[y]\n\t"
        "v_mul_f32 %[y], %[y], %[x]\n\t"
        "s_mov_b64 vcc, %[lt3]\n\t"
        "v_cndmask_b32_dpp %[x], %[p], %[one], vcc" TGS_QP(2)
        "v_mul_f32 %[y], %[y], %[x]\n\t"
        "v_mul_f32 %[x], %[y], %[p]\n\t"
        : [y] "=&v"(y), [x] "=&v"(x)
        : [p] "v"(p), [T] "v"(T), [one] "v"(one), [lt1] "s"(qm.lt1), [lt2] "s"(qm.lt2), [lt3] "s"(qm.lt3)
        : "vcc");
}
// c <- max of c over the quad; x3 <- x of the quad's lane 3
__device__ __forceinline__ void quad_max_bcast3(float& c, float x, float& x3)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %[c], %[c], %[c] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[x3], %[x]" TGS_QP(3)
        "s_nop 0\n\t"
        "v_max_f32_dpp %[c], %[c], %[c] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        : [c] "+v"(c), [x3] "=&v"(x3)
        : [x] "v"(x));
}

// The nine sums of a pass over the quadrant's 4 pixels -- the 4 lanes {l, l+4, l+8, l+12} of a row that share an entry slot -- in 18
// v_add_f32_dpp, delivered WHERE THEY ARE ADDED: lane (pixel i, slot e) of a row takes components i and 4 + i of entry e (and 8 if i == 0).
// Level 1 (row_ror:4) on all nine; level 2 (row_ror:8) writes component c's total into ONE register per group of four with bank_mask
// = the bank (lanes 4 i .. 4 i + 3: pixel i) that adds it -- the six v_cndmask that picked a lane's components out of nine full sums
// (until round 4) are gone; v8 is complete in every lane.
__device__ __forceinline__ void row_stride4_sum9_banked(float (&v)[9], float& s0, float& s1)
{
#define TGS_ROR4(I) "v_add_f32_dpp %" #I ", %" #I ", %" #I " row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
#define TGS_ROR8B(D, I, B) "v_add_f32_dpp %[" #D "], %" #I ", %" #I " row_ror:8 row_mask:0xf bank_mask:" #B "\n\t"
    asm volatile("s_nop 1\n\t"
                 TGS_ROR4(0) TGS_ROR4(1) TGS_ROR4(2) TGS_ROR4(3) TGS_ROR4(4) TGS_ROR4(5) TGS_ROR4(6) TGS_ROR4(7) TGS_ROR4(8)
                 TGS_ROR8B(s0, 0, 0x1) TGS_ROR8B(s0, 1, 0x2) TGS_ROR8B(s0, 2, 0x4) TGS_ROR8B(s0, 3, 0x8)
                 TGS_ROR8B(s1, 4, 0x1) TGS_ROR8B(s1, 5, 0x2) TGS_ROR8B(s1, 6, 0x4) TGS_ROR8B(s1, 7, 0x8)
                 "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), [s0] "=&v"(s0), [s1] "=&v"(s1));
#undef TGS_ROR4
#undef TGS_ROR8B
}

// x where `keep`, else 0 -- ONE v_cndmask on the loaded VALUE, for loads issued from a clamped address so that they are in flight together.
// Neither `keep ? x : 0` (turned back into a load under a branch of its own: one memory round trip per value) nor `x * (keep ? 1 : 0)` (round 5;
// 0 * NaN = NaN: a non-finite dL_dpixel at the clamped element -- pixel (0, 0) -- reached every padding lane of the right / bottom edge tiles
// and from there every splat on those tiles; ADVICE of round 5).  Not volatile: the compiler places it like any vector instruction.
__device__ __forceinline__ float select_loaded(bool keep, float x)
{
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(__builtin_amdgcn_ballot_w64(keep)));
    return r;
}

// (error, n_nonempty) of the frame's Meta in ONE load -- the render kernels need both before anything else, and as two loads with a
// branch between them they were two L2 round trips in front of every tile's first record fetch
__device__ __forceinline__ uint2 frame_flags(const ImgState& s)
{
    static_assert(offsetof(Meta, n_nonempty) == offsetof(Meta, error) + 4 && offsetof(Meta, error) % 8 == 0, "Meta layout");
    const unsigned long long v = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(&s.meta->error));
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
// (error, n_nonempty, n_heavy, n_mid) in one 16-B load
__device__ __forceinline__ uint4 frame_counts(const ImgState& s)
{
    static_assert(offsetof(Meta, error) % 16 == 0 && offsetof(Meta, n_mid) == offsetof(Meta, error) + 12, "Meta layout");
    const tgs_v4f t = __builtin_nontemporal_load(reinterpret_cast<const tgs_v4f*>(&s.meta->error));
    return make_uint4(__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w));
}
// Light tiles: fewer than LIGHT_MAX instances (Meta::n_mid counts the tiles with at least that many, k_scan's length classes) -- a single
// staging round, so 4 waves that take the tile's 16 blocks four each need no per-pixel state across rounds: the forward gives such a
// tile ONE 256-thread workgroup (a longer list gets four, one per quarter), and the backward puts three of them into one 1024-thread
// workgroup (bwd_light_group -- what its staging arrays hold).
// Measured (per-tile stamps of the one-workgroup-per-tile kernels, config 3): the 1811 tiles below 128 entries hold 16 % of the instances
// but took 30 % of k_render_fwd's and 24 % of k_render_bwd's workgroup time -- ~3 us of launch / descriptor / record-fetch latency each
// during which a 16-wave workgroup of its own holds half of a CU's wave slots.
constexpr int LIGHT_MAX = 128;
constexpr int BWD_LIGHT_PER_WG = 3;
// A frame that tgs_forward_async could not fit into the caller's binning capacity: every kernel behind k_scan returns.
__device__ __forceinline__ bool frame_rejected(const ImgState& s)
{
    return (__builtin_nontemporal_load(&s.meta->error) & META_ERR_CAPACITY) != 0u;
}

#ifndef TGS_STAMPS
#define TGS_STAMPS 0
#endif
__device__ __forceinline__ void stamp(const ImgState& s, uint32_t tile, int which)
{
#if TGS_STAMPS
    if (threadIdx.x == 0) {
        s.stamps[8 * (size_t)tile + which] = wall_clock64();
        if (!(which & 1)) { s.stamps[8 * (size_t)tile + 4 + which] = 0ull; s.stamps[8 * (size_t)tile + 5 + which] = 0ull; }
    }
#endif
}
// diagnostic builds: time a wave spent between a round's staging barrier and its next barrier (which = 0 forward, 1 backward)
__device__ __forceinline__ unsigned long long busy_clock()
{
#if TGS_STAMPS
    return wall_clock64();
#else
    return 0ull;
#endif
}
__device__ __forceinline__ void busy_report(const ImgState& s, uint32_t tile, int which, unsigned long long dt)
{
#if TGS_STAMPS
    if ((threadIdx.x & 63u) == 0u) { atomicAdd(&s.stamps[8 * (size_t)tile + 4 + 2 * which], dt); atomicMax(&s.stamps[8 * (size_t)tile + 5 + 2 * which], dt); }
#endif
}

// (several workgroups per tile: the latest end)
__device__ __forceinline__ void stamp_max(const ImgState& s, uint32_t tile, int which)
{
#if TGS_STAMPS
    if (threadIdx.x == 0) atomicMax(&s.stamps[8 * (size_t)tile + which], wall_clock64());
#endif
}
// (light groups: the first thread of the tile's quarter stamps)
__device__ __forceinline__ void stamp_if(const ImgState& s, uint32_t tile, int which, bool who)
{
#if TGS_STAMPS
    if (who) s.stamps[8 * (size_t)tile + which] = wall_clock64();
#endif
}

#ifndef TGS_FAST_MATH
#define TGS_FAST_MATH 1                // v_exp_f32 / v_rcp_f32 forms in the render loops (parity measured in tests)
#endif
__device__ __forceinline__ float tgs_exp(float x)
{
#if TGS_FAST_MATH
    return __expf(x);
#else
    return expf(x);
#endif
}
__device__ __forceinline__ float tgs_div(float a, float b)
{
#if TGS_FAST_MATH
    return a * __builtin_amdgcn_rcpf(b);               // v_rcp_f32 (1 ulp) + v_mul_f32
#else
    return a / b;
#endif
}
// wave priority by list length: the kernel ends when the longest list ends, so long lists go first
__device__ __forceinline__ void set_wave_priority(uint32_t n)
{
    if (n > 1024u) __builtin_amdgcn_s_setprio(3);
    else if (n > 512u) __builtin_amdgcn_s_setprio(2);
    else if (n > 256u) __builtin_amdgcn_s_setprio(1);
}

// Sums 36 per-lane values (4 list entries x 9 gradient components, v[e*9+k]) over the 64 lanes of a wave.
// Result: r[k], and in row e (lanes 16e..16e+15) every lane holds the total of entry e, component k.
// v_permlane32_swap / v_permlane16_swap halve the register count while they fold lane halves, so the
// whole reduction is 90 VALU operations instead of 36 x 6.
// NOTE (hipcc / ROCm 7.2, gfx950): `r[0] + r[1]` written directly on the two results of
// __builtin_amdgcn_permlane{16,32}_swap is miscompiled to `v_add_f32 v, r0, r0` (seen in the .s; the second
// result is dropped), so the swap is issued as inline asm.  tgs_selftest_reduce36 (tests/) checks it on hardware.
__device__ __forceinline__ float swap32_add(float x, float y)
{
    // in-place swap of x's upper 32 lanes with y's lower 32 lanes; s_nop 1 = the 2 wait states a VALU-written
    // operand needs before a permlane swap reads it (hipcc pads nothing inside asm)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float swap16_add(float x, float y)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ void wave_reduce36(const float (&v)[36], float (&r)[9])
{
    float a[18];
#pragma unroll
    for (int i = 0; i < 18; i++) a[i] = swap32_add(v[i], v[i + 18]);     // lanes <32: value i, lanes >=32: value i+18
#pragma unroll
    for (int i = 0; i < 9; i++) {
        float x = swap16_add(a[i], a[i + 9]);                            // rows 0..3: values i, i+9, i+18, i+27
        TGS_DPP_ADD(x, 0xB1, 0xf);
        TGS_DPP_ADD(x, 0x4E, 0xf);
        TGS_DPP_ADD(x, 0x141, 0xf);
        TGS_DPP_ADD(x, 0x140, 0xf);
        r[i] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// bitonic network, all comparators ascending ("flip" + "disperse" form): because the larger key
// always moves to the higher index, virtual +inf padding behind n never has to exist in memory.
//   for k = 2,4,..,npad:  flip(k);  for j = k/4,..,1: disperse(j)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_flip(uint32_t t, uint32_t k, uint32_t& i, uint32_t& l)
{
    const uint32_t h = k >> 1;                              // (k is a power of two: no division)
    i = ((t & ~(h - 1)) << 1) | (t & (h - 1));              // block * k + offset, offset < k / 2
    l = i ^ (k - 1);                                        // block * k + (k - 1 - offset)
}
__device__ __forceinline__ void pair_disperse(uint32_t t, uint32_t j, uint32_t& i, uint32_t& l)
{
    i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    l = i + j;
}
__device__ __forceinline__ uint32_t next_pow2(uint32_t n)
{
    return n <= 1 ? 1u : 1u << (32 - __builtin_clz(n - 1));
}

template <typename KeyPtr>
__device__ __forceinline__ void cmp_swap(KeyPtr keys, uint32_t i, uint32_t l, uint32_t n)
{
    if (l < n) {
        const unsigned long long a = keys[i], c = keys[l];
        if (a > c) { keys[i] = c; keys[l] = a; }
    }
}

// Steps of the network that stay inside one aligned chunk of SORT_CHUNK = 128 keys are run by ONE wave in registers
// (regs_sort128 / regs_disperse_from64 below), without LDS traffic or workgroup barriers.
constexpr uint32_t SORT_CHUNK = 128;
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// ---- the same network on one chunk of 128 keys held in REGISTERS: element e of the chunk is k[e & 1] of lane e >> 1 ----
// A step of the network pairs element e with e ^ m (m = k - 1 for flip(k), m = j for disperse(j)); the element whose bit (k/2
// resp. j) is clear keeps the smaller key.  disperse(1) and flip(2) pair a lane's own two registers; every other partner sits in
// another lane: lane ^ 1, 2, 3, 7, 8, 15 are one DPP move per 32-bit half (quad_perm, row_half_mirror, row_ror:8, row_mirror),
// lane ^ 4 two (half mirror + quad reverse), lane ^ 16, 31, 32, 63 a ds_bpermute (LDS crossbar, no memory).  28 steps sort 128 keys
// without an LDS access, ~5 VALU per key and step; as LDS read-compare-write round trips the chunk-local steps kept the LDS pipe of
// a CU busy for the whole kernel.  Call from convergent code with all 64 lanes active.  The keys are unique (the index is their low
// word), so one comparison decides a pair for both sides; between two +inf paddings either answer is right.
template <int CTRL> __device__ __forceinline__ unsigned long long key_dpp(unsigned long long k)
{
    // (bound_ctrl: every lane has a source in these permutations, and with it the compiler need not preset the destination)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)k, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(k >> 32), CTRL, 0xf, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long key_bperm(unsigned long long k, uint32_t src_lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)k);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)(k >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// the key of lane ^ M
template <uint32_t M> __device__ __forceinline__ unsigned long long key_xor(unsigned long long k, uint32_t lane)
{
    if constexpr (M == 1) return key_dpp<0xB1>(k);          // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return key_dpp<0x4E>(k);     // quad_perm [2,3,0,1]
    else if constexpr (M == 3) return key_dpp<0x1B>(k);     // quad_perm [3,2,1,0]
    else if constexpr (M == 7) return key_dpp<0x141>(k);    // row_half_mirror
    else if constexpr (M == 15) return key_dpp<0x140>(k);   // row_mirror
    else if constexpr (M == 8) return key_dpp<0x128>(k);    // row_ror:8
    else if constexpr (M == 4) return key_dpp<0x1B>(key_dpp<0x141>(k));   // (lane ^ 7) ^ 3
    else return key_bperm(k, lane ^ M);
}
// disperse(1) / flip(2): element 2l against 2l + 1
__device__ __forceinline__ void regs_pair(unsigned long long& k0, unsigned long long& k1)
{
    const unsigned long long a = k0, c = k1;
    const bool sw = a > c;
    k0 = sw ? c : a; k1 = sw ? a : c;
}
// disperse(J), J = 2 .. 64: the same register of lane ^ (J / 2)
template <uint32_t J> __device__ __forceinline__ void regs_disperse_step(unsigned long long& k0, unsigned long long& k1, uint32_t lane)
{
    const bool lower = (lane & (J >> 1)) == 0u;
    const unsigned long long o0 = key_xor<(J >> 1)>(k0, lane), o1 = key_xor<(J >> 1)>(k1, lane);
    k0 = ((o0 < k0) == lower) ? o0 : k0;
    k1 = ((o1 < k1) == lower) ? o1 : k1;
}
// flip(K), K = 4 .. 128: the OTHER register of lane ^ (K / 2 - 1)
template <uint32_t K> __device__ __forceinline__ void regs_flip_step(unsigned long long& k0, unsigned long long& k1, uint32_t lane)
{
    const bool lower = (lane & (K >> 2)) == 0u;
    const unsigned long long o0 = key_xor<(K >> 1) - 1>(k1, lane), o1 = key_xor<(K >> 1) - 1>(k0, lane);
    k0 = ((o0 < k0) == lower) ? o0 : k0;
    k1 = ((o1 < k1) == lower) ? o1 : k1;
}
// disperse steps j = 64 .. 1: the tail of a merge whose wider steps ran outside the chunk
__device__ __forceinline__ void regs_disperse_from64(unsigned long long& k0, unsigned long long& k1, uint32_t lane)
{
    regs_disperse_step<64>(k0, k1, lane); regs_disperse_step<32>(k0, k1, lane); regs_disperse_step<16>(k0, k1, lane);
    regs_disperse_step<8>(k0, k1, lane); regs_disperse_step<4>(k0, k1, lane); regs_disperse_step<2>(k0, k1, lane);
    regs_pair(k0, k1);
}
// the whole network on 128 keys (pad with ~0ull behind the real ones)
__device__ __forceinline__ void regs_sort128(unsigned long long& k0, unsigned long long& k1, uint32_t lane)
{
    regs_pair(k0, k1);
    regs_flip_step<4>(k0, k1, lane); regs_pair(k0, k1);
    regs_flip_step<8>(k0, k1, lane); regs_disperse_step<2>(k0, k1, lane); regs_pair(k0, k1);
    regs_flip_step<16>(k0, k1, lane); regs_disperse_step<4>(k0, k1, lane); regs_disperse_step<2>(k0, k1, lane); regs_pair(k0, k1);
    regs_flip_step<32>(k0, k1, lane); regs_disperse_step<8>(k0, k1, lane); regs_disperse_step<4>(k0, k1, lane); regs_disperse_step<2>(k0, k1, lane);
    regs_pair(k0, k1);
    regs_flip_step<64>(k0, k1, lane); regs_disperse_step<16>(k0, k1, lane); regs_disperse_step<8>(k0, k1, lane); regs_disperse_step<4>(k0, k1, lane);
    regs_disperse_step<2>(k0, k1, lane); regs_pair(k0, k1);
    regs_flip_step<128>(k0, k1, lane); regs_disperse_step<32>(k0, k1, lane); regs_disperse_step<16>(k0, k1, lane); regs_disperse_step<8>(k0, k1, lane);
    regs_disperse_step<4>(k0, k1, lane); regs_disperse_step<2>(k0, k1, lane); regs_pair(k0, k1);
}
// chunk [base, base + 128) of a list of n keys (LDS or global memory) <-> registers (virtual +inf behind n)
template <typename KeyPtr>
__device__ __forceinline__ void regs_load_chunk(KeyPtr lk, uint32_t base, uint32_t n, uint32_t lane, unsigned long long& k0, unsigned long long& k1)
{
    const uint32_t e = base + 2 * lane;
    k0 = e < n ? lk[e] : ~0ull;
    k1 = e + 1 < n ? lk[e + 1] : ~0ull;
}
template <typename KeyPtr>
__device__ __forceinline__ void regs_store_chunk(KeyPtr lk, uint32_t base, uint32_t n, uint32_t lane, unsigned long long k0, unsigned long long k1)
{
    const uint32_t e = base + 2 * lane;
    if (e < n) lk[e] = k0;
    if (e + 1 < n) lk[e + 1] = k1;
}

// SH basis constants (cuda_rasterizer/auxiliary.h:22-39)
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                           SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
__device__ constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                           SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                           SH_C3_6 = -0.5900435899266435f;

// auxiliary.h:41-44 -- evaluated in double because the reference's literals are double
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

// auxiliary.h:46-56
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, uint32_t gx, uint32_t gy, uint32_t& minx,
                                         uint32_t& miny, uint32_t& maxx, uint32_t& maxy)
{
    int v;
    v = (int)((px - (float)max_radius) / (float)TILE); v = v < 0 ? 0 : v; minx = (uint32_t)v < gx ? (uint32_t)v : gx;
    v = (int)((py - (float)max_radius) / (float)TILE); v = v < 0 ? 0 : v; miny = (uint32_t)v < gy ? (uint32_t)v : gy;
    v = (int)((px + (float)max_radius + (float)TILE - 1.0f) / (float)TILE); v = v < 0 ? 0 : v; maxx = (uint32_t)v < gx ? (uint32_t)v : gx;
    v = (int)((py + (float)max_radius + (float)TILE - 1.0f) / (float)TILE); v = v < 0 ? 0 : v; maxy = (uint32_t)v < gy ? (uint32_t)v : gy;
}

struct CamParams {          // kernel argument; the matrices stay device pointers like in the reference
    const float* view;      // [16] transposed world-to-camera (consumed as m[0]x+m[4]y+m[8]z+m[12])
    const float* proj;      // [16] full projection, same convention
    const float* campos;    // [3]
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int W, H;
    uint32_t gx, gy;
};
struct ViewMat { float m[16]; };   // a matrix pulled into registers/SGPRs at kernel entry
__device__ __forceinline__ ViewMat load_mat(const float* __restrict__ p)
{
    ViewMat r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.m[i] = p[i];
    return r;
}

// EWA projection shared by forward (forward.cu:74-113) and backward (backward.cu:163-198).
struct Cov2D {
    float tx, ty, tz;        // clamped view-space mean
    float txtz, tytz;
    mat3 T, W, Vrk, cov;     // cov BEFORE the +0.3 dilation
};
__device__ __forceinline__ Cov2D compute_cov2d(float mx, float my, float mz, const float* cov3D, const CamParams& c, const ViewMat& V)
{
#pragma clang fp contract(off)
    Cov2D o;
    const float* vm = V.m;
    float tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
    float ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
    float tz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
    const float limx = 1.3f * c.tan_fovx, limy = 1.3f * c.tan_fovy;
    o.txtz = tx / tz; o.tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, o.txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, o.tytz)) * tz;
    mat3 J = m3make(c.focal_x / tz, 0.0f, -(c.focal_x * tx) / (tz * tz), 0.0f, c.focal_y / tz, -(c.focal_y * ty) / (tz * tz), 0.f, 0.f, 0.f);
    o.W = m3make(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    o.T = m3mul(o.W, J);
    o.Vrk = m3make(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 Tt = m3t(o.T), Vt = m3t(o.Vrk);
    mat3 tmp = m3mul(Tt, Vt);
    o.cov = m3mul(tmp, o.T);
    o.tx = tx; o.ty = ty; o.tz = tz;
    return o;
}

// quaternion (r,x,y,z) -> GLM-layout rotation matrix of forward.cu:134-138 (not normalised, :127)
__device__ __forceinline__ mat3 quat_to_R(float r, float x, float y, float z)
{
#pragma clang fp contract(off)
    return m3make(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                  2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                  2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}
// computeCov3D (forward.cu:118-152): the six upper-triangle entries of Sigma = (S R)^T (S R) from the scale (times the modifier) and the
// un-normalised quaternion.  ONE function for the forward and for both backward kernels: the backward evaluates it again instead of
// reading a copy the forward would have to store (24 B per Gaussian and view written, 24 B read -- and undefined for a Gaussian outside
// the view's frustum), and the two must agree to the bit.
__device__ __forceinline__ void compute_cov3d(float scale_modifier, float s0, float s1, float s2, float4 q, float (&cov3d)[6])
{
#pragma clang fp contract(off)
    const float sx = scale_modifier * s0, sy = scale_modifier * s1, sz = scale_modifier * s2;
    const mat3 S = m3make(sx, 0.f, 0.f, 0.f, sy, 0.f, 0.f, 0.f, sz);
    const mat3 R = quat_to_R(q.x, q.y, q.z, q.w);
    const mat3 Mx = m3mul(S, R);
    const mat3 Sg = m3mul(m3t(Mx), Mx);
    cov3d[0] = Sg.m[0][0]; cov3d[1] = Sg.m[0][1]; cov3d[2] = Sg.m[0][2];
    cov3d[3] = Sg.m[1][1]; cov3d[4] = Sg.m[1][2]; cov3d[5] = Sg.m[2][2];
}
#endif  // __HIPCC__

// ---------------------------------------------------------------------------------------------
// host-side launchers (tgs_forward.hip / tgs_backward.hip), called by tgs_api.hip
// ---------------------------------------------------------------------------------------------
struct FwdIn {
    int P, D, M;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp, *background;
    int prefiltered;
    float* out_color;
    int* radii;
    int prune;        // 1 (default): rectangle tiles the splat cannot reach with alpha >= 1/255 get no instance (tgs_set_instance_pruning)
};
struct BwdIn {
    int P, D, M;
    const float *means3D, *shs, *colors_precomp, *scales, *rotations, *cov3D_precomp, *background;
    const int* radii;
    const float* dL_dpix;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
    int accumulate;   // 1: parameter gradients (all but dL_dmean2D / dL_dconic) are added to what the buffers hold
    const Meta* meta; // the frame's Meta: a frame rejected by tgs_forward_async contributes nothing
    int block0, nblocks;   // k_preprocess_bwd_batch only: first PRE_BLOCK-sized block of Gaussians and how many this launch covers (0: all)
    long long dsh_plane;   // k_preprocess_bwd_batch_split only: 0 = dL_dsh is [P, M, 3] row-major; > 0 = LEVEL-MAJOR, coefficient k of Gaussian p at
                           // dL_dsh[k * dsh_plane + 3 p + c] (floats; a multiple of 4, dL_dsh 16-byte aligned) -- tgs_backward_batch_range_planes
};

// One view of a batch for k_preprocess_bwd_batch: everything that differs between the views (kernel argument)
constexpr int BATCH_VIEWS = 8;
struct BatchView {
    CamParams cam;
    GeomState g;
    BinState b;
    const Meta* meta;
    const int* radii;
    float* dL_dmean2D;   // [P,3] of this view
    float* dL_dcolor;    // [P,3] of this view (colors_precomp path) or nullptr
};
struct BatchViews { int n; BatchView v[BATCH_VIEWS]; };
// One view of a batch for k_preprocess_fwd_batch
struct FwdView {
    CamParams cam;
    GeomState g;
    ImgState s;
    int* radii;
};
struct FwdViews { int n; FwdView v[BATCH_VIEWS]; };

}  // namespace tgs
