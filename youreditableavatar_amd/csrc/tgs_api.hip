// tgs_api.hip -- the C ABI of include/tgs_raster.h: argument checks, state-buffer carving, launches.
// Host orchestration counterpart of cuda_rasterizer/rasterizer_impl.cu:141-434.
#include "tgs_device.hpp"
#include "../../include/tgs_raster.h"
#include "../../include/tgs_raster_testing.h"   // the test-only shims are defined in this file

#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace tgs {
void launch_preprocess_fwd(hipStream_t, const FwdIn&, const CamParams&, const GeomState&, const ImgState&);
void launch_preprocess_fwd_batch(hipStream_t, const FwdIn&, const FwdViews&);
void launch_scan(hipStream_t, const GeomState&, const ImgState&, uint32_t nblocks, uint32_t T, uint32_t sort_cap, unsigned long long r_capacity,
                 uint32_t tile_bound, uint32_t heavy_bound, uint32_t mid_bound, Meta* host_meta, int light);
void launch_bin_count(hipStream_t, int P, const GeomState&, const ImgState&, uint32_t gx, uint32_t T);
void launch_scatter(hipStream_t, int P, const GeomState&, const ImgState&, const BinState&, uint32_t gx, uint32_t T);
void launch_tile_sort(hipStream_t, const GeomState&, const ImgState&, const BinState&, uint32_t gx, uint32_t T, uint64_t r_bound, const Meta* m,
                      uint32_t sort_cap, uint32_t tile_bound, uint32_t heavy_bound, uint32_t mid_bound);
void launch_render_fwd(hipStream_t, const ImgState&, const BinState&, int W, int H, uint32_t gx, uint32_t T, const Meta* m, const float* bg,
                       float* out_color, uint32_t tile_bound, uint32_t mid_bound, int light);
void launch_mark_visible(hipStream_t, int P, const float* means3D, const float* view, uint8_t* present);
void launch_render_bwd(hipStream_t, const ImgState&, const BinState&, int W, int H, uint32_t gx, uint32_t tiles, const float* bg, const float* dL_dpix,
                       bool deterministic, uint32_t mid_tiles, int light, uint32_t T);
void launch_preprocess_bwd(hipStream_t, const BwdIn&, const CamParams&, const GeomState&, const BinState&);
void launch_preprocess_bwd_batch(hipStream_t, const BwdIn&, const BatchViews&);
void launch_selftest_reduce36(hipStream_t, const float* in, float* out);
}  // namespace tgs

using namespace tgs;

static thread_local char g_err[512] = "";

#include <atomic>
#include <cstdlib>
// -1: not set by the API -> environment variable TGS_DETERMINISTIC decides (default 0)
static std::atomic<int> g_deterministic{-1};
static std::atomic<uint32_t> g_sort_cap{SORT_LDS_CAP};
static std::atomic<int> g_fwd_group{2};
static std::atomic<int> g_prune{1};
#ifndef TGS_LIGHT_TILES_DEFAULT
#define TGS_LIGHT_TILES_DEFAULT 1
#endif
static bool deterministic_mode()
{
    const int v = g_deterministic.load(std::memory_order_relaxed);
    if (v >= 0) return v != 0;
    const char* e = getenv("TGS_DETERMINISTIC");
    return e && e[0] && e[0] != '0';
}

// Everything that tunes one call (include/tgs_raster.h: tgs_options_t), resolved ONCE at the entry point and passed down by value: the
// kernels and launchers never read a global.  Fields a caller leaves at their default fall back to the test-only setters' values.
struct Opts {
    int prune;
    bool deterministic;
    int fwd_group;
    uint32_t sort_cap;
    int64_t tile_bound, heavy_bound, mid_bound;
    int light;
};
static thread_local int64_t t_tile_bound = 0;               // tgs_set_tile_bound (test-only shim): default tile bound of this thread's calls without options
// batch: the *_views entry points (several views in flight on several streams), where the light groups pay: +4 % on the 8-view step of
// config 3 (2.16 -> 2.07 ms), while a view that has the GPU to itself loses 1-2 % (its kernels end with the light groups' ~10-us tail)
static Opts resolve_options(const tgs_options_t* o, bool batch = false)
{
    Opts r;
    r.prune = g_prune.load(std::memory_order_relaxed);
    r.deterministic = deterministic_mode();
    r.fwd_group = g_fwd_group.load(std::memory_order_relaxed);
    r.sort_cap = g_sort_cap.load(std::memory_order_relaxed);
    r.tile_bound = t_tile_bound; r.heavy_bound = 0; r.mid_bound = 0;
    static const int env_light = [] { const char* e = getenv("TGS_LIGHT_TILES"); return e ? atoi(e) : -1; }();     // A/B knob, read once
    r.light = env_light >= 0 ? (env_light ? 1 : 0) : (batch ? TGS_LIGHT_TILES_DEFAULT : 0);
    if (!o) return r;
    const size_t n = o->struct_size;
#define TGS_HAS(f) (n >= offsetof(tgs_options_t, f) + sizeof(o->f))
    if (TGS_HAS(instance_pruning) && o->instance_pruning >= 0) r.prune = o->instance_pruning ? 1 : 0;
    if (TGS_HAS(deterministic) && o->deterministic >= 0) r.deterministic = o->deterministic != 0;
    if (TGS_HAS(forward_group) && o->forward_group > 0) r.fwd_group = o->forward_group > BATCH_VIEWS ? BATCH_VIEWS : o->forward_group;
    if (TGS_HAS(sort_lds_cap) && o->sort_lds_cap >= 2 && o->sort_lds_cap <= SORT_LDS_CAP && !(o->sort_lds_cap & (o->sort_lds_cap - 1))) r.sort_cap = o->sort_lds_cap;
    if (TGS_HAS(tile_bound)) r.tile_bound = o->tile_bound > 0 ? o->tile_bound : 0;      // (explicit options: 0 really means none)
    if (TGS_HAS(heavy_bound) && o->heavy_bound > 0) r.heavy_bound = o->heavy_bound;
    if (TGS_HAS(mid_bound) && o->mid_bound > 0) r.mid_bound = o->mid_bound;
    if (TGS_HAS(light_tiles) && o->light_tiles >= 0) r.light = o->light_tiles ? 1 : 0;
#undef TGS_HAS
    return r;
}

// ---- optional per-stage timing (bench only): hipEvents recorded on the caller's stream, no sync ----
#include <mutex>
#include <vector>
namespace {
struct ProfRec { int stage; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
bool g_prof_on = false;
size_t g_prof_cap = 0;
std::vector<ProfRec> g_prof;
hipEvent_t g_prof_open = nullptr;

unsigned g_prof_mask = ~0u;          // stages that get events (tgs_profile_stages)

void prof_begin_stage(hipStream_t st, int stage)
{
    if (!g_prof_on || !((g_prof_mask >> stage) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on || g_prof.size() >= g_prof_cap) { g_prof_open = nullptr; return; }
    if (hipEventCreate(&g_prof_open) != hipSuccess) { g_prof_open = nullptr; return; }
    (void)hipEventRecord(g_prof_open, st);
}
void prof_end_stage(hipStream_t st, int stage)
{
    if (!g_prof_on || !((g_prof_mask >> stage) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_open) return;
    ProfRec r; r.stage = stage; r.e0 = g_prof_open; g_prof_open = nullptr;
    if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
    (void)hipEventRecord(r.e1, st);
    g_prof.push_back(r);
}
}  // namespace

namespace tgs {
int set_error(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
int hip_status(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return TGS_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return TGS_ERR_HIP;
}
}  // namespace tgs

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail(TGS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// the reference's CHECK_CUDA (auxiliary.h:166-173): in debug mode synchronise after every stage
#define STAGE_BEGIN(id) prof_begin_stage(st, id)
#define STAGE_CHECK(name, id)                                                                             \
    do {                                                                                                  \
        prof_end_stage(st, id);                                                                                                  \
        hipError_t e_ = hipGetLastError();                                                                \
        if (e_ == hipSuccess && debug) e_ = hipStreamSynchronize(st);                                     \
        if (e_ != hipSuccess) return fail(TGS_ERR_HIP, "stage %s: %s", name, hipGetErrorString(e_));       \
    } while (0)

static CamParams make_cam(const float* view, const float* proj, const float* campos, float tan_fovx, float tan_fovy, float scale_modifier,
                          int W, int H)
{
    CamParams c;
    c.view = view; c.proj = proj; c.campos = campos;
    c.tan_fovx = tan_fovx; c.tan_fovy = tan_fovy;
    c.focal_y = H / (2.0f * tan_fovy);            // rasterizer_impl.cu:222-223
    c.focal_x = W / (2.0f * tan_fovx);
    c.scale_modifier = scale_modifier;
    c.W = W; c.H = H;
    c.gx = (uint32_t)((W + TILE - 1) / TILE);
    c.gy = (uint32_t)((H + TILE - 1) / TILE);
    return c;
}

extern "C" {

int tgs_abi_version(void) { return TGS_ABI_VERSION; }

void tgs_set_instance_pruning(int on) { g_prune.store(on ? 1 : 0, std::memory_order_relaxed); }

void tgs_set_forward_group(int views_per_launch)
{
    g_fwd_group.store(views_per_launch < 1 ? 1 : (views_per_launch > BATCH_VIEWS ? BATCH_VIEWS : views_per_launch), std::memory_order_relaxed);
}

int tgs_set_sort_lds_cap(unsigned cap)
{
    if (cap < 2 || cap > SORT_LDS_CAP || (cap & (cap - 1))) return fail(TGS_ERR_INVALID, "sort cap must be a power of two in [2, %u]", SORT_LDS_CAP);
    g_sort_cap.store(cap, std::memory_order_relaxed);
    return TGS_OK;
}

void tgs_set_deterministic(int on) { g_deterministic.store(on < 0 ? -1 : (on ? 1 : 0), std::memory_order_relaxed); }

int tgs_selftest_reduce36(void* stream, const float* in, float* out)
{
    launch_selftest_reduce36((hipStream_t)stream, in, out);
    return hipGetLastError() == hipSuccess ? TGS_OK : TGS_ERR_HIP;
}

int tgs_profile_begin(int max_records)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_on) return TGS_ERR_INVALID;
    g_prof.clear();
    g_prof_cap = max_records > 0 ? (size_t)max_records : 0;
    g_prof.reserve(g_prof_cap);
    g_prof_open = nullptr;
    g_prof_on = true;
    return TGS_OK;
}

void tgs_profile_stages(unsigned mask)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_mask = mask;
}

int tgs_profile_end(double* ms_sum, int64_t* counts)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return TGS_ERR_INVALID;
    g_prof_on = false;
    for (int i = 0; i < TGS_STAGE_COUNT; i++) { ms_sum[i] = 0.0; counts[i] = 0; }
    for (ProfRec& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess &&
            r.stage >= 0 && r.stage < TGS_STAGE_COUNT) { ms_sum[r.stage] += ms; counts[r.stage]++; }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    return TGS_OK;
}
const char* tgs_last_error(void) { return g_err; }

// pinned Meta staging + event of the speculative forward: one per host thread AND device (an event belongs to the device that was
// current when it was created; a thread that renders on cuda:0 and then on cuda:1 gets a slot for each)
struct SpecSlot { Meta* meta; hipEvent_t ready; };
static SpecSlot* spec_slot()
{
    thread_local std::vector<SpecSlot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    if ((size_t)dev >= slots.size()) slots.resize((size_t)dev + 1, SpecSlot{nullptr, nullptr});
    SpecSlot& slot = slots[(size_t)dev];
    if (!slot.meta) {
        if (hipHostMalloc((void**)&slot.meta, sizeof(Meta), hipHostMallocDefault) != hipSuccess) { slot.meta = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&slot.ready, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(slot.meta); slot.meta = nullptr; return nullptr; }
    }
    return &slot;
}

// r_capacity < 0: the reference's protocol (read R back, then size the binning buffer).  r_capacity >= 0: sync-free --
// the binning buffer is sized for r_capacity instances before anything runs and nothing is read back.
// tgs_set_render_streams: k_render_fwd of view k goes to render stream k mod n (behind an event on the view's own stream)
static thread_local std::vector<hipStream_t> t_render_streams;
static thread_local int64_t t_last_nonempty = -1;           // tgs_last_nonempty_tiles (legacy read-out; tgs_frame_info_t carries it explicitly)
static uint32_t bounded_tiles(const Opts& o, size_t T) { return (o.tile_bound > 0 && (uint64_t)o.tile_bound < (uint64_t)T) ? (uint32_t)o.tile_bound : (uint32_t)T; }
// bound on the tiles with >= LIGHT_MAX instances (the 1024-thread render kernel's grid): the caller's mid_bound, else the tile bound
static uint32_t bounded_mid(const Opts& o, size_t T)
{
    const uint32_t tb = bounded_tiles(o, T);
    // the SAME predicate as the forward's (forward_impl: `classes`): a mid bound is only enforced -- by k_scan, which rejects a frame with more
    // such tiles -- together with a tile bound below the tile count; without that enforcement the backward must not size its grid by it
    return (tb < (uint32_t)T && o.mid_bound > 0 && (uint64_t)o.mid_bound < tb) ? (uint32_t)o.mid_bound : tb;
}

static int64_t forward_impl(const Opts& opt, Meta* host_meta, hipStream_t render_stream, tgs_frame_info_t* info, int preprocessed, int64_t r_capacity, int64_t* speculative_true_R, tgs_alloc_fn alloc, void* alloc_ctx, void* stream, int P, int D, int M, const float* background, int width,
                    int height, const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                    int prefiltered, float* out_color, int* radii, int debug)
{
    const bool async = r_capacity >= 0;
    t_last_nonempty = -1;                                   // (known again once this call has read the frame's Meta)
    if (info) { info->num_rendered = -1; info->nonempty_tiles = -1; info->flags = 0; info->mid_tiles = -1; }
    if (r_capacity > 0x7fffffffll) return fail(TGS_ERR_INVALID, "r_capacity exceeds 2^31-1");
    hipStream_t st = (hipStream_t)stream;
    g_err[0] = 0;
    if (!alloc) return fail(TGS_ERR_INVALID, "alloc callback is NULL");
    if (P < 0 || width <= 0 || height <= 0) return fail(TGS_ERR_INVALID, "bad sizes P=%d W=%d H=%d", P, width, height);
    if (P == 0) {   // rasterize_points.cu:81: nothing runs, the image keeps its zero fill (empty inputs have no pointers to check)
        if (!out_color) return fail(TGS_ERR_INVALID, "NULL required pointer");
        HIP_TRY(hipMemsetAsync(out_color, 0, 3 * (size_t)width * height * sizeof(float), st));
        if (host_meta) memset(host_meta, 0, sizeof(Meta));   // (pinned host memory: no kernel of this frame writes it)
        if (info && !(async && !speculative_true_R)) { info->num_rendered = 0; info->nonempty_tiles = 0; info->mid_tiles = 0; }
        if (async) {   // the caller still gets a (zeroed) Meta to query
            ImgState s0;
            const size_t bytes = img_carve(s0, nullptr, (size_t)width * height, (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE));
            char* ip = (char*)alloc(alloc_ctx, TGS_BUF_IMAGE, bytes);
            if (!ip) return fail(TGS_ERR_ALLOC, "state buffer allocation failed");
            HIP_TRY(hipMemsetAsync(ip, 0, 256, st));
        }
        return 0;
    }
    if ((shs == nullptr) == (colors_precomp == nullptr)) return fail(TGS_ERR_INVALID, "provide exactly one of shs / colors_precomp");
    const bool has_sr = scales != nullptr && rotations != nullptr;
    if (has_sr == (cov3D_precomp != nullptr) || (scales == nullptr) != (rotations == nullptr))
        return fail(TGS_ERR_INVALID, "provide exactly one of (scales, rotations) / cov3D_precomp");
    if (shs && (D < 0 || D > 3 || M < (D + 1) * (D + 1))) return fail(TGS_ERR_INVALID, "SH degree %d needs M >= %d (M=%d)", D, (D + 1) * (D + 1), M);
    if (!background || !means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos || !out_color)
        return fail(TGS_ERR_INVALID, "NULL required pointer");
    const CamParams cam = make_cam(viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, scale_modifier, width, height);
    if (cam.gx > 65535u || cam.gy > 65535u) return fail(TGS_ERR_INVALID, "image too large");
    const size_t N = (size_t)width * height, T = (size_t)cam.gx * cam.gy;
    const bool has_sh = shs != nullptr;

    const uint32_t sort_cap = opt.sort_cap;
    GeomState g; ImgState s; BinState b;
    const size_t geom_bytes = geom_carve(g, nullptr, (size_t)P, has_sh, has_sr);
    const size_t img_bytes = img_carve(s, nullptr, N, T);
    char* geom_ptr = (char*)alloc(alloc_ctx, TGS_BUF_GEOM, geom_bytes);
    char* img_ptr = (char*)alloc(alloc_ctx, TGS_BUF_IMAGE, img_bytes);
    if (!geom_ptr || !img_ptr) return fail(TGS_ERR_ALLOC, "state buffer allocation failed");
    geom_carve(g, geom_ptr, (size_t)P, has_sh, has_sr);
    img_carve(s, img_ptr, N, T);

    // Meta sits at the head of the image buffer; k_preprocess_fwd* clears it itself (block_sum_tiles: nothing else writes Meta before
    // k_scan), everything else is written before it is read -- no memset launch in front of a frame

    FwdIn in;
    in.P = P; in.D = D; in.M = M; in.means3D = means3D; in.shs = shs; in.colors_precomp = colors_precomp; in.opacities = opacities;
    in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp; in.background = background;
    in.prefiltered = prefiltered; in.out_color = out_color; in.radii = radii; in.prune = opt.prune;

    uint64_t R = 0;
    Meta meta;
    memset(&meta, 0, sizeof(meta));
    if (!preprocessed) {
        STAGE_BEGIN(TGS_STAGE_PREPROCESS_FWD);
        launch_preprocess_fwd(st, in, cam, g, s);
        STAGE_CHECK("preprocess", TGS_STAGE_PREPROCESS_FWD);
    }
    SpecSlot* spec = nullptr;
    if (async && speculative_true_R) {
        spec = spec_slot();
        if (!spec) return fail(TGS_ERR_HIP, "pinned staging for the speculative forward could not be allocated");
    }
    // sync-free grids cover `tb` tiles (the caller's bound on the tiles with instances, or all of them); k_scan rejects a frame with more
    const uint32_t tb = async ? bounded_tiles(opt, T) : (uint32_t)T;
    const bool classes = async && tb < T;                   // class bounds come with a tile bound only
    const uint32_t hb = (classes && opt.heavy_bound > 0 && (uint64_t)opt.heavy_bound < tb) ? (uint32_t)opt.heavy_bound : tb;
    const uint32_t mb = (classes && opt.mid_bound > 0 && (uint64_t)opt.mid_bound < tb) ? (uint32_t)opt.mid_bound : tb;
    STAGE_BEGIN(TGS_STAGE_SCAN);
    launch_bin_count(st, P, g, s, cam.gx, (uint32_t)T);
    launch_scan(st, g, s, (uint32_t)n_blocks((size_t)P), (uint32_t)T, sort_cap, async ? (unsigned long long)r_capacity : ~0ull, tb, hb, mb,
                spec ? spec->meta : host_meta, opt.light);
    STAGE_CHECK("scan", TGS_STAGE_SCAN);
    if (spec) {
        // speculative synchronous forward: k_scan itself has written Meta into the pinned host slot; the event marks its end, the remaining
        // stages are enqueued against the guessed capacity without waiting, and only then the host waits for the event -- the GPU never
        // idles behind the read-back, and no copy sits in the stream
        HIP_TRY(hipEventRecord(spec->ready, st));
    }
    if (!async) {
        // the one host synchronisation of the forward pass (rasterizer_impl.cu:280-281): R sizes the binning buffer
        HIP_TRY(hipMemcpyAsync(&meta, s.meta, sizeof(Meta), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (meta.error & 1u) return fail(TGS_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
        R = meta.R;
        if (R > 0x7fffffffull) return fail(TGS_ERR_TOO_MANY, "%llu tile instances exceed 2^31-1", (unsigned long long)R);
        t_last_nonempty = (int64_t)meta.n_nonempty;
        if (info) { info->num_rendered = (int64_t)meta.R; info->nonempty_tiles = (int64_t)meta.n_nonempty; info->flags = (int32_t)meta.error; info->mid_tiles = (int32_t)meta.n_mid; }
    } else {
        R = (uint64_t)r_capacity;
    }
    const size_t bin_bytes = bin_carve(b, nullptr, (size_t)R);
    char* bin_ptr = (char*)alloc(alloc_ctx, TGS_BUF_BINNING, bin_bytes);
    if (!bin_ptr) return fail(TGS_ERR_ALLOC, "binning buffer allocation failed");
    bin_carve(b, bin_ptr, (size_t)R);
    const Meta* known = async ? nullptr : &meta;

    if (R > 0) {
        STAGE_BEGIN(TGS_STAGE_SCATTER);
        launch_scatter(st, P, g, s, b, cam.gx, (uint32_t)T);
        STAGE_CHECK("scatter", TGS_STAGE_SCATTER);
    }
    if (R > 0) {
        STAGE_BEGIN(TGS_STAGE_TILE_SORT);
        launch_tile_sort(st, g, s, b, cam.gx, (uint32_t)T, R, known, sort_cap, tb, hb, mb);
        STAGE_CHECK("tile_sort", TGS_STAGE_TILE_SORT);
    }
    hipStream_t rst = st;
    if (render_stream && render_stream != st && !spec) {     // binning and compositing on different streams (tgs_set_render_streams)
        hipEvent_t binned;
        HIP_TRY(hipEventCreateWithFlags(&binned, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(binned, st));
        HIP_TRY(hipStreamWaitEvent(render_stream, binned, 0));
        (void)hipEventDestroy(binned);
        rst = render_stream;
    }
    STAGE_BEGIN(TGS_STAGE_RENDER_FWD);
    launch_render_fwd(rst, s, b, width, height, cam.gx, (uint32_t)T, known, background, out_color, tb, mb, opt.light);
    STAGE_CHECK("render", TGS_STAGE_RENDER_FWD);
    if (spec) {
        HIP_TRY(hipEventSynchronize(spec->ready));
        meta = *spec->meta;
        if (meta.error & 1u) return fail(TGS_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
        if (meta.R > 0x7fffffffull) return fail(TGS_ERR_TOO_MANY, "%llu tile instances exceed 2^31-1", (unsigned long long)meta.R);
        *speculative_true_R = (int64_t)meta.R;
        t_last_nonempty = (int64_t)meta.n_nonempty;
        if (info) { info->num_rendered = (int64_t)meta.R; info->nonempty_tiles = (int64_t)meta.n_nonempty; info->flags = (int32_t)(meta.error & ~META_ERR_CAPACITY); info->mid_tiles = (int32_t)meta.n_mid; }
        if ((meta.error & META_ERR_CAPACITY) || meta.pad[0] != 0u) {     // (pad[0]: more tiles with instances than the caller's bound, k_scan)
            // the guess was too small: every kernel behind the scan returned at
            // once; clear the flag and run those stages again with the exact sizes, as tgs_forward does
            HIP_TRY(hipMemsetAsync(&s.meta->error, 0, sizeof(uint32_t), st));
            R = meta.R;
            const size_t exact_bytes = bin_carve(b, nullptr, (size_t)R);
            char* exact_ptr = (char*)alloc(alloc_ctx, TGS_BUF_BINNING, exact_bytes);
            if (!exact_ptr) return fail(TGS_ERR_ALLOC, "binning buffer allocation failed");
            bin_carve(b, exact_ptr, (size_t)R);
            if (R > 0) {
                launch_scatter(st, P, g, s, b, cam.gx, (uint32_t)T);
                launch_tile_sort(st, g, s, b, cam.gx, (uint32_t)T, R, &meta, sort_cap, (uint32_t)T, (uint32_t)T, (uint32_t)T);
            }
            launch_render_fwd(st, s, b, width, height, cam.gx, (uint32_t)T, &meta, background, out_color, (uint32_t)T, (uint32_t)T, opt.light);
            HIP_TRY(hipGetLastError());
        }
    }
    return (int64_t)R;
}

int64_t tgs_forward(tgs_alloc_fn alloc, void* alloc_ctx, void* stream, int P, int D, int M, const float* background, int width,
                    int height, const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                    int prefiltered, float* out_color, int* radii, int debug)
{
    return tgs_forward_opt(nullptr, TGS_FWD_SYNC, 0, nullptr, alloc, alloc_ctx, stream, P, D, M, background, width, height, means3D, shs, colors_precomp, opacities,
                           scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, radii, debug);
}

int64_t tgs_forward_opt(const tgs_options_t* o, int mode, int64_t r, tgs_frame_info_t* info, tgs_alloc_fn alloc, void* alloc_ctx, void* stream, int P, int D, int M,
                        const float* background, int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                        float* out_color, int* radii, int debug)
{
    const Opts opt = resolve_options(o);
    int64_t true_R = 0;
    if (mode == TGS_FWD_SYNC) r = -1;
    else if (mode != TGS_FWD_ASYNC && mode != TGS_FWD_SPECULATIVE) return fail(TGS_ERR_INVALID, "tgs_forward_opt: unknown mode %d", mode);
    else if (r < 0) return fail(TGS_ERR_INVALID, "r_capacity / r_guess must be >= 0");
    return forward_impl(opt, nullptr, nullptr, info, 0, r, mode == TGS_FWD_SPECULATIVE ? &true_R : nullptr, alloc, alloc_ctx, stream, P, D, M, background, width, height,
                        means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
                        prefiltered, out_color, radii, debug);
}

int64_t tgs_forward_async(int64_t r_capacity, tgs_alloc_fn alloc, void* alloc_ctx, void* stream, int P, int D, int M, const float* background,
                          int width, int height, const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                          const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                          int prefiltered, float* out_color, int* radii, int debug)
{
    if (r_capacity < 0) return fail(TGS_ERR_INVALID, "r_capacity must be >= 0");
    return tgs_forward_opt(nullptr, TGS_FWD_ASYNC, r_capacity, nullptr, alloc, alloc_ctx, stream, P, D, M, background, width, height, means3D, shs, colors_precomp, opacities,
                           scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, radii, debug);
}

int64_t tgs_forward_speculative(int64_t r_guess, int64_t* num_rendered, tgs_alloc_fn alloc, void* alloc_ctx, void* stream, int P, int D, int M,
                                const float* background, int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                                const float* opacities, const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                int prefiltered, float* out_color, int* radii, int debug)
{
    if (r_guess < 0 || !num_rendered) return fail(TGS_ERR_INVALID, "r_guess must be >= 0 and num_rendered non-NULL");
    *num_rendered = 0;
    tgs_frame_info_t info;
    const int64_t r = tgs_forward_opt(nullptr, TGS_FWD_SPECULATIVE, r_guess, &info, alloc, alloc_ctx, stream, P, D, M, background, width, height, means3D, shs, colors_precomp,
                                      opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color,
                                      radii, debug);
    if (r >= 0 && info.num_rendered >= 0) *num_rendered = info.num_rendered;
    return r;
}

int tgs_frame_status(void* stream, const void* img_buffer, int64_t* num_rendered, int* flags)
{
    g_err[0] = 0;
    if (!img_buffer || !num_rendered || !flags) return fail(TGS_ERR_INVALID, "NULL required pointer");
    Meta meta;
    ImgState s;
    img_carve(s, (char*)img_buffer, 0, 0);                  // Meta is the first field of the image buffer
    HIP_TRY(hipMemcpyAsync(&meta, s.meta, sizeof(Meta), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *num_rendered = (int64_t)meta.R;
    *flags = (int)meta.error;
    if (meta.error & META_ERR_TILE_BOUND)
        return fail(TGS_ERR_INVALID, "a backward of this frame ran with a tile bound below its %u tiles with instances: gradients are incomplete", meta.n_nonempty);
    return TGS_OK;
}

// strict: tgs_backward / tgs_backward_accumulate as the reference's Rasterizer::backward declares them (every output required);
// tgs_backward_opt: the outputs its caller discards may be NULL (dL_dconic; dL_dcolor with shs; dL_dcov3D with scales + rotations)
static int backward_impl(bool strict, const Opts& opt, int accumulate, void* stream, int P, int D, int M, int64_t R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                 float tan_fovy, const int* radii, const void* geom_buffer, const void* binning_buffer, const void* img_buffer,
                 const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                 float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug)
{
    hipStream_t st = (hipStream_t)stream;
    g_err[0] = 0;
    if (P == 0) return TGS_OK;
    if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(TGS_ERR_INVALID, "bad sizes");
    const bool has_sh = shs != nullptr, has_sr = scales != nullptr && rotations != nullptr;
    if (has_sh == (colors_precomp != nullptr)) return fail(TGS_ERR_INVALID, "provide exactly one of shs / colors_precomp");
    if (has_sr == (cov3D_precomp != nullptr)) return fail(TGS_ERR_INVALID, "provide exactly one of (scales, rotations) / cov3D_precomp");
    if (!geom_buffer || !binning_buffer || !img_buffer || !radii || !dL_dpix || !dL_dmean2D || !dL_dopacity || !dL_dmean3D ||
        (has_sh && !dL_dsh) || (!has_sh && !dL_dcolor) || (!has_sr && !dL_dcov3D) ||
        (strict && (!dL_dconic || (!accumulate && (!dL_dcolor || !dL_dcov3D)))))
        return fail(TGS_ERR_INVALID, "NULL required pointer");
    const CamParams cam = make_cam(viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, scale_modifier, width, height);
    const size_t N = (size_t)width * height, T = (size_t)cam.gx * cam.gy;
    GeomState g; ImgState s; BinState b;
    geom_carve(g, (char*)geom_buffer, (size_t)P, has_sh, has_sr);
    img_carve(s, (char*)img_buffer, N, T);
    bin_carve(b, (char*)binning_buffer, (size_t)R);

    BwdIn in;
    in.P = P; in.D = D; in.M = M; in.means3D = means3D; in.shs = shs; in.colors_precomp = colors_precomp; in.scales = scales;
    in.rotations = rotations; in.cov3D_precomp = cov3D_precomp; in.background = background; in.radii = radii; in.dL_dpix = dL_dpix;
    in.dL_dmean2D = dL_dmean2D; in.dL_dconic = dL_dconic; in.dL_dopacity = dL_dopacity; in.dL_dcolor = dL_dcolor;
    in.dL_dmean3D = dL_dmean3D; in.dL_dcov3D = dL_dcov3D; in.dL_dsh = dL_dsh; in.dL_dscale = dL_dscale; in.dL_drot = dL_drot;
    in.accumulate = accumulate;
    in.meta = s.meta;
    in.block0 = 0; in.nblocks = 0;

    if (R > 0) {
        STAGE_BEGIN(TGS_STAGE_RENDER_BWD);
        launch_render_bwd(st, s, b, width, height, cam.gx, bounded_tiles(opt, T), background, dL_dpix, opt.deterministic, bounded_mid(opt, T), opt.light, (uint32_t)T);
        STAGE_CHECK("render_bwd", TGS_STAGE_RENDER_BWD);
    }
    STAGE_BEGIN(TGS_STAGE_PREPROCESS_BWD);
    launch_preprocess_bwd(st, in, cam, g, b);
    STAGE_CHECK("preprocess_bwd", TGS_STAGE_PREPROCESS_BWD);
    return TGS_OK;
}

int tgs_backward(void* stream, int P, int D, int M, int64_t R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                 float tan_fovy, const int* radii, const void* geom_buffer, const void* binning_buffer, const void* img_buffer,
                 const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                 float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug)
{
    return backward_impl(true, resolve_options(nullptr), 0, stream, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                         viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, img_buffer, dL_dpix, dL_dmean2D,
                         dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug);
}

int tgs_backward_opt(const tgs_options_t* o, int accumulate, void* stream, int P, int D, int M, int64_t R, const float* background, int width, int height,
                     const float* means3D, const float* shs, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                     const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                     const int* radii, const void* geom_buffer, const void* binning_buffer, const void* img_buffer, const float* dL_dpix, float* dL_dmean2D,
                     float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                     int debug)
{
    return backward_impl(false, resolve_options(o), accumulate ? 1 : 0, stream, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                         cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, img_buffer, dL_dpix, dL_dmean2D,
                         dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug);
}

int tgs_backward_accumulate(void* stream, int P, int D, int M, int64_t R, const float* background, int width, int height, const float* means3D,
                            const float* shs, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                            const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                            float tan_fovy, const int* radii, const void* geom_buffer, const void* binning_buffer, const void* img_buffer,
                            const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                            float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug)
{
    return backward_impl(true, resolve_options(nullptr), 1, stream, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                         viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, img_buffer, dL_dpix, dL_dmean2D,
                         dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug);
}

void tgs_state_sizes(int P, int width, int height, int has_sh, int has_scale_rot, int64_t r_capacity, size_t sizes3[3])
{
    GeomState g; ImgState s; BinState b;
    const size_t gx = (size_t)((width + TILE - 1) / TILE), gy = (size_t)((height + TILE - 1) / TILE);
    sizes3[TGS_BUF_GEOM] = geom_carve(g, nullptr, (size_t)(P > 0 ? P : 0), has_sh != 0, has_scale_rot != 0);
    sizes3[TGS_BUF_BINNING] = bin_carve(b, nullptr, (size_t)(r_capacity > 0 ? r_capacity : 0));
    sizes3[TGS_BUF_IMAGE] = img_carve(s, nullptr, (size_t)width * height, gx * gy);
}

// allocation "callback" of the *_views entry points: hands out the caller's preset buffers
static void* alloc_preset(void* ctx, int which, size_t bytes)
{
    const tgs_view_t* v = (const tgs_view_t*)ctx;
    if (which == TGS_BUF_GEOM) return bytes <= v->geom_bytes ? const_cast<void*>(v->geom_buffer) : nullptr;
    if (which == TGS_BUF_BINNING) return bytes <= v->binning_bytes ? const_cast<void*>(v->binning_buffer) : nullptr;
    if (which == TGS_BUF_IMAGE) return bytes <= v->img_bytes ? const_cast<void*>(v->img_buffer) : nullptr;
    return nullptr;
}

void tgs_set_tile_bound(int64_t n) { t_tile_bound = n > 0 ? n : 0; }
int64_t tgs_last_nonempty_tiles(void) { return t_last_nonempty; }

int tgs_set_render_streams(void* const* streams, int n)
{
    t_render_streams.clear();
    for (int i = 0; i < n && streams; i++) t_render_streams.push_back((hipStream_t)streams[i]);
    return TGS_OK;
}

size_t tgs_sizeof_view(void) { return sizeof(tgs_view_t); }
size_t tgs_sizeof_options(void) { return sizeof(tgs_options_t); }

int tgs_forward_views(void* const* streams, int n_streams, int64_t r_capacity, int P, int D, int M, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                      const float* cov3D_precomp, int prefiltered, int n_views, tgs_view_t* views)
{
    return tgs_forward_views_opt(nullptr, streams, n_streams, r_capacity, P, D, M, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                                 prefiltered, n_views, views);
}

int tgs_forward_views_opt(const tgs_options_t* o, void* const* streams, int n_streams, int64_t r_capacity, int P, int D, int M, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                          const float* cov3D_precomp, int prefiltered, int n_views, tgs_view_t* views)
{
    const Opts opt0 = resolve_options(o, true);
    g_err[0] = 0;
    if (n_views == 0) return TGS_OK;
    if (!streams || n_streams <= 0 || n_views < 0 || !views || r_capacity < 0) return fail(TGS_ERR_INVALID, "bad arguments");
    const int debug = 0;
    const bool has_sh = shs != nullptr, has_sr = scales != nullptr && rotations != nullptr;
    // views that can share the per-Gaussian stage: same model, the usual case (P > 0, all pointers there; the rest is
    // validated by forward_impl below)
    // Views per launch of the shared per-Gaussian stage (tgs_set_forward_group, TGS_FORWARD_GROUP; default 2).  More views per launch
    // read the SH rows -- more than half of what the stage reads -- fewer times: 51 us for one view, 78 / 125 / 218 us for 2 / 4 / 8.
    // But the views of a group start their remaining stages together, and with four streams that costs more than the bytes save
    // beyond pairs: 0.302 / 0.296 / 0.308 / 0.306 ms per frame for groups of 1 / 2 / 4 / 8 at config 3.  (With the counting atomics
    // in this stage, until round 2, the launch was bound by them and no grouping paid.)
    static const int env_group = [] { const char* e = getenv("TGS_FORWARD_GROUP"); return e ? atoi(e) : 0; }();   // tuning knob, read once
    int group = opt0.fwd_group;
    if (env_group > 0) group = env_group > BATCH_VIEWS ? BATCH_VIEWS : env_group;
    bool per_view_colors = false;
    for (int k = 0; k < n_views; k++) per_view_colors = per_view_colors || views[k].colors_precomp != nullptr;
    if (per_view_colors && shs) return fail(TGS_ERR_INVALID, "provide exactly one of shs / colors_precomp");
    const bool batched = !per_view_colors && group > 1 && n_views > 1 && P > 0 && means3D && opacities && ((shs == nullptr) != (colors_precomp == nullptr)) && (has_sr != (cov3D_precomp != nullptr)) &&
                         (!has_sh || (D >= 0 && D <= 3 && M >= (D + 1) * (D + 1)));
    for (int v0 = 0; v0 < n_views; v0 += group) {
        const int nv = n_views - v0 < group ? n_views - v0 : group;
        hipStream_t st0 = (hipStream_t)streams[v0 % n_streams];
        hipEvent_t pre_done = nullptr;
        for (int k = 0; k < nv; k++) {
            const tgs_view_t& v = views[v0 + k];
            if (!v.geom_buffer || !v.binning_buffer || !v.img_buffer || !v.out_color || !v.background || !v.viewmatrix || !v.projmatrix || !v.campos ||
                v.width <= 0 || v.height <= 0)
                return fail(TGS_ERR_INVALID, "view %d: bad sizes or NULL required pointer", v0 + k);
        }
        if (batched) {
            FwdIn in;
            memset(&in, 0, sizeof(in));
            in.P = P; in.D = D; in.M = M; in.means3D = means3D; in.shs = shs; in.colors_precomp = colors_precomp; in.opacities = opacities;
            in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp; in.prefiltered = prefiltered;
            in.prune = opt0.prune;
            FwdViews fv;
            memset(&fv, 0, sizeof(fv));
            fv.n = nv;
            for (int k = 0; k < nv; k++) {
                const tgs_view_t& v = views[v0 + k];
                FwdView& o = fv.v[k];
                o.cam = make_cam(v.viewmatrix, v.projmatrix, v.campos, v.tan_fovx, v.tan_fovy, scale_modifier, v.width, v.height);
                if (o.cam.gx > 65535u || o.cam.gy > 65535u) return fail(TGS_ERR_INVALID, "image too large");
                const size_t N = (size_t)v.width * v.height, T = (size_t)o.cam.gx * o.cam.gy;
                if (geom_carve(o.g, nullptr, (size_t)P, has_sh, has_sr) > v.geom_bytes || img_carve(o.s, nullptr, N, T) > v.img_bytes)
                    return fail(TGS_ERR_ALLOC, "view %d: state buffers smaller than tgs_state_sizes()", v0 + k);
                geom_carve(o.g, (char*)v.geom_buffer, (size_t)P, has_sh, has_sr);
                img_carve(o.s, (char*)v.img_buffer, N, T);
                o.radii = v.radii_out;
            }
            {
                hipStream_t st = st0;
                STAGE_BEGIN(TGS_STAGE_PREPROCESS_FWD);
                launch_preprocess_fwd_batch(st, in, fv);
                STAGE_CHECK("preprocess_batch", TGS_STAGE_PREPROCESS_FWD);
            }
            if (n_streams > 1) {
                HIP_TRY(hipEventCreateWithFlags(&pre_done, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(pre_done, st0));
            }
        }
        for (int k = 0; k < nv; k++) {
            tgs_view_t& v = views[v0 + k];
            hipStream_t st = (hipStream_t)streams[(v0 + k) % n_streams];
            if (pre_done && st != st0) HIP_TRY(hipStreamWaitEvent(st, pre_done, 0));
            hipStream_t render_stream = t_render_streams.empty() ? nullptr : t_render_streams[(size_t)(v0 + k) % t_render_streams.size()];
            Opts opt = opt0;                                // the bounds of a view travel in its tgs_view_t
            opt.tile_bound = v.tile_bound > 0 ? v.tile_bound : 0;
            opt.heavy_bound = v.heavy_bound > 0 ? v.heavy_bound : 0; opt.mid_bound = v.mid_bound > 0 ? v.mid_bound : 0;
            const int64_t r = forward_impl(opt, (Meta*)v.host_meta, render_stream, nullptr, batched ? 1 : 0, r_capacity, nullptr, alloc_preset, &v, st, P, D, M, v.background, v.width, v.height, means3D, shs,
                                           v.colors_precomp ? v.colors_precomp : colors_precomp,
                                           opacities, scales, scale_modifier, rotations, cov3D_precomp, v.viewmatrix, v.projmatrix, v.campos, v.tan_fovx, v.tan_fovy,
                                           prefiltered, v.out_color, v.radii_out, 0);
            if (r < 0) { if (pre_done) (void)hipEventDestroy(pre_done); return (int)r; }
            v.R = r;
        }
        if (pre_done) (void)hipEventDestroy(pre_done);
    }
    return TGS_OK;
}

static int backward_render_impl(const Opts& opt, void* stream, int P, int64_t R, const float* background, int width, int height, const void* binning_buffer,
                                const void* img_buffer, const float* dL_dpix);

int tgs_backward_render_views(void* const* streams, int n_streams, int P, int n_views, const tgs_view_t* views)
{
    return tgs_backward_render_views_opt(nullptr, streams, n_streams, P, n_views, views);
}

int tgs_backward_render_views_opt(const tgs_options_t* o, void* const* streams, int n_streams, int P, int n_views, const tgs_view_t* views)
{
    if (n_views == 0) return TGS_OK;
    if (!streams || n_streams <= 0 || n_views < 0 || !views) return fail(TGS_ERR_INVALID, "bad arguments");
    const Opts opt0 = resolve_options(o, true);
    for (int k = 0; k < n_views; k++) {
        const tgs_view_t& v = views[k];
        Opts opt = opt0;
        opt.tile_bound = v.tile_bound > 0 ? v.tile_bound : 0;
        opt.mid_bound = v.mid_bound > 0 ? v.mid_bound : 0;
        const int r = backward_render_impl(opt, streams[k % n_streams], P, v.R, v.background, v.width, v.height, v.binning_buffer, v.img_buffer, v.dL_dpix);
        if (r < 0) return r;
    }
    return TGS_OK;
}

int tgs_backward_render(void* stream, int P, int64_t R, const float* background, int width, int height, const void* binning_buffer,
                        const void* img_buffer, const float* dL_dpix)
{
    return backward_render_impl(resolve_options(nullptr), stream, P, R, background, width, height, binning_buffer, img_buffer, dL_dpix);
}

int tgs_backward_render_opt(const tgs_options_t* o, void* stream, int P, int64_t R, const float* background, int width, int height, const void* binning_buffer,
                            const void* img_buffer, const float* dL_dpix)
{
    return backward_render_impl(resolve_options(o), stream, P, R, background, width, height, binning_buffer, img_buffer, dL_dpix);
}

static int backward_render_impl(const Opts& opt, void* stream, int P, int64_t R, const float* background, int width, int height, const void* binning_buffer,
                                const void* img_buffer, const float* dL_dpix)
{
    hipStream_t st = (hipStream_t)stream;
    g_err[0] = 0;
    const int debug = 0;
    if (P == 0) return TGS_OK;
    if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(TGS_ERR_INVALID, "bad sizes");
    if (!background || !binning_buffer || !img_buffer || !dL_dpix) return fail(TGS_ERR_INVALID, "NULL required pointer");
    const uint32_t gx = (uint32_t)((width + TILE - 1) / TILE), gy = (uint32_t)((height + TILE - 1) / TILE);
    ImgState s; BinState b;
    img_carve(s, (char*)img_buffer, (size_t)width * height, (size_t)gx * gy);
    bin_carve(b, (char*)binning_buffer, (size_t)R);
    if (R > 0) {
        STAGE_BEGIN(TGS_STAGE_RENDER_BWD);
        launch_render_bwd(st, s, b, width, height, gx, bounded_tiles(opt, (size_t)gx * gy), background, dL_dpix, opt.deterministic, bounded_mid(opt, (size_t)gx * gy), opt.light, gx * gy);
        STAGE_CHECK("render_bwd", TGS_STAGE_RENDER_BWD);
    }
    return TGS_OK;
}

int tgs_backward_batch(void* stream, int P, int D, int M, int n_views, const tgs_view_t* views, const float* means3D, const float* shs,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp, float* dL_dopacity,
                       float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int accumulate)
{
    return tgs_backward_batch_range(stream, P, D, M, n_views, views, means3D, shs, scales, scale_modifier, rotations, cov3D_precomp, dL_dopacity, dL_dmean3D,
                                    dL_dcov3D, dL_dsh, dL_dscale, dL_drot, accumulate, 0, P);
}

int tgs_backward_batch_range(void* stream, int P, int D, int M, int n_views, const tgs_view_t* views, const float* means3D, const float* shs,
                             const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp, float* dL_dopacity,
                             float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int accumulate, int first, int count)
{
    return tgs_backward_batch_range_planes(stream, P, D, M, n_views, views, means3D, shs, scales, scale_modifier, rotations, cov3D_precomp, dL_dopacity, dL_dmean3D,
                                           dL_dcov3D, dL_dsh, dL_dscale, dL_drot, accumulate, first, count, 0);
}

int tgs_backward_batch_range_planes(void* stream, int P, int D, int M, int n_views, const tgs_view_t* views, const float* means3D, const float* shs,
                                    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp, float* dL_dopacity,
                                    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int accumulate, int first, int count,
                                    int64_t dsh_plane_stride)
{
    hipStream_t st = (hipStream_t)stream;
    g_err[0] = 0;
    const int debug = 0;
    if (P == 0 || n_views == 0 || count == 0) return TGS_OK;
    if (P < 0 || n_views < 0 || !views) return fail(TGS_ERR_INVALID, "bad sizes");
    if (first < 0 || count < 0 || first % PRE_BLOCK != 0 || (long long)first + count > P || ((first + count) % PRE_BLOCK != 0 && first + count != P))
        return fail(TGS_ERR_INVALID, "Gaussian range [%d, %d + %d) must start and end on multiples of %d (or end at P = %d)", first, first, count, PRE_BLOCK, P);
    const bool has_sh = shs != nullptr, has_sr = scales != nullptr && rotations != nullptr;
    if (has_sr == (cov3D_precomp != nullptr)) return fail(TGS_ERR_INVALID, "provide exactly one of (scales, rotations) / cov3D_precomp");
    if (has_sh && (D < 0 || D > 3 || M < (D + 1) * (D + 1))) return fail(TGS_ERR_INVALID, "SH degree %d needs M >= %d (M=%d)", D, (D + 1) * (D + 1), M);
    if (!means3D || !dL_dopacity || !dL_dmean3D || (has_sh && !dL_dsh) || (has_sr && (!dL_dscale || !dL_drot)) || (!has_sr && !dL_dcov3D))
        return fail(TGS_ERR_INVALID, "NULL required pointer");
    if (dsh_plane_stride != 0 && (!has_sh || M != 16 || dsh_plane_stride < 3 * (int64_t)P || dsh_plane_stride % 4 != 0 || ((uintptr_t)dL_dsh & 15u) != 0))
        return fail(TGS_ERR_INVALID, "level-major dL_dsh needs SH colours with M = 16, a plane stride >= 3 P that is a multiple of 4 floats, and a 16-byte aligned dL_dsh");
    BwdIn in;
    memset(&in, 0, sizeof(in));
    in.P = P; in.D = D; in.M = M; in.means3D = means3D; in.shs = shs; in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp;
    in.dL_dopacity = dL_dopacity; in.dL_dmean3D = dL_dmean3D; in.dL_dcov3D = has_sr ? nullptr : dL_dcov3D; in.dL_dsh = dL_dsh;
    in.dL_dscale = has_sr ? dL_dscale : nullptr; in.dL_drot = has_sr ? dL_drot : nullptr;
    in.block0 = first / PRE_BLOCK; in.nblocks = (int)n_blocks((size_t)count);
    in.dsh_plane = (long long)dsh_plane_stride;
    for (int v0 = 0; v0 < n_views; v0 += BATCH_VIEWS) {
        BatchViews bv;
        memset(&bv, 0, sizeof(bv));
        bv.n = n_views - v0 < BATCH_VIEWS ? n_views - v0 : BATCH_VIEWS;
        for (int k = 0; k < bv.n; k++) {
            const tgs_view_t& w = views[v0 + k];
            if (w.width <= 0 || w.height <= 0 || w.R < 0 || !w.viewmatrix || !w.projmatrix || !w.campos || !w.radii || !w.geom_buffer ||
                !w.binning_buffer || !w.img_buffer || !w.dL_dmean2D || (!has_sh && !w.dL_dcolor))
                return fail(TGS_ERR_INVALID, "view %d: bad sizes or NULL required pointer", v0 + k);
            BatchView& o = bv.v[k];
            o.cam = make_cam(w.viewmatrix, w.projmatrix, w.campos, w.tan_fovx, w.tan_fovy, scale_modifier, w.width, w.height);
            ImgState s;
            geom_carve(o.g, (char*)w.geom_buffer, (size_t)P, has_sh, has_sr);
            img_carve(s, (char*)w.img_buffer, (size_t)w.width * w.height, (size_t)o.cam.gx * o.cam.gy);
            bin_carve(o.b, (char*)w.binning_buffer, (size_t)w.R);
            o.meta = s.meta; o.radii = w.radii; o.dL_dmean2D = w.dL_dmean2D; o.dL_dcolor = has_sh ? nullptr : w.dL_dcolor;
        }
        in.accumulate = (accumulate || v0 > 0) ? 1 : 0;       // later chunks add to what the first one stored
        STAGE_BEGIN(TGS_STAGE_PREPROCESS_BWD);
        launch_preprocess_bwd_batch(st, in, bv);
        STAGE_CHECK("preprocess_bwd_batch", TGS_STAGE_PREPROCESS_BWD);
    }
    return TGS_OK;
}

int tgs_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present)
{
    hipStream_t st = (hipStream_t)stream;
    g_err[0] = 0;
    (void)projmatrix;       // the reference's x/y frustum test is commented out (auxiliary.h:154)
    if (P == 0) return TGS_OK;
    if (P < 0 || !means3D || !viewmatrix || !present) return fail(TGS_ERR_INVALID, "bad arguments");
    launch_mark_visible(st, P, means3D, viewmatrix, present);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TGS_ERR_HIP, "mark_visible: %s", hipGetErrorString(e));
    return TGS_OK;
}

int64_t tgs_state_field(void* stream, const char* field, int P, int width, int height, int64_t R, int has_sh, int has_scale_rot,
                        const void* geom_buffer, const void* binning_buffer, const void* img_buffer, void* dst, size_t dst_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    g_err[0] = 0;
    const uint32_t gx = (uint32_t)((width + TILE - 1) / TILE), gy = (uint32_t)((height + TILE - 1) / TILE);
    const size_t N = (size_t)width * height, T = (size_t)gx * gy;
    GeomState g; ImgState s; BinState b;
    geom_carve(g, (char*)geom_buffer, (size_t)P, has_sh != 0, has_scale_rot != 0);
    img_carve(s, (char*)img_buffer, N, T);
    bin_carve(b, (char*)binning_buffer, (size_t)R);
    const void* src = nullptr; size_t count = 0, esz = 4, stride = 0, rows = 0;   // stride != 0: `rows` rows of `esz` bytes, `stride` apart
    if (!strcmp(field, "n_contrib")) { src = s.n_contrib; count = N; }
    else if (!strcmp(field, "final_T")) { src = s.final_T; count = N; }
    else if (!strcmp(field, "ranges")) { src = s.ranges; count = 2 * T; }
    else if (!strcmp(field, "means2D")) { src = g.pack; count = 2 * (size_t)P; esz = 8; stride = 64; rows = (size_t)P; }
    else if (!strcmp(field, "depths")) { src = g.depth; count = (size_t)P; }
    else if (!strcmp(field, "conic_opacity")) { src = (const char*)g.pack + 8; count = 4 * (size_t)P; esz = 16; stride = 64; rows = (size_t)P; }
    else if (!strcmp(field, "rgb")) { src = (const char*)g.pack + 24; count = 3 * (size_t)P; esz = 12; stride = 64; rows = (size_t)P; }
    else if (!strcmp(field, "tiles_touched")) { src = g.tiles_touched; count = (size_t)P; }
    else if (!strcmp(field, "tile_order")) { src = s.tile_order; count = T; }
    else if (!strcmp(field, "stamps")) { src = s.stamps; count = 8 * T; esz = 8; }
    else if (!strcmp(field, "block_masks")) { src = (const char*)b.recC + 4; count = (size_t)R; esz = 4; stride = 8; rows = (size_t)R; }   // 16-bit culling mask per sorted instance
    else if (!strcmp(field, "quad_masks")) { src = b.qmask; count = (size_t)R; esz = 8; }                               // 64-bit quadrant mask per sorted instance
    else if (!strcmp(field, "point_list")) { src = b.keys; count = (size_t)R; esz = 4; stride = 8; rows = (size_t)R; }   // low 32 bits of each sorted key
    else return fail(TGS_ERR_INVALID, "unknown field %s", field);
    const size_t total = stride ? rows * esz : count * esz;
    if (dst_bytes < total) return fail(TGS_ERR_INVALID, "dst too small for %s", field);
    if (count == 0) return 0;
    if (stride) HIP_TRY(hipMemcpy2DAsync(dst, esz, src, stride, esz, rows, hipMemcpyDeviceToDevice, st));
    else HIP_TRY(hipMemcpyAsync(dst, src, total, hipMemcpyDeviceToDevice, st));
    return (int64_t)count;
}

}  // extern "C"
