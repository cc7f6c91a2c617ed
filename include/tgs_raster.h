/*
 * tgs_raster.h -- C ABI of the MI355X-native differentiable 3D-Gaussian rasterizer
 * (libtgs_raster.so, built from youreditableavatar_amd/csrc by plain hipcc for gfx950).
 *
 * Drop-in boundary.  Each entry point replaces one static of the reference's
 * CudaRasterizer::Rasterizer (Edit_core/thirdparties/diff-gaussian-rasterization/
 * cuda_rasterizer/rasterizer.h:20-85), i.e. exactly what the reference's own FFI for this path
 * (rasterize_points.cu:35-217 behind ext.cpp:15-19) binds:
 *
 *   tgs_forward       <- Rasterizer::forward      rasterizer.h:33-58   (impl rasterizer_impl.cu:198-336)
 *   tgs_backward      <- Rasterizer::backward     rasterizer.h:60-85   (impl rasterizer_impl.cu:340-434)
 *   tgs_mark_visible  <- Rasterizer::markVisible  rasterizer.h:24-31   (impl rasterizer_impl.cu:141-153)
 *
 * Same argument meaning and order; the differences are the ones a C ABI forces:
 *   - the three std::function<char*(size_t)> allocators (rasterize_points.cu:27-33) become one
 *     callback  void* alloc(void* ctx, int which, size_t bytes)  returning a DEVICE pointer that
 *     stays valid until the matching backward; which = TGS_BUF_GEOM / _BINNING / _IMAGE;
 *   - an explicit hipStream_t (as void*): every kernel and copy is enqueued on it (the reference
 *     uses the legacy default stream, rasterizer_impl.cu:148,289);
 *   - errors are return codes (< 0) plus tgs_last_error() instead of C++ exceptions
 *     (std::runtime_error at rasterizer_impl.cu:242-245, auxiliary.h:166-173);
 *   - an absent input (the reference's nullptr: colors_precomp / shs / scales+rotations /
 *     cov3D_precomp, rasterizer_impl.cu:321,389,411) is a NULL pointer here as well.
 * All pointers are device pointers to contiguous fp32 (int32 for radii) unless noted.  The layout
 * inside the three state buffers is private to this library (the reference's is private too:
 * rasterizer_impl.h:29-65); they only have to be handed back to tgs_backward unmodified.
 * The render entry points keep no state between calls and are re-entrant (Rasterizer's statics are stateless too, rasterizer.h:20-85):
 * everything a backward needs travels in the three state buffers, and everything that tunes a call travels in an explicit
 * tgs_options_t (the *_opt entry points below; NULL = defaults) -- instance pruning, the deterministic backward, the LDS sort budget,
 * the forward group and the bounds on the tiles with instances.  Two threads that render through one library with different options
 * do not see each other's (tests/test_gpu_api.py::test_two_threads_with_different_options).
 * (The seven process- / thread-wide setters of rounds 1-2 -- tgs_set_sort_lds_cap / _instance_pruning / _forward_group / _deterministic /
 * _tile_bound / _render_streams and tgs_last_nonempty_tiles -- are NOT part of this boundary any more: they are declared in
 * include/tgs_raster_testing.h, test-only shims that store defaults for calls made WITHOUT options.)  Process-wide:
 * the optional bench profiler (tgs_profile_*) and two tuning variables of the environment, read once: TGS_BIN_WGS (binning chunks per
 * view, default 128) and TGS_FORWARD_GROUP.  Per calling thread: the message of tgs_last_error() and the pinned 64-byte staging
 * slot + event of the speculative forward (one per thread and device).
 */
#ifndef TGS_RASTER_H
#define TGS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGS_ABI_VERSION 3      /* 2: tgs_view_t carries the tile bounds + host_meta, tgs_options_t / *_opt entry points, dc/rest SH colours;
                                  3: tgs_options_t without side_stream (round 4's side-stream colours: measured slower twice, removed) */

enum { TGS_BUF_GEOM = 0, TGS_BUF_BINNING = 1, TGS_BUF_IMAGE = 2 };

enum {
    TGS_OK = 0,
    TGS_ERR_INVALID = -1,     /* bad argument (e.g. neither shs nor colors_precomp)            */
    TGS_ERR_HIP = -2,         /* a HIP call or kernel failed (debug=1 checks after every stage) */
    TGS_ERR_ALLOC = -3,       /* the allocation callback returned NULL                          */
    TGS_ERR_TOO_MANY = -4,    /* more than 2^31-1 tile instances                                */
    TGS_ERR_PREFILTERED = -5  /* prefiltered=1 but a Gaussian was culled (auxiliary.h:156-160)   */
};

typedef void* (*tgs_alloc_fn)(void* ctx, int which, size_t bytes);

/* Returns num_rendered (>= 0) or a negative TGS_ERR_*.  out_color[3,H,W], radii[P] (int32; may be
 * NULL like rasterizer.h:56).  D = active SH degree, M = SH coefficient triplets per Gaussian. */
int64_t tgs_forward(tgs_alloc_fn alloc, void* alloc_ctx, void* stream,
                    int P, int D, int M,
                    const float* background, int width, int height,
                    const float* means3D, const float* shs, const float* colors_precomp,
                    const float* opacities, const float* scales, float scale_modifier,
                    const float* rotations, const float* cov3D_precomp,
                    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                    float tan_fovx, float tan_fovy, int prefiltered,
                    float* out_color, int* radii, int debug);

/* Sync-free forward for callers that render many frames per step (multi-view batches).  Rasterizer::forward
 * reads num_rendered back to size the binning buffer (rasterizer_impl.cu:280-281: the GPU idles while the host
 * catches up); here the CALLER bounds it: the binning buffer is requested for r_capacity tile instances before
 * anything runs, nothing is read back and the call returns r_capacity -- pass that as R to tgs_backward[_accumulate]
 * and tgs_state_field.  A frame that needs more instances than r_capacity is REJECTED on the device (a tile list longer than the LDS
 * sort is not a reason: it is sorted in global memory by workgroups of the same launch): the kernels behind the scan do no work, out_color
 * is the background, dL_dmean2D is zero, the other gradient outputs are left untouched (nothing is accumulated), and
 * tgs_frame_status reports TGS_FRAME_REJECTED; render such a frame again with tgs_forward.  The prefiltered check
 * (TGS_ERR_PREFILTERED) also moves to tgs_frame_status (TGS_FRAME_PREFILTERED). */
int64_t tgs_forward_async(int64_t r_capacity, tgs_alloc_fn alloc, void* alloc_ctx, void* stream,
                          int P, int D, int M,
                          const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp,
                          const float* opacities, const float* scales, float scale_modifier,
                          const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                          float tan_fovx, float tan_fovy, int prefiltered,
                          float* out_color, int* radii, int debug);

/* Rasterizer::forward with the read-back taken off the critical path: the caller passes the instance count it expects (e.g. last
 * frame's, with headroom); the binning buffer is requested for r_guess instances, Meta is copied to the host right behind the scan,
 * every later stage is enqueued WITHOUT waiting for it, and only then the host waits for that copy.  If the frame does not fit, the
 * stages behind the scan (which returned at once) run again with the exact sizes, like tgs_forward -- the result is always the
 * complete frame.  *num_rendered = the true count; the RETURN value is what the binning buffer is carved for (r_guess, or the true
 * count after the retry): pass that as R to tgs_backward / tgs_state_field. */
int64_t tgs_forward_speculative(int64_t r_guess, int64_t* num_rendered, tgs_alloc_fn alloc, void* alloc_ctx, void* stream,
                                int P, int D, int M,
                                const float* background, int width, int height,
                                const float* means3D, const float* shs, const float* colors_precomp,
                                const float* opacities, const float* scales, float scale_modifier,
                                const float* rotations, const float* cov3D_precomp,
                                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                float tan_fovx, float tan_fovy, int prefiltered,
                                float* out_color, int* radii, int debug);

enum { TGS_FRAME_PREFILTERED = 1, TGS_FRAME_REJECTED = 2, TGS_FRAME_TILE_BOUND = 4 /* a backward ran with a tile bound below the frame's non-empty tiles: gradients incomplete */ };
/* Synchronises `stream` and returns the frame's true num_rendered and its TGS_FRAME_* flags.  The same 64 bytes sit at
 * the start of the image buffer (u64 num_rendered, u32 longest list, u32 overflow tiles, u32 flags, ...), so a batch
 * can also gather them on the device and read them back once. */
int tgs_frame_status(void* stream, const void* img_buffer, int64_t* num_rendered, int* flags);

/* ---- explicit per-call options (re-entrancy; ABI 2) ----
 * Every field's zero / -1 value means "the library default" (or, for the first four, whatever the test-only setter of the same name
 * last stored), so `tgs_options_t o = {sizeof o};` followed by the fields a caller cares about is the whole protocol. */
typedef struct {
    uint32_t struct_size;      /* sizeof(tgs_options_t) of the caller's build: fields beyond it read as defaults */
    int32_t instance_pruning;  /* 1: no instance in rectangle tiles the splat cannot reach with alpha >= 1/255; 0: the reference's lists; -1: default (on) */
    int32_t deterministic;     /* 1: bitwise-reproducible per-pixel backward (k_render_bwd_det); 0: default kernel; -1: default (TGS_DETERMINISTIC, else 0) */
    int32_t forward_group;     /* tgs_forward_views: views per launch of the per-Gaussian stage, 1..8; 0: default (2) */
    uint32_t sort_lds_cap;     /* longest tile list sorted inside LDS, power of two in [2, 8192]; 0: default (8192) */
    int64_t tile_bound;        /* sync-free / speculative forward and the per-pixel backward: upper bound on the tiles that hold instances
                                  (grids are sized by it; a forward with more is rejected / repeated; a backward with more sets
                                  TGS_FRAME_TILE_BOUND in the frame's flags and returns TGS_ERR_INVALID from tgs_frame_status); 0: none */
    int64_t heavy_bound, mid_bound;   /* the same for the tiles with >= 1024 / >= 128 instances (classes of the tile sort; the second also sizes the
                                  render kernels' grids when light_tiles is on); only read with tile_bound */
    int32_t light_tiles;       /* 1: tiles with fewer than 128 instances are set apart in the render kernels: the forward composites such a tile
                                  with ONE 256-thread workgroup (a longer list gets four, one per 8x8-pixel quarter), the backward three of them
                                  per 1024-thread workgroup (the others one per workgroup); 0: every tile with instances is treated alike;
                                  -1: default (1 in the *_views entry points, 0 in the single-view ones; TGS_LIGHT_TILES overrides both).
                                  Forward and backward of a frame may differ in this option: every forward records the light tiles'
                                  descriptors (k_scan), so a backward's grid is valid either way */
} tgs_options_t;
/* What a forward learned about its frame (filled when non-NULL; the synchronous and the speculative forward read the frame's Meta,
 * the sync-free one cannot: num_rendered / nonempty_tiles are -1 there). */
typedef struct {
    int64_t num_rendered;      /* true instance count of the frame */
    int64_t nonempty_tiles;    /* tiles that hold instances: the exact tile_bound for this frame's backward */
    int32_t flags;             /* TGS_FRAME_* */
    int32_t mid_tiles;         /* tiles with >= 128 instances: the exact mid_bound for this frame's backward (-1: unknown) */
} tgs_frame_info_t;
enum { TGS_FWD_SYNC = 0, TGS_FWD_ASYNC = 1, TGS_FWD_SPECULATIVE = 2 };
/* tgs_forward (mode TGS_FWD_SYNC; r ignored), tgs_forward_async (TGS_FWD_ASYNC; r = r_capacity) or tgs_forward_speculative
 * (TGS_FWD_SPECULATIVE; r = r_guess) with explicit options.  Returns what the binning buffer is carved for (pass it as R to the backward). */
int64_t tgs_forward_opt(const tgs_options_t* opt, int mode, int64_t r, tgs_frame_info_t* info,
                        tgs_alloc_fn alloc, void* alloc_ctx, void* stream,
                        int P, int D, int M,
                        const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                        float tan_fovx, float tan_fovy, int prefiltered,
                        float* out_color, int* radii, int debug);

/* Gradient outputs need NOT be zero-initialised (the reference requires torch::zeros,
 * rasterize_points.cu:151-159); every element is written.  dL_dconic[P,4] is scratch.
 * dL_dsh may be NULL when M == 0, dL_dscale / dL_drot may be NULL when scales == NULL. */
int tgs_backward(void* stream, int P, int D, int M, int64_t R,
                 const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                 const void* geom_buffer, const void* binning_buffer, const void* img_buffer,
                 const float* dL_dpix,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 int debug);

/* Multi-view batches (new capability, not in the reference): same as tgs_backward, but the PARAMETER gradients
 * (dL_dopacity, dL_dmean3D, dL_dsh, dL_dscale, dL_drot, and dL_dcolor / dL_dcov3D where they are inputs' gradients)
 * are ADDED to what the buffers hold, so a batch of views accumulates without a separate pass.  dL_dmean2D and
 * dL_dconic are still overwritten.  dL_dcolor may be NULL on the SH path, dL_dcov3D on the scale/rotation path. */
int tgs_backward_accumulate(void* stream, int P, int D, int M, int64_t R,
                            const float* background, int width, int height,
                            const float* means3D, const float* shs, const float* colors_precomp,
                            const float* scales, float scale_modifier, const float* rotations,
                            const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                            const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                            const void* geom_buffer, const void* binning_buffer, const void* img_buffer,
                            const float* dL_dpix,
                            float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                            float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                            int debug);

/* tgs_backward (accumulate = 0) / tgs_backward_accumulate (1) with explicit options (deterministic, tile_bound).
 * Round 6: the outputs that are intermediates of the reference's two-kernel backward and that its Python layer discards may be NULL here --
 * dL_dconic always, dL_dcolor when shs != NULL, dL_dcov3D when scales / rotations != NULL (backward.cu:399-557 -> :144-274 hands them from
 * kernel to kernel; __init__.py:137-152 returns them to inputs that are None): they are then not written -- 52 of the ~300 B the
 * per-Gaussian kernel stores per Gaussian.  tgs_backward / tgs_backward_accumulate keep the reference's contract (every output required). */
int tgs_backward_opt(const tgs_options_t* opt, int accumulate, void* stream, int P, int D, int M, int64_t R,
                     const float* background, int width, int height,
                     const float* means3D, const float* shs, const float* colors_precomp,
                     const float* scales, float scale_modifier, const float* rotations,
                     const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                     const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                     const void* geom_buffer, const void* binning_buffer, const void* img_buffer,
                     const float* dL_dpix,
                     float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                     float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                     int debug);

/* present[P]: 1 byte per Gaussian, 1 iff view-space z > 0.2 (auxiliary.h:154). */
int tgs_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* present);

/* Introspection of a finished forward pass (tests, bench): copies are enqueued on `stream`.
 * field: "n_contrib" (u32[H*W]), "final_T" (f32[H*W]), "ranges" (u32[2*T]), "point_list" (u32[R]),
 * "means2D" (f32[2P]), "depths" (f32[P]), "conic_opacity" (f32[4P]), "rgb" (f32[3P], SH path only),
 * "tiles_touched" (u32[P]), "block_masks" (u32[R]: 4x4 blocks of the instance's tile it can reach, bit 4*by + bx), "quad_masks"
 * (u64[R]: the same per 2x2-pixel quadrant, bit 8*row + column of the tile's 8x8 quadrant grid), "tile_order" (u32[T]).
 * dst is a device pointer with room for the whole field.
 * Returns the element count or a negative error. */
int64_t tgs_state_field(void* stream, const char* field, int P, int width, int height, int64_t R,
                        int has_sh, int has_scale_rot,
                        const void* geom_buffer, const void* binning_buffer, const void* img_buffer,
                        void* dst, size_t dst_bytes);

/* Bench instrumentation (process-global, not re-entrant): between begin and end every pipeline stage is
 * bracketed by two hipEvents recorded on the caller's stream -- no synchronisation is added.  end()
 * waits for the events and returns per-stage summed milliseconds and launch counts (arrays of
 * TGS_STAGE_COUNT).  Records beyond max_records are dropped. */
enum {
    TGS_STAGE_PREPROCESS_FWD = 0, TGS_STAGE_SCAN, TGS_STAGE_SCATTER, TGS_STAGE_TILE_SORT, TGS_STAGE_RENDER_FWD,
    TGS_STAGE_RENDER_BWD, TGS_STAGE_PREPROCESS_BWD, TGS_STAGE_COUNT
};
int tgs_profile_begin(int max_records);
int tgs_profile_end(double* ms_sum, int64_t* counts);
/* Restrict the events to the stages whose bit (1u << TGS_STAGE_*) is set; default all.  Events cost a few microseconds of
 * queue time each, so a throughput measurement keeps only the stage it reports. */
void tgs_profile_stages(unsigned mask);

/* ---- "next" row 1: the caller-side SH -> RGB of every training step, fused ----
 * Replaces TetGS.get_points_rgb (Edit_core/tetgs_scene/tetgs_model.py:413-442: F.normalize(positions - camera_center),
 * eval_sh of Edit_core/utils/spherical_harmonics.py:117-172, + 0.5, clamp_min 0) and its autograd.
 * sh[P,M,3] (M <= 16), levels = sh_levels (1..4, uses levels^2 coefficients).  Exactly one of
 * (positions[P,3] + camera_center[3]) / directions[P,3].  colors[P,3].  Backward writes every element of
 * dL_dsh[P,M,3] (zeros above the active levels), and dL_dpositions or dL_ddirections (either may be NULL). */
int tgs_sh_rgb_forward(void* stream, int P, int M, int levels, const float* sh, const float* positions,
                       const float* camera_center, const float* directions, float* colors);
int tgs_sh_rgb_backward(void* stream, int P, int M, int levels, const float* sh, const float* positions,
                        const float* camera_center, const float* directions, const float* dL_dcolors,
                        float* dL_dsh, float* dL_dpositions, float* dL_ddirections);

/* The same colours from the TWO parameter tensors the reference's models keep -- _sh_coordinates_dc [P,1,3] and _sh_coordinates_rest
 * [P,M_rest,3] (tetgs_model.py:234-239) -- without the concatenation that the `sh_coordinates` property performs in front of every get_points_rgb call
 * (:268-272) and the split of its gradient.  Only the levels^2 - 1 active rest rows are read; at levels == 1 (the inpainting stage)
 * sh_rest / dL_dsh_rest are not touched and may be NULL (the gradient of the rest rows is exactly zero there).  For levels > 1 the backward
 * writes every element of dL_dsh_dc[P,3] and dL_dsh_rest[P,M_rest,3] (zeros above the active levels). */
int tgs_sh_rgb_dcrest_forward(void* stream, int P, int M_rest, int levels, const float* sh_dc, const float* sh_rest, const float* positions,
                              const float* camera_center, const float* directions, float* colors);
int tgs_sh_rgb_dcrest_backward(void* stream, int P, int M_rest, int levels, const float* sh_dc, const float* sh_rest, const float* positions,
                               const float* camera_center, const float* directions, const float* dL_dcolors, float* dL_dsh_dc, float* dL_dsh_rest,
                               float* dL_dpositions, float* dL_ddirections);

/* ---- the binding side (BASELINE.json: "tetgs_scene Gaussian model bindings") ----
 * What the model classes do in front of every rasterizer call, each as a framework kernel of its own plus its backward
 * (Edit_core/tetgs_scene/tetgs_model.py): strengths :261-265 opacity = sigmoid(all_densities); scaling :279-281 scales = exp(_scales);
 * quaternions :283-286 quats = x / max(|x|, 1e-12); points :252-258 points = ori_points + normals * _points (one learnable offset per
 * mesh-bound Gaussian).  One kernel forward, one backward.  Every group is optional: pass NULL for its inputs AND outputs.
 * raw_density[P,1] raw_scales[P,3] raw_quats[P,4] ori_points[P,3] normals[P,3] deltas[P,1] -> opacity[P,1] scales[P,3] quats[P,4] points[P,3].
 * The backward takes the forward's OUTPUTS opacity / scales (sigmoid' and exp' from their values), the inputs raw_quats / normals, the
 * incoming gradients g_*, and writes the requested d_* (a NULL d_* is skipped). */
int tgs_bind_forward(void* stream, int P, const float* raw_density, const float* raw_scales, const float* raw_quats, const float* ori_points,
                     const float* normals, const float* deltas, float* opacity, float* scales, float* quats, float* points);
int tgs_bind_backward(void* stream, int P, const float* raw_quats, const float* normals, const float* opacity, const float* scales,
                      const float* g_opacity, const float* g_scales, const float* g_quats, const float* g_points,
                      float* d_density, float* d_scales, float* d_quats, float* d_deltas);

/* The same for the TWO-group models of the editing stages (EditTetGS, Edit_core/tetgs_scene/tetgs_edit_2d.py:280-318; Edit3DTetGS,
 * tetgs_edit_3d.py:272-331), whose properties concatenate [keep, edit] and then apply the activation on every access, with the keep group frozen
 * (requires_grad=False, tetgs_edit_2d.py:237-262).  The forward writes rows [0, Pk) from the keep group and rows [Pk, Pk + Pe) from the
 * edit group of opacity[Pk+Pe,1] scales[Pk+Pe,3] quats[Pk+Pe,4] points[Pk+Pe,3] in one launch (a NULL output is skipped).  Edit
 * positions: edit_points[Pe,3] (EditTetGS.points, tetgs_edit_2d.py:281-283), or -- edit_points NULL -- ori_edit_points[Pe,3] +
 * edit_normals[Pe,3] * edit_offsets[Pe,1] (Edit3DTetGS.points, tetgs_edit_3d.py:273-278).  The backward takes the FULL [Pk+Pe]
 * forward outputs and incoming gradients and writes gradients of the edit group only (Pe rows): d_points[Pe,3] for plain edit
 * positions or d_offsets[Pe,1] for normal-bound ones; a NULL d_* is skipped. */
int tgs_bind_groups_forward(void* stream, int Pk, int Pe, const float* keep_density, const float* keep_scales, const float* keep_quats, const float* keep_points,
                            const float* edit_density, const float* edit_scales, const float* edit_quats, const float* edit_points,
                            const float* ori_edit_points, const float* edit_normals, const float* edit_offsets,
                            float* opacity, float* scales, float* quats, float* points);
int tgs_bind_groups_backward(void* stream, int Pk, int Pe, const float* edit_quats, const float* edit_normals, const float* opacity, const float* scales,
                             const float* g_opacity, const float* g_scales, const float* g_quats, const float* g_points,
                             float* d_density, float* d_scales, float* d_quats, float* d_points, float* d_offsets);

/* ---- "next" row 4: simple-knn ----
 * distCUDA2 (Edit_core/thirdparties/simple-knn/spatial.cu:15-26 -> SimpleKNN::knn, simple_knn.cu:185-221):
 * mean_dist2[i] = mean of the 3 smallest squared distances from points[i] to the other points (FLT_MAX terms, i.e.
 * +inf, when fewer than 3 others exist, like the reference).  workspace: device scratch of
 * tgs_dist2_workspace_bytes(P) bytes. */
size_t tgs_dist2_workspace_bytes(int P);
int tgs_dist2(void* stream, int P, const float* points, float* mean_dist2, void* workspace, size_t workspace_bytes);
/* The K (<= 32) nearest neighbours of every point among the same points, itself included: what the reference asks of its
 * third-party knn_points(points[None], points[None], K) (tetgs_scene/tetgs_model.py:6 import; :36 with K = 4, :180 with K = 16).  dists[P,K] squared
 * distances ascending (FLT_MAX and index -1 where P < K), idx[P,K] int64.  Same workspace as tgs_dist2. */
int tgs_knn_self(void* stream, int P, int K, const float* points, float* dists, long long* idx, void* workspace, size_t workspace_bytes);

/* ---- Batched backward (multi-view steps) ----
 * Rasterizer::backward runs its per-Gaussian half (computeCov2DCUDA + preprocessCUDA of backward.cu) once per view; with
 * in-place accumulation that is 192 B of SH read plus 192 B of dL_dsh read AND written per Gaussian per view.  A batch
 * splits the backward: tgs_backward_render does the per-pixel half of one view (tile partials stay in that view's
 * binning buffer), and ONE tgs_backward_batch call then does the per-Gaussian half for all views, reading the
 * view-independent inputs once and storing (accumulate = 0) or adding (accumulate = 1) the summed parameter gradients
 * once.  Per-view outputs: dL_dmean2D[P,3] and, on the colors_precomp path (shs == NULL), dL_dcolor[P,3].
 * A view rejected by tgs_forward_async contributes nothing (its dL_dmean2D is zero). */
typedef struct {
    int width, height;
    float tan_fovx, tan_fovy;
    const float *viewmatrix, *projmatrix, *campos;
    const int* radii;
    const void *geom_buffer, *binning_buffer, *img_buffer;
    int64_t R;                 /* what tgs_forward / tgs_forward_async returned for this view */
    float* dL_dmean2D;
    float* dL_dcolor;          /* NULL on the SH path */
    /* used by the *_views entry points only (NULL / 0 elsewhere) */
    const float* background;   /* [3] */
    float* out_color;          /* [3,H,W] */
    int* radii_out;            /* [P], written by tgs_forward_views (the same memory `radii` points to) */
    const float* dL_dpix;      /* [3,H,W] */
    size_t geom_bytes, binning_bytes, img_bytes;   /* capacities of the three caller-allocated state buffers */
    const float* colors_precomp;  /* [P,3] colours of THIS view (tgs_forward_views; overrides the shared argument) or NULL */
    int64_t tile_bound;        /* *_views entry points: upper bound on the tiles that hold instances, or 0 (none): the sync-free grids are then
                                  sized for that many tiles instead of all of them (surplus workgroups of thousands of empty tiles cost ~4 % of a
                                  frame each way at config 3); a frame with MORE non-empty tiles is rejected like one that exceeds r_capacity */
    int64_t heavy_bound, mid_bound;   /* the same for the two upper classes of the tile sort: tiles with >= 1024 / >= 128 instances (the second
                                  includes the first); 0: none.  Only read when tile_bound is set. */
    void* host_meta;           /* tgs_forward_views: 64 bytes of pinned, device-visible host memory or NULL.  The scan kernel writes the frame's
                                  Meta record there itself (num_rendered at byte 0, longest list at 8, lists beyond the LDS sort at 12, flags at
                                  16, tiles with instances / with >= 1024 / with >= 128 at 20 / 24 / 28, 1 at byte 32 if a tile bound was
                                  exceeded: the frame is rejected): the caller's verdict needs no device-to-host copy in the stream */
} tgs_view_t;
/* Whole-batch entry points: one call enqueues the forward (or the per-pixel backward) of every view, view k on
 * streams[k % n_streams], with state buffers the CALLER allocated up front (tgs_state_sizes) -- no allocation callback, no
 * read-back, one trip through the host language per batch instead of several per view.  tgs_forward_views is
 * tgs_forward_async per view (SH path or shared colors_precomp); it fills views[k].R.  The caller orders the streams
 * against its own work (events) and checks every frame afterwards (tgs_frame_status / the Meta records). */
void tgs_state_sizes(int P, int width, int height, int has_sh, int has_scale_rot, int64_t r_capacity, size_t sizes3[3]);
int tgs_forward_views(void* const* streams, int n_streams, int64_t r_capacity, int P, int D, int M,
                      const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                      const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      int prefiltered, int n_views, tgs_view_t* views);
int tgs_backward_render_views(void* const* streams, int n_streams, int P, int n_views, const tgs_view_t* views);
/* The same with explicit options (instance pruning, forward group, sort budget, render_split / deterministic); the tile bounds of a
 * view travel in its tgs_view_t. */
int tgs_forward_views_opt(const tgs_options_t* opt, void* const* streams, int n_streams, int64_t r_capacity, int P, int D, int M,
                          const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                          const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                          int prefiltered, int n_views, tgs_view_t* views);
int tgs_backward_render_views_opt(const tgs_options_t* opt, void* const* streams, int n_streams, int P, int n_views, const tgs_view_t* views);
int tgs_backward_render_opt(const tgs_options_t* opt, void* stream, int P, int64_t R, const float* background, int width, int height,
                            const void* binning_buffer, const void* img_buffer, const float* dL_dpix);
/* sizeof(tgs_view_t) / sizeof(tgs_options_t) of THIS build: a binding checks them against its own declaration before it walks an array */
size_t tgs_sizeof_view(void);
size_t tgs_sizeof_options(void);

int tgs_backward_render(void* stream, int P, int64_t R, const float* background, int width, int height,
                        const void* binning_buffer, const void* img_buffer, const float* dL_dpix);
int tgs_backward_batch(void* stream, int P, int D, int M, int n_views, const tgs_view_t* views,
                       const float* means3D, const float* shs, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp,
                       float* dL_dopacity, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                       float* dL_dscale, float* dL_drot, int accumulate);
/* The same pass restricted to Gaussians [first, first + count): first and first + count multiples of 256 (or first + count == P).
 * A data-parallel step runs it range by range so that the all-reduce of one range's gradients (RCCL, on its own stream) overlaps
 * the pass over the next range; every output element of the range is final when the call's kernels are. */
int tgs_backward_batch_range(void* stream, int P, int D, int M, int n_views, const tgs_view_t* views,
                             const float* means3D, const float* shs, const float* scales, float scale_modifier,
                             const float* rotations, const float* cov3D_precomp,
                             float* dL_dopacity, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                             float* dL_dscale, float* dL_drot, int accumulate, int first, int count);

/* The same with dL_dsh LEVEL-MAJOR (round 6; dsh_plane_stride = 0: exactly tgs_backward_batch_range): coefficient k of Gaussian p is written to
 * dL_dsh[k * dsh_plane_stride + 3 p + c] (floats), dsh_plane_stride >= 3 P a multiple of 4, dL_dsh 16-byte aligned, M = 16.  Why: the gradients of
 * the (D + 1)^2 coefficients that are live at the step's active degree are then ONE contiguous piece of the caller's gradient buffer, which a
 * data-parallel step hands to the collective as it is (multiview.FlatGradients(level_major=True)); the reference's row-major [P, M, 3] layout
 * (rasterize_points.cu:157) interleaves live and dead coefficients Gaussian by Gaussian. */
int tgs_backward_batch_range_planes(void* stream, int P, int D, int M, int n_views, const tgs_view_t* views,
                                    const float* means3D, const float* shs, const float* scales, float scale_modifier,
                                    const float* rotations, const float* cov3D_precomp,
                                    float* dL_dopacity, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                                    float* dL_dscale, float* dL_drot, int accumulate, int first, int count, int64_t dsh_plane_stride);

/* ---- "next" row 2: the trainers' photometric loss ----
 * loss = (1 - dssim_factor) * l1_loss(img, gt) + dssim_factor * (1 - ssim(img, gt)), window 11, sigma 1.5, zero padding
 * (Edit_core/utils/loss_utils.py:17-18 and :39-63, composed as in tetgs_texture/refine.py:245-247), over
 * `planes` = batch x channels image planes of height x width.  out3[0] = loss, out3[1] = ssim, out3[2] = l1 (device
 * floats); dL_dimg (may be NULL) receives d loss / d img.  dssim_factor 1 / 0 give 1 - ssim / l1 and their gradients. */
size_t tgs_l1_ssim_workspace_bytes(int planes, int height, int width);
int tgs_l1_ssim(void* stream, int planes, int height, int width, const float* img, const float* gt, float dssim_factor,
                float* out3, float* dL_dimg, void* workspace, size_t workspace_bytes);
/* The gradient pass on its own, for an autograd backward: the SAME img / gt / dssim_factor / workspace a tgs_l1_ssim call (with dL_dimg NULL
 * or not) has filled, kept unmodified since; dL_dimg = upstream[0] * d loss / d img with `upstream` a DEVICE scalar (the gradient arriving
 * at the loss; NULL = 1) -- the chain rule is folded into the pass instead of a separate image-sized multiply. */
int tgs_l1_ssim_backward(void* stream, int planes, int height, int width, const float* img, const float* gt, float dssim_factor,
                         const float* upstream, float* dL_dimg, const void* workspace, size_t workspace_bytes);


/* Hardware self-test of the wave-level 36-value reduction used by the backward render kernel:
 * in[64][36] (one row per lane) -> out[4][9], out[e][k] = sum over lanes of in[lane][e*9+k]. */
int tgs_selftest_reduce36(void* stream, const float* in, float* out);

const char* tgs_last_error(void);   /* thread-local message of the last failing call */
int tgs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TGS_RASTER_H */
