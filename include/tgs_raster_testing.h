/*
 * tgs_raster_testing.h -- TEST-ONLY entry points of libtgs_raster.so.  NOT part of the drop-in boundary (include/tgs_raster.h): nothing here
 * has a counterpart in the reference's interface (cuda_rasterizer/rasterizer.h:20-85), and a caller that wants any of these behaviours
 * passes them per call in a tgs_options_t (tgs_raster.h, the *_opt entry points).  What is declared here are the process- / thread-wide
 * setters of rounds 1-2, kept as shims: they store DEFAULTS that a call made WITHOUT options falls back to, so that old test code keeps
 * running; plus one experiment knob and one read-out of the calling thread's last frame.  The library exports them; tests/ may use them.
 */
#ifndef TGS_RASTER_TESTING_H
#define TGS_RASTER_TESTING_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Test-only shim (process-wide, default on): a splat gets no tile instance in a tile of its 3-sigma rectangle where it stays below
 * alpha = 1/255 on every pixel (the reference creates the instance, rasterizer_impl.cu:98-109, and skips it pixel by pixel,
 * forward.cu:340-343).  Images and gradients do not change; num_rendered and the internal n_contrib (a list position) do.
 * Off = the reference's instance lists, e.g. to count fragments the way the reference's state defines them. */
void tgs_set_instance_pruning(int on);
/* Experiment knob (per calling thread): tgs_forward_views puts k_render_fwd of view k on streams[k mod n] -- behind an event on the
 * view's own stream -- so that binning (L2 atomics, latency) and compositing (VALU) of different views run on streams of their own.
 * n = 0 restores one stream per view. */
int tgs_set_render_streams(void* const* streams, int n);
/* Test-only shim, per calling thread: the same bound for the single-view sync-free entry points (tgs_forward_async, tgs_forward_speculative -- which
 * repeats the stages behind the scan with exact sizes when the guess was too small -- and tgs_backward / tgs_backward_render /
 * tgs_backward_accumulate of a frame KNOWN to have at most that many non-empty tiles).  0 (default): none.  It stays set until changed. */
void tgs_set_tile_bound(int64_t n_tiles_with_instances);
/* Non-empty tiles of the last frame this thread rendered with tgs_forward or tgs_forward_speculative (their Meta read-back); -1 if none. */
int64_t tgs_last_nonempty_tiles(void);
/* Test-only shim (process-wide): views per launch of the per-Gaussian forward stage inside tgs_forward_views (1..8, default 2).  Groups
 * read the SH rows once per group; measured with four streams, pairs pay (-2 % per frame) and larger groups do not (the views of a
 * group start their remaining stages together). */
void tgs_set_forward_group(int views_per_launch);
/* Test-only shim (process-wide): longest tile list that is depth-sorted inside LDS; longer lists take the
 * multi-workgroup global-memory path.  Power of two in [2, 8192]; default 8192. */
int tgs_set_sort_lds_cap(unsigned cap);
/* Test-only shim, process-wide switch for the backward render kernel: 1 = fixed summation order inside a tile (gradients
 * bitwise reproducible run to run, about 2.5x slower in that kernel), 0 = LDS float atomics inside a tile
 * (default), -1 = follow the environment variable TGS_DETERMINISTIC.  Neither mode uses global atomics. */
void tgs_set_deterministic(int on);

#ifdef __cplusplus
}
#endif
#endif /* TGS_RASTER_TESTING_H */
