"""The N > 1 step's collective path through the REAL backend at world size 1 (the pool has one GPU per box): backend "nccl" = RCCL, the
gradients of the 8-view step of config 3 all-reduced in 2 Gaussian ranges (bench.py's --grad-chunks default) behind the per-Gaussian pass (SyncFreeBatch.run_views(grad_chunks,
on_chunk) + FlatGradients.all_reduce_rows(even_alone=True)).  A sum over one rank moves no data between GPUs: this times RCCL's launches and
the stream choreography, not xGMI.  Stand-alone (bench.py runs it as a child process with a time limit and copies the JSON line into
secondary.rccl_world1, so that a problem in the collective library cannot take the bench line down).   python tools/rccl_world1.py [config]"""
import json, os, socket, sys, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, torch.distributed as dist
from youreditableavatar_amd import scenes
from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch
from diff_gaussian_rasterization import GaussianRasterizationSettings

cfg_i = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
cfg = scenes.CONFIGS[cfg_i]; P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
cloud = scenes.config_cloud(cfg_i)
g = lambda x, rg=False: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev).requires_grad_(rg)
L = {k: g(cloud[k], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
flat = FlatGradients([L[k] for k in ("means3D", "opacities", "scales", "rotations", "shs")], sh_params={4: 0})
S = []
for k in range(8):
    c = scenes.orbit_camera(W, H, azimuth_deg=(k * 137.5) % 360.0)
    S.append(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=g(c.bg), scale_modifier=1.0, viewmatrix=g(c.viewmatrix),
                                           projmatrix=g(c.projmatrix), sh_degree=D, campos=g(c.campos), prefiltered=False, debug=False))
dL = g(scenes.upstream_gradient(W, H, seed=4321))
batch = SyncFreeBatch(streams=4)
pend = []
S_full = S
# the same views at active SH degree 0 (the inpainting stage, paint_2dgs.py:61-63: 16 800 of the reference's ~22 800 rasterizer iterations):
# only the live coefficient of the stored [P,16,3] gradient is reduced -- 56 B per Gaussian instead of 236
S_deg0 = [s._replace(sh_degree=0) for s in S]
active = None


def step(reduce, chunks=2):
    batch.run_views(S, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], None, accumulate=False, upstream_view=lambda v, image: dL, grad_chunks=chunks,
                    on_chunk=(lambda first, count: pend.extend(flat.all_reduce_rows(first, count, even_alone=True, sh_degree=active))) if reduce else (lambda f, c: None))
    n = len(pend)
    for w in pend:
        w.wait()
    pend.clear()
    return n


def timed(reduce, n, warm, chunks=2):
    for _ in range(warm):
        step(reduce, chunks)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):                                      # median of three blocks
        t0 = time.perf_counter()
        for _ in range(n):
            handles = step(reduce, chunks)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[1], handles


ms_r, handles = timed(True, 10, 4)
ms_n, _ = timed(False, 10, 2)
# where the difference comes from: the same step with 1 / 2 / 4 Gaussian ranges, with and without the collectives (a sum over one rank moves no
# byte: what is left is RCCL's launch + the stream hand-over per collective, and the cost of cutting the per-Gaussian pass into ranges)
table = {}
for c in (1, 2, 4):
    table[str(c)] = {"with_reduce": round(timed(True, 8, 2, c)[0], 4), "without": round(timed(False, 8, 2, c)[0], 4)}
per_collective_us = round((table["4"]["with_reduce"] - table["4"]["without"]) / 4 * 1e3, 1)
cut2 = round(table["2"]["without"] / table["1"]["without"] - 1.0, 4)
ref = flat.flat.clone()
step(True)
torch.cuda.synchronize()
same = bool(torch.equal(ref, flat.flat))
# active degree 0: the live-row reduce
S, active = S_deg0, 0
d0_r, _ = timed(True, 10, 4)
d0_n, _ = timed(False, 10, 2)
ref0 = flat.flat.clone()
step(True)
torch.cuda.synchronize()
same0 = bool(torch.equal(ref0, flat.flat))
deg0 = {"ms_per_step_with_reduce": round(d0_r, 4), "ms_per_step_without": round(d0_n, 4), "bytes_reduced_per_step": flat.reduced_bytes(sh_degree=0),
        "bytes_of_the_whole_buffer": flat.reduced_bytes(), "gradients_unchanged_by_the_one_rank_sum": same0}
deg0["layout"] = "row-major [P,16,3]: the live coefficients are packed into a staging slice in front of the collective and unpacked behind it"
# the same with the SH gradients LEVEL-MAJOR (FlatGradients(level_major=True), round 6): the live coefficient is the leading plane of the parameter's
# region -- a slice of the flat buffer goes to the collective, nothing is staged
L_lm = {k: g(cloud[k], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
flat_lm = FlatGradients([L_lm[k] for k in ("means3D", "opacities", "scales", "rotations", "shs")], sh_params={4: 0}, level_major=True)
L, flat = L_lm, flat_lm
lm_r, _ = timed(True, 10, 4)
lm_n, _ = timed(False, 10, 2)
ref1 = flat.flat.clone()
step(True)
torch.cuda.synchronize()
deg0_lm = {"ms_per_step_with_reduce": round(lm_r, 4), "ms_per_step_without": round(lm_n, 4), "overhead_ms": round(lm_r - lm_n, 4),
           "gradients_unchanged_by_the_one_rank_sum": bool(torch.equal(ref1, flat.flat)), "staging_slices": len(flat.__dict__.get("_stage", {})),
           "sh_gradients_equal_the_row_major_run": bool(torch.equal(L_lm["shs"].grad.contiguous(), ref0[-P * 48:].view(P, 16, 3))),
           "layout": "level-major: 16 planes of 3 P floats; the collective takes the leading plane as it is"}
deg0["overhead_ms"] = round(d0_r - d0_n, 4)
# ... and the SH-3 headline step with the level-major layout (every plane live: the same 118 MB, written plane by plane)
S, active = S_full, None
lm3_n, _ = timed(False, 10, 2)
lm3_r, _ = timed(True, 10, 2)
full_lm = {"ms_per_step_with_reduce": round(lm3_r, 4), "ms_per_step_without": round(lm3_n, 4), "row_major_without": round(ms_n, 4)}
print(json.dumps({"ms_per_step_with_reduce": round(ms_r, 4), "ms_per_step_without": round(ms_n, 4), "backend": str(dist.get_backend()), "collective_handles_per_step": handles,
                  "gradients_unchanged_by_the_one_rank_sum": same, "frames_rerendered": batch.rejected, "active_sh_degree_0": deg0_lm, "active_sh_degree_0_row_major": deg0, "sh_degree_3_level_major": full_lm,
                  "ms_per_step_by_ranges": table, "rccl_cost_per_collective_us_at_world_1": per_collective_us,
                  "range_cutting_cost_frac": {"2": cut2, "4": round(table["4"]["without"] / table["1"]["without"] - 1.0, 4)},
                  "overhead_frac_of_the_default_two_ranges": round(table["2"]["with_reduce"] / table["1"]["without"] - 1.0, 4),
                  "what": f"8-view step of config {cfg_i} with its gradients all-reduced in 2 Gaussian ranges (the default of bench.py --grad-chunks) through RCCL at world size 1 (one coalesced collective per range, "
                          "overlapped with the per-Gaussian pass): the N > 1 control flow on the production backend; no data crosses xGMI.  ms_per_step_by_ranges separates "
                          "the cost of cutting the per-Gaussian pass into ranges (without) from RCCL's per-collective launch + stream hand-over (with - without)"}), flush=True)
dist.destroy_process_group()
