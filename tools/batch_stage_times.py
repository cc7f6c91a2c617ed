"""Per-stage GPU times of the 8-view BATCH step (SyncFreeBatch.run_views: the headline path) of config 3, each kernel alone on the GPU (one stream, events around
every stage), and the step's time with four streams:   python tools/batch_stage_times.py [scale_mult] [morton 0|1]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from youreditableavatar_amd import scenes
from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch
from diff_gaussian_rasterization import GaussianRasterizationSettings, _C
dev = torch.device("cuda", 0)
cfg = scenes.CONFIGS[3]; P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
sm = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cloud = scenes.make_cloud(P, D, cfg["seed"], scale_mult=sm)
if len(sys.argv) > 2 and sys.argv[2] == "1":
    cloud = scenes.morton_order(cloud)
g = lambda x, rg=False: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev).requires_grad_(rg)
L = {k: g(cloud[k], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
FlatGradients([L[k] for k in ("means3D", "opacities", "scales", "rotations", "shs")])
S = []
for k in range(8):
    c = scenes.orbit_camera(W, H, azimuth_deg=(k * 137.5) % 360.0)
    S.append(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=g(c.bg), scale_modifier=1.0, viewmatrix=g(c.viewmatrix),
                                           projmatrix=g(c.projmatrix), sh_degree=D, campos=g(c.campos), prefiltered=False, debug=False))
dL = g(scenes.upstream_gradient(W, H, seed=4321))
b = SyncFreeBatch(streams=4)
step = lambda: b.run_views(S, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], None, accumulate=False, upstream_view=lambda v, image: dL)
for _ in range(4):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 10 * 1e3)
b.streams = 1
step(); torch.cuda.synchronize()
_C.profile_begin(64 * 8)
step(); step()
torch.cuda.synchronize()
pr = _C.profile_end()
t = {k: round(ms / 16 * 1e3, 1) for k, (ms, n) in pr.items() if n}
print(f"scale x{sm} morton {sys.argv[2] if len(sys.argv) > 2 else 0}: step {sorted(ts)[1]:.4f} ms ({sorted(ts)[1] / 8:.4f} per frame, four streams), rerendered {b.rejected}; us per frame and stage, alone: {t}, sum {sum(t.values()):.0f}")
