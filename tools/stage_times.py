"""Per-stage GPU times (one stream, every kernel alone) of one forward + backward frame of config 3, view 0, through the one-view
entry points: python tools/stage_times.py [scale_mult] [opacity].  scale_mult > 1 grows the splats (more tiles per splat: the
wave-cooperative paths of k_scatter / slab_sum); opacity overrides the cloud's opacities (early termination)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from youreditableavatar_amd import scenes
from diff_gaussian_rasterization import _C
dev = torch.device("cuda", 0)
cfg = scenes.CONFIGS[3]; P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
sm = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cloud = scenes.make_cloud(P, D, cfg["seed"], scale_mult=sm)
if os.environ.get("STAGE_MORTON") == "1":            # the same cloud numbered along a 3-D Morton curve (mesh-bound Gaussians come spatially ordered)
    cloud = scenes.morton_order(cloud)
if len(sys.argv) > 2:
    cloud["opacities"] = np.full_like(cloud["opacities"], float(sys.argv[2]))
g = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
e = torch.Tensor([])
c = scenes.orbit_camera(W, H, azimuth_deg=0.0)
dL = g(scenes.upstream_gradient(W, H))
args = (g(c.bg), g(cloud["means3D"]), e, g(cloud["opacities"]), g(cloud["scales"]), g(cloud["rotations"]), 1.0, e, g(c.viewmatrix), g(c.projmatrix),
        c.tanfovx, c.tanfovy, H, W, g(cloud["shs"]), D, g(c.campos), False, False)
INFO = os.environ.get("STAGE_INFO") == "1"       # like the drop-in: the forward's class counts size the backward's grids (tile_bound / mid_bound)
def frame():
    kw = {}
    if INFO:
        R, color, radii, geom, binning, img, _, (tiles, mid) = _C.rasterize_gaussians(*args, info=True)
        kw = dict(tile_bound=tiles, mid_bound=max(mid, 1))
    else:
        R, color, radii, geom, binning, img = _C.rasterize_gaussians(*args)
    _C.rasterize_gaussians_backward(args[0], args[1], radii, e, args[4], args[5], 1.0, e, args[8], args[9], c.tanfovx, c.tanfovy, dL, args[14], D, args[16],
                                    geom, R, binning, img, False, **kw)
    return R, img, geom, binning
for i in range(3):
    frame()
_C.profile_begin(4096)
for i in range(10):
    R, img, geom, binning = frame()
torch.cuda.synchronize()
pr = _C.profile_end()
nc = _C.state_field("n_contrib", P, W, H, R, True, True, geom, binning, img).float().mean().item()
t = {k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in pr.items() if n}
print(f"scale x{sm}: R {R}, mean n_contrib {nc:.1f}, us per stage {t}, sum {sum(t.values()):.0f} us")
