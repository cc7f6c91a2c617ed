import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from youreditableavatar_amd import scenes
from diff_gaussian_rasterization import _C
dev = torch.device("cuda", 0)
cfg = scenes.CONFIGS[3]; P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
cloud = scenes.make_cloud(P, D, cfg["seed"], scale_mult=float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
g = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
e = torch.Tensor([])
c = scenes.orbit_camera(W, H, azimuth_deg=0.0)
args = (g(c.bg), g(cloud["means3D"]), e, g(cloud["opacities"]), g(cloud["scales"]), g(cloud["rotations"]), 1.0, e, g(c.viewmatrix), g(c.projmatrix), c.tanfovx, c.tanfovy, H, W, g(cloud["shs"]), D, g(c.campos), False, False)
for i in range(3): _C.rasterize_gaussians(*args)
_C.profile_begin(4096)
for i in range(10): _C.rasterize_gaussians(*args)
torch.cuda.synchronize()
pr = _C.profile_end()
print({k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in pr.items() if n})
