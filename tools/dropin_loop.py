"""One view per step through the unchanged reference API (GaussianRasterizer + autograd) at config 3 -- what bench.py reports as
secondary.dropin_api -- as a stand-alone program for `rocprofv3 --kernel-trace --stats -- python3 tools/dropin_loop.py [steps]`."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from youreditableavatar_amd import scenes
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
cfg = scenes.CONFIGS[3]; P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
cloud = scenes.config_cloud(3)
g = lambda x, rg=False: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev).requires_grad_(rg)
L = {k: g(cloud[k], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
S = []
for k in range(16):
    c = scenes.orbit_camera(W, H, azimuth_deg=(k * 137.5) % 360.0)
    S.append(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=g(c.bg), scale_modifier=1.0, viewmatrix=g(c.viewmatrix),
                                           projmatrix=g(c.projmatrix), sh_degree=D, campos=g(c.campos), prefiltered=False, debug=False))
dL = g(scenes.upstream_gradient(W, H, seed=4321))


def step(i):
    for t in L.values():
        t.grad = None                                       # optimizer.zero_grad(set_to_none=True) (refine.py:323)
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    img, _ = GaussianRasterizer(S[i % len(S)])(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
    img.backward(dL)


for i in range(20):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(20 + i)
torch.cuda.synchronize()
print(json.dumps({"dropin_ms_per_frame": round((time.perf_counter() - t0) / steps * 1e3, 4)}))
