"""The reference trainers' per-step protocol at 2048 x 2048 on the config-3 cloud (what bench.py reports as secondary.trainer_protocol), as a
stand-alone program for `rocprofv3 --kernel-trace --stats -- python3 tools/trainer_protocol.py [sh_degree] [steps]`:
bindings.gaussian_bind (sigmoid / exp / normalize of the raw parameters) + sh_color.points_rgb_dc_rest (the model's dc / rest parameters) -> GaussianRasterizer(colors_precomp) -> l1_ssim_loss -> backward, one view per step, everything through autograd
(tetgs_texture/paint_2dgs.py:159-166, tetgs_scene/tetgs_model.py:524-537,605-614, refine.py:245-247)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from youreditableavatar_amd import scenes
from youreditableavatar_amd.loss import l1_ssim_loss
from youreditableavatar_amd.sh_color import points_rgb_dc_rest
from youreditableavatar_amd.bindings import gaussian_bind
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
cloud = scenes.config_cloud(3)
P, W, H = cloud["means3D"].shape[0], 2048, 2048
g = lambda x, rg=False: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev).requires_grad_(rg)
# the model's raw parameters (tetgs_model.py:196-229): densities before the sigmoid, log-scales, unnormalised quaternions
op = np.clip(cloud["opacities"], 1e-4, 1 - 1e-4)
L = {"means3D": g(cloud["means3D"], True), "all_densities": g(np.log(op / (1 - op)), True), "_scales": g(np.log(cloud["scales"]), True),
     "_quaternions": g(cloud["rotations"], True)}
L["sh_dc"] = g(cloud["shs"][:, :1], True)               # tetgs_model.py:234-239: the two colour parameters of the model
if deg > 0:
    L["sh_rest"] = g(cloud["shs"][:, 1:], True)
S = []
for k in range(16):
    c = scenes.orbit_camera(W, H, azimuth_deg=(k * 137.5) % 360.0)
    S.append(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=g(c.bg), scale_modifier=1.0, viewmatrix=g(c.viewmatrix),
                                           projmatrix=g(c.projmatrix), sh_degree=deg, campos=g(c.campos), prefiltered=False, debug=False))
gt = torch.rand(3, H, W, device=dev)


def step(i):
    rs = S[i % len(S)]
    for t in L.values():
        t.grad = None
    colors = points_rgb_dc_rest(L["sh_dc"], L.get("sh_rest"), deg + 1, positions=L["means3D"], camera_centers=rs.campos)
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    opacities, scales, rotations, _ = gaussian_bind(L["all_densities"], L["_scales"], L["_quaternions"])     # strengths / scaling / quaternions of the model, one kernel
    img, _ = GaussianRasterizer(rs)(means3D=L["means3D"], means2D=m2, opacities=opacities, colors_precomp=colors, scales=scales, rotations=rotations)
    l1_ssim_loss(img, gt, 0.2).backward()


for i in range(10):
    step(i)
torch.cuda.synchronize()
if len(sys.argv) > 3:
    # diagnostic: python tools/trainer_protocol.py <deg> <steps per block> <blocks> -> ms per step of every block, the slowest step of each, and
    # the speculative forward's state behind it ([bound, consecutive misses, calls left in cool-down, tile bound, light tiles of the last frame])
    import youreditableavatar_amd.diff_gaussian_rasterization as dgr
    blocks, k = int(sys.argv[3]), 10
    if os.environ.get("TP_NOGC") == "1":
        import gc; gc.collect(); gc.disable()
    for b in range(blocks):
        ts = []
        for i in range(steps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); step(k); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); k += 1
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            step(k); k += 1
        torch.cuda.synchronize()
        print(f"block {b}: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms per step back to back; step by step: median {sorted(ts)[len(ts) // 2]:.4f}, max {max(ts):.4f}; state {list(dgr._speculation.state.values())}")
    sys.exit(0)
t0 = time.perf_counter()
for i in range(steps):
    step(10 + i)
torch.cuda.synchronize()
print(json.dumps({"trainer_protocol_ms_per_step": round((time.perf_counter() - t0) / steps * 1e3, 4), "sh_degree": deg, "gaussians": P, "image": [W, H]}))
