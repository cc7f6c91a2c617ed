import sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from youreditableavatar_amd import scenes, multiview
from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch
from diff_gaussian_rasterization import GaussianRasterizationSettings, _C
dev = torch.device("cuda", 0)
cfg = scenes.CONFIGS[3]; P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
cloud = scenes.make_cloud(P, D, cfg["seed"])
g = lambda x, rg=False: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev).requires_grad_(rg)
means3D, opac, scales, rots, shs = g(cloud["means3D"], True), g(cloud["opacities"], True), g(cloud["scales"], True), g(cloud["rotations"], True), g(cloud["shs"], True)
flat = FlatGradients([means3D, opac, scales, rots, shs])
dL = g(scenes.upstream_gradient(W, H, seed=1))
V = 8
settings = []
for k in range(V):
    c = scenes.orbit_camera(W, H, azimuth_deg=k * 360.0 / 64)
    settings.append(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=g(c.bg), scale_modifier=1.0,
        viewmatrix=g(c.viewmatrix), projmatrix=g(c.projmatrix), sh_degree=D, campos=g(c.campos), prefiltered=False, debug=False))
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = SyncFreeBatch(streams=ns)
marks = []
def up(images):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)      # forwards joined here
    return dL
of, ob = _C.backward_render_views, _C.backward_batch_raw
def brv(*a):
    r = of(*a); return r
def bbr(stream, *a):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)      # render_bwd joined
    return ob(stream, *a)
_C.backward_batch_raw = bbr
for it in range(5):
    flat.zero_(); marks.clear()
    s = torch.cuda.Event(enable_timing=True); s.record()
    batch.run_views(settings, means3D, opac, shs, scales, rots, up)
    e = torch.cuda.Event(enable_timing=True); e.record(); torch.cuda.synchronize()
    if len(marks) == 2:
        print(f"streams {ns}: forwards {s.elapsed_time(marks[0])/V:.3f}  render_bwd {marks[0].elapsed_time(marks[1])/V:.3f}  pergauss {marks[1].elapsed_time(e)/V:.3f}  total {s.elapsed_time(e)/V:.3f} ms/frame")
