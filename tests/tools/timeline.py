"""Diagnostic (needs a -DTGS_STAMPS=1 build): per-tile start/end stamps of the render kernels (forward: first quarter's start to the last
quarter's end), and the busy time of the backward's waves inside their workgroup."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from youreditableavatar_amd import scenes
from tests import util
from diff_gaussian_rasterization import _C
cloud, cams, dL = scenes.config_scene(3)
inp = util.scene_input(cloud, cams[0])
dev = torch.device('cuda:0')
t = lambda k: (torch.from_numpy(np.ascontiguousarray(inp[k], np.float32)).to(dev) if inp.get(k) is not None else torch.Tensor([]))
bg, means3D, opac, view, proj, campos = t("bg"), t("means3D"), t("opacities"), t("viewmatrix"), t("projmatrix"), t("campos")
sh, colors, scales, rots, cov = t("shs"), t("colors_precomp"), t("scales"), t("rotations"), t("cov3D_precomp")
H, W, D = 1080, 1920, 3
dLt = torch.from_numpy(dL).to(dev)
for it in range(3):
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(bg, means3D, colors, opac, scales, rots, 1.0, cov, view, proj, inp['tanfovx'], inp['tanfovy'], H, W, sh, D, campos, False, False)
    g = _C.rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rots, 1.0, cov, view, proj, inp['tanfovx'], inp['tanfovy'], dLt, sh, D, campos, geom, R, binning, img, False)
torch.cuda.synchronize()
P = means3D.shape[0]
st = _C.state_field('stamps', P, W, H, R, True, True, geom, binning, img).cpu().numpy().reshape(-1, 8)
rg = _C.state_field('ranges', P, W, H, R, True, True, geom, binning, img).cpu().numpy().reshape(-1, 2)
n = rg[:,1]-rg[:,0]
for name, a, b in (('fwd', 0, 1), ('bwd', 2, 3)):
    sel = n > 0
    t0, t1 = st[sel, a].astype(np.int64), st[sel, b].astype(np.int64)
    base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0   # us
    dur = t1 - t0
    nn = n[sel]
    print(f'{name}: kernel span {t1.max():.1f} us; last start {t0.max():.1f} us; sum of WG durations {dur.sum():.0f} us -> /512 workgroup slots (2 per CU) = {dur.sum()/512:.1f} us')
    order = np.argsort(-dur)[:6]
    print('  longest WGs: ', [(int(nn[i]), round(float(dur[i]),1), round(float(t0[i]),1)) for i in order], '(list len, dur us, start us)')
    # concurrency profile
    for q in (0.25, 0.5, 0.75, 0.9, 1.0):
        tt = t1.max() * q
        print(f'  t={tt:.0f}us running WGs: {int(((t0 <= tt) & (t1 > tt)).sum())}', end=';')
    print()
    # utilisation of the 16 wave slots inside a workgroup: sum of the waves' busy time (between a round's staging barrier and their next
    # barrier) against 16 x the workgroup's span; and against 16 x the slowest wave
    bs, bm = st[sel, 4 + 2 * (a // 2)].astype(np.float64) / 100.0, st[sel, 5 + 2 * (a // 2)].astype(np.float64) / 100.0
    for lo, hi in ((128, 512), (512, 1 << 30)):
        m = (nn >= lo) & (nn < hi) & (bm > 0)
        if m.any():
            print(f'  n in [{lo},{hi}): wave busy / (16 x WG span) = {bs[m].sum() / (16 * dur[m].sum()):.3f}; wave busy / (16 x slowest wave) = {bs[m].sum() / (16 * bm[m].sum()):.3f}; slowest wave / WG span = {bm[m].sum() / dur[m].sum():.3f}')
    nz = nn > 0
    print('  us per list entry for non-empty WGs: median', np.median(dur[nz]/nn[nz]), ' heavy(>1000):', np.median(dur[nn>1000]/nn[nn>1000]) if (nn>1000).any() else None)
    # duration vs list length: fixed cost per tile
    for lo, hi in ((1, 16), (16, 64), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 4096)):
        m = (nn >= lo) & (nn < hi)
        if m.any():
            print(f'  n in [{lo},{hi}): tiles {int(m.sum())}, median dur {np.median(dur[m]):.2f} us, total {dur[m].sum():.0f} us')
