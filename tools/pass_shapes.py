"""Lane-slot accounting of k_render_bwd's mappings on config 3, view 0 -- on the CPU, from the oracle's state (test infrastructure: this
tool is a measurement aid, not product code).

A "slot" is one (pixel, entry) pair evaluated by one lane.  Compared:
  quad4   : today -- a DPP row = one 2x2 quadrant x 4 consecutive entries of its own list; the block list is cut into chunks of 64 / 128
            entries, the wave runs max_q ceil(n_q / 4) passes per chunk, 64 slots per pass
  row16   : a DPP row = 16 consecutive entries of the quadrant's list, the row loops over the quadrant's 4 pixels; the wave runs
            max_q ceil(n_q / 16) chunk-steps per round, 4 x 64 slots each
Lists are bounded by the quadrant's deepest last contributor (backward.cu:487), rounds of `BCH` list positions, back to front.
    python tools/pass_shapes.py [BCH]
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from youreditableavatar_amd import scenes
from oracle import oracle

BCH = int(sys.argv[1]) if len(sys.argv) > 1 else 384
cfg = scenes.CONFIGS[3]; P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
cloud = scenes.config_cloud(3)
cam = scenes.orbit_camera(W, H, azimuth_deg=0.0)
color, radii, st = oracle.forward(bg=cam.bg, means3D=cloud["means3D"], opacities=cloud["opacities"], viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
                                  campos=cam.campos, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, image_height=H, image_width=W, sh_degree=D, shs=cloud["shs"],
                                  scales=cloud["scales"], rotations=cloud["rotations"])
t = lambda n, dt=None: torch.from_numpy(st.field(n).astype(dt) if dt else st.field(n))
pl = t("point_list", np.int64); rg = t("ranges", np.int64).view(-1, 2); m2 = t("means2D").view(-1, 2); co = t("conic_opacity").view(-1, 4)
nc = t("n_contrib", np.int64).view(H, W)
R = pl.numel(); gx = (W + 15) // 16; T = rg.shape[0]
lens = rg[:, 1] - rg[:, 0]
tile_of = torch.repeat_interleave(torch.arange(T), lens)
pos = torch.arange(R) - rg[tile_of, 0]
px = torch.arange(16, dtype=torch.float32)
Hp, Wp = (H + 15) // 16 * 16, gx * 16
ncp = torch.zeros(Hp, Wp, dtype=torch.long); ncp[:H, :W] = nc
tq = ncp.view(Hp // 16, 16, gx, 16).permute(0, 2, 1, 3).reshape(T, 16, 16)                      # [T, y, x]
quad_max = tq.view(T, 4, 2, 2, 4, 2, 2).amax(dim=(3, 6)).permute(0, 1, 3, 2, 4).reshape(T, 64)   # [T, block*4 + quadrant]
tile_qmax = tq.reshape(T, -1).amax(dim=1)

AQ = torch.zeros(R, 64, dtype=torch.bool)          # entry reaches the quadrant (alpha >= 1/255 on one of its pixels) AND lies in front of its last contributor
alive_pairs = 0; valid_pairs = 0
CH = 1 << 17
for s in range(0, R, CH):
    ids, tt, pp = pl[s:s + CH], tile_of[s:s + CH], pos[s:s + CH]
    x0, y0 = ((tt % gx) * 16).float(), ((tt // gx) * 16).float()
    dx = m2[ids, 0, None, None] - (x0[:, None, None] + px[None, None, :])
    dy = m2[ids, 1, None, None] - (y0[:, None, None] + px[None, :, None])
    q = co[ids]
    power = -0.5 * (q[:, 0, None, None] * dx * dx + q[:, 2, None, None] * dy * dy) - q[:, 1, None, None] * dx * dy
    alive = (power <= 0) & (torch.clamp(q[:, 3, None, None] * torch.exp(power), max=0.99) >= 1.0 / 255.0)       # [n, y, x]
    alive_pairs += int(alive.sum())
    valid = alive & (pp[:, None, None] < tq[tt])
    valid_pairs += int(valid.sum())
    aq = alive.view(-1, 4, 2, 2, 4, 2, 2).any(dim=6).any(dim=3).permute(0, 1, 3, 2, 4).reshape(-1, 64)
    AQ[s:s + CH] = aq & (pp[:, None] < quad_max[tt])
print(f"instances {R}, alive pairs {alive_pairs}, valid in the backward (in front of the pixel's last contributor) {valid_pairs}")
print(f"(quadrant, entry) pairs in the bounded lists: {int(AQ.sum())}  -> 4 slots each: {4 * int(AQ.sum())}; valid / slots = {valid_pairs / (4 * int(AQ.sum())):.3f}")

# rounds back to front from the tile's qmax: slot t of round r = position qmax - 1 - (r * BCH + t)
back = tile_qmax[tile_of] - 1 - pos
inr = back >= 0
rnd = torch.where(inr, back // BCH, torch.zeros_like(back))
AB = AQ.view(R, 16, 4).any(dim=2)                                                    # [R, block]
seg = tile_of * 64 + rnd                                                             # (tile, round)
assert int(rnd.max()) < 64
order = torch.argsort(seg * (1 << 20) + torch.where(inr, back % BCH, torch.zeros_like(back)), stable=True)    # walk order inside a round
seg_o = seg[order]; AB_o = AB[order] & inr[order, None]; AQ_o = AQ[order].view(R, 16, 4) & inr[order, None, None]
segstart = torch.ones(R, dtype=torch.bool); segstart[1:] = seg_o[1:] != seg_o[:-1]
seg_id = torch.cumsum(segstart.long(), 0) - 1
nseg = int(seg_id.max()) + 1
cs = torch.cumsum(AB_o.long(), dim=0)
start_idx = torch.nonzero(segstart).squeeze(1)
base_vals = cs[start_idx] - AB_o[start_idx].long()
rank = cs - AB_o.long() - base_vals[seg_id]                                           # rank in the block's list of the round
slots_valid = valid_pairs
for ch in (64, 128, 1 << 20):
    chunk = rank // ch
    nchunk = int(chunk.max()) + 1
    key = (seg_id[:, None] * 16 + torch.arange(16)[None, :]) * nchunk + chunk
    tot = torch.zeros(nseg * 16 * nchunk, 4, dtype=torch.long)
    for qd in range(4):
        m = AQ_o[:, :, qd]
        tot[:, qd].index_add_(0, key[m], torch.ones(int(m.sum()), dtype=torch.long))
    p4 = ((tot + 3) // 4).amax(dim=1)
    p16 = ((tot + 15) // 16).amax(dim=1)
    p8 = ((tot + 7) // 8).amax(dim=1)
    name = "whole round" if ch > 4096 else f"{ch}-entry chunks"
    print(f"BCH {BCH}, {name}: quad4 passes {int(p4.sum())} = {64 * int(p4.sum()) / 1e6:.1f} M slots (valid {slots_valid / (64 * int(p4.sum())):.3f}); "
          f"row16 chunk-steps {int(p16.sum())} = {256 * int(p16.sum()) / 1e6:.1f} M slots (valid {slots_valid / (256 * int(p16.sum())):.3f}); "
          f"row8 (2 quadrant rows of 8? n/a) steps {int(p8.sum())}")
    # per-wave totals -> in-tile imbalance (slowest wave / mean wave) for the two mappings
    w4 = p4.view(nseg, 16, nchunk).sum(dim=2); w16 = p16.view(nseg, 16, nchunk).sum(dim=2)
    for nm, w, cost in (("quad4", w4, 1.0), ("row16", w16, 4.0)):
        tot_w = w.sum(dim=1).float(); mx = w.amax(dim=1).float()
        print(f"    {nm}: sum over (tile, round) of slowest wave x 16 = {int((mx * 16).sum() * cost)} pass-equivalents vs sum of all waves {int(tot_w.sum() * cost)}  (balance {float(tot_w.sum() / (mx * 16).sum()):.3f})")
ideal = int(AQ_o.sum())
print(f"ideal (every row always busy, 4 slots per (quadrant, entry)): {4 * ideal / 1e6:.1f} M slots")
