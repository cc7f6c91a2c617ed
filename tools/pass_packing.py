"""Would another assignment of a tile's 64 quadrant lists to the rows of its waves need fewer passes?  (round 6; CPU, from the oracle's
state -- a measurement aid like pass_shapes.py, not product code.)

k_render_bwd today: wave w of a tile's workgroup owns block w; its four 16-lane rows walk the four quadrant lists of that block, four
entries per pass; the wave runs max over its rows of ceil(n_q / 4) passes per chunk, the workgroup's round lasts as long as its slowest
wave (the 16 waves meet at the round's barriers).  Two figures per mapping, summed over all (tile, round):
  slots            = 64 x wave passes                      (what the vector units issue)
  pass-equivalents = n_waves x the slowest wave's passes   (what the workgroup's wave slots are held for)
Mappings modelled (pixel state parked in LDS between rounds, so any row may walk any quadrant's list):
  today      block-bound rows, block list cut into chunks of 64 / 128 entries (pass_shapes.py's quad4)
  sorted     per round the 64 quadrants sorted by list length, wave j takes ranks 4j .. 4j+3 (its rows have similar lengths);
             lists per 64-slot group of the staged round ("grp64": what a wave can build from four arbitrary mask bits with the LDS
             it has) or whole-round lists ("round")
  lpt8       8 waves per tile, every row walks TWO quadrant lists one after the other (longest with shortest)
  pool2      two tiles (consecutive in tile_order) share one 16-wave workgroup: 128 lists on 64 rows, longest-processing-time first
    python tools/pass_packing.py [BCH] [forward]
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from youreditableavatar_amd import scenes
from oracle import oracle

BCH = int(sys.argv[1]) if len(sys.argv) > 1 else 384
FORWARD = len(sys.argv) > 2 and sys.argv[2] == "forward"      # forward: unbounded lists front to back, rounds of BCH from the list's start, pixels stop at their last contributor
SCALE = float(os.environ.get("PACK_SCALE", "1"))
cfg = scenes.CONFIGS[3]; P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
cloud = scenes.config_cloud(3) if SCALE == 1.0 else scenes.make_cloud(P, D, cfg["seed"], scale_mult=SCALE)      # PACK_SCALE=4: tools/stage_times.py 4's frame
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
tq = ncp.view(Hp // 16, 16, gx, 16).permute(0, 2, 1, 3).reshape(T, 16, 16)
quad_max = tq.view(T, 4, 2, 2, 4, 2, 2).amax(dim=(3, 6)).permute(0, 1, 3, 2, 4).reshape(T, 64)   # [T, block*4 + quadrant]
tile_qmax = tq.reshape(T, -1).amax(dim=1)

AQ = torch.zeros(R, 64, dtype=torch.bool)
CHK = 1 << 17
for s in range(0, R, CHK):
    ids, tt, pp = pl[s:s + CHK], tile_of[s:s + CHK], pos[s:s + CHK]
    x0, y0 = ((tt % gx) * 16).float(), ((tt // gx) * 16).float()
    dx = m2[ids, 0, None, None] - (x0[:, None, None] + px[None, None, :])
    dy = m2[ids, 1, None, None] - (y0[:, None, None] + px[None, :, None])
    q = co[ids]
    power = -0.5 * (q[:, 0, None, None] * dx * dx + q[:, 2, None, None] * dy * dy) - q[:, 1, None, None] * dx * dy
    alive = (power <= 0) & (torch.clamp(q[:, 3, None, None] * torch.exp(power), max=0.99) >= 1.0 / 255.0)
    aq = alive.view(-1, 4, 2, 2, 4, 2, 2).any(dim=6).any(dim=3).permute(0, 1, 3, 2, 4).reshape(-1, 64)
    AQ[s:s + CHK] = aq & (pp[:, None] < quad_max[tt])      # (the forward stops a quadrant's walk there too: all its pixels are done)

if FORWARD:
    back = pos.clone(); inr = pos < tile_qmax[tile_of]      # the quarter / tile stops staging when every pixel is done
else:
    back = tile_qmax[tile_of] - 1 - pos; inr = back >= 0
rnd = torch.where(inr, back // BCH, torch.zeros_like(back))
slot = torch.where(inr, back % BCH, torch.zeros_like(back))
NG = (BCH + 63) // 64
seg = tile_of * 64 + rnd
assert int(rnd.max()) < 64
keep = inr & AQ.any(dim=1)
segk, slotk, AQk = seg[keep], slot[keep], AQ[keep]
useg, seg_id = torch.unique(segk, return_inverse=True)
nseg = useg.numel()
# counts[seg, group of 64 staged slots, quadrant]
cnt = torch.zeros(nseg * NG, 64, dtype=torch.long)
cnt.index_add_(0, seg_id * NG + slotk // 64, AQk.long())
cnt = cnt.view(nseg, NG, 64)
tot = cnt.sum(dim=1)                                         # [seg, 64] whole-round list lengths
seg_tile = useg // 64
print(f"{'forward' if FORWARD else 'backward'}, rounds of {BCH}: {nseg} (tile, round) pairs over {int(torch.unique(seg_tile).numel())} tiles; "
      f"(quadrant, entry) pairs {int(tot.sum())} -> ideal slots {4 * int(tot.sum()) / 1e6:.2f} M")

def c4(x): return (x + 3) // 4

def report(name, wave_passes, n_waves):
    """wave_passes [nseg', n_waves]: passes per wave of a workgroup-round"""
    slots = 64 * int(wave_passes.sum())
    pe = int((wave_passes.amax(dim=1) * n_waves).sum())
    print(f"  {name:58s} slots {slots / 1e6:6.2f} M   pass-equivalents {pe:8d}   balance {float(wave_passes.sum()) / max(pe, 1):.3f}")
    return slots, pe

# ---- today: rows bound to the block's quadrants; the block list is cut into chunks of 64 / 128 BLOCK entries (needs the rank inside the block list) ----
order = torch.argsort(segk * 1024 + slotk, stable=True)
so, ao = seg_id[order], AQk[order].view(-1, 16, 4)
ab = ao.any(dim=2)
cs = torch.cumsum(ab.long(), dim=0)
first = torch.ones(so.numel(), dtype=torch.bool); first[1:] = so[1:] != so[:-1]
start = torch.nonzero(first).squeeze(1)
base = (cs[start] - ab[start].long())[so]
rank = cs - ab.long() - base
res = {}
for ch in (64, 128):
    chunk = rank // ch
    nchunk = int(chunk.max()) + 1
    key = (so[:, None] * 16 + torch.arange(16)[None, :]) * nchunk + chunk
    tt4 = torch.zeros(nseg * 16 * nchunk, 4, dtype=torch.long)
    for qd in range(4):
        m = ao[:, :, qd]
        tt4[:, qd].index_add_(0, key[m], torch.ones(int(m.sum()), dtype=torch.long))
    wp = c4(tt4).amax(dim=1).view(nseg, 16, nchunk).sum(dim=2)
    res[f"today{ch}"] = report(f"today: block-bound rows, {ch}-entry block chunks", wp, 16)

# ---- sorted: quadrants by descending whole-round length, wave j = ranks 4j..4j+3 ----
srt, perm = torch.sort(tot, dim=1, descending=True)
wp = c4(srt.view(nseg, 16, 4)).amax(dim=2)
res["sorted_round"] = report("sorted rows, whole-round lists (50 KB of lists)", wp, 16)
cg = torch.gather(cnt, 2, perm[:, None, :].expand(-1, NG, -1))                     # [seg, group, rank]
wp = c4(cg.view(nseg, NG, 16, 4)).amax(dim=3).sum(dim=1)
res["sorted_grp64"] = report("sorted rows, lists per 64-slot group of the round", wp, 16)
cg2 = cg.view(nseg, NG // 2, 2, 64).sum(dim=2) if NG % 2 == 0 else None
if cg2 is not None:
    wp = c4(cg2.view(nseg, NG // 2, 16, 4)).amax(dim=3).sum(dim=1)
    res["sorted_grp128"] = report("sorted rows, lists per 128-slot group", wp, 16)
# unsorted but free rows make no difference; sorted by length but 8 waves, two lists per row (rank r with rank 63 - r)
pair = srt[:, :32] + srt.flip(dims=(1,))[:, :32]
wp = c4(pair.view(nseg, 8, 4)).amax(dim=2)                                         # (lists walked back to back: ceil of the sum -- optimistic by < 1 pass per row)
res["lpt8"] = report("8 waves per tile, two lists per row (longest + shortest)", wp, 8)
prs, _ = torch.sort(pair, dim=1, descending=True)
wp = c4(prs.view(nseg, 8, 4)).amax(dim=2)
res["lpt8s"] = report("8 waves per tile, pairs sorted again over the waves", wp, 8)
# 4 waves per tile, four lists per row (LPT greedy)
def lpt(lengths, rows):
    """lengths [n, m] sorted descending -> per-row loads [n, rows] by longest-processing-time-first"""
    n, m = lengths.shape
    load = torch.zeros(n, rows, dtype=torch.long)
    for i in range(m):
        j = load.argmin(dim=1)
        load[torch.arange(n), j] += lengths[:, i]
    return load
ld = lpt(srt, 16)
ls, _ = torch.sort(ld, dim=1, descending=True)
res["lpt4"] = report("4 waves per tile, four lists per row (LPT)", c4(ls.view(nseg, 4, 4)).amax(dim=2), 4)

# ---- pool2: two tiles, consecutive in the order the kernels take them (descending list length), share a workgroup; round r of both ----
tiles = torch.unique(seg_tile)
tl = lens[tiles]
ordr = torch.argsort(tl, descending=True, stable=True)
rank_of_tile = torch.full((T,), -1, dtype=torch.long); rank_of_tile[tiles[ordr]] = torch.arange(tiles.numel())
grp = rank_of_tile[seg_tile] // 2
key2 = grp * 64 + (useg % 64)
u2, inv2 = torch.unique(key2, return_inverse=True)
both = torch.zeros(u2.numel(), 2, 64, dtype=torch.long)
both[inv2, rank_of_tile[seg_tile] % 2] = tot
ls2, _ = torch.sort(both.view(-1, 128), dim=1, descending=True)
ld = lpt(ls2, 64)
ls, _ = torch.sort(ld, dim=1, descending=True)
res["pool2"] = report("two tiles per 16-wave workgroup, 128 lists on 64 rows (LPT)", c4(ls.view(-1, 16, 4)).amax(dim=2), 16)

# ---- by list-length class of the tile: where do today's pass-equivalents sit? ----
wp_today = None
print("by tile class (today, 128-entry chunks vs sorted grp64 vs lpt8):")
chunk = rank // 128; nchunk = int(chunk.max()) + 1
key = (so[:, None] * 16 + torch.arange(16)[None, :]) * nchunk + chunk
tt4 = torch.zeros(nseg * 16 * nchunk, 4, dtype=torch.long)
for qd in range(4):
    m = ao[:, :, qd]
    tt4[:, qd].index_add_(0, key[m], torch.ones(int(m.sum()), dtype=torch.long))
wp_today = c4(tt4).amax(dim=1).view(nseg, 16, nchunk).sum(dim=2)
wp_sorted = c4(cg.view(nseg, NG, 16, 4)).amax(dim=3).sum(dim=1)
wp_lpt8 = c4(prs.view(nseg, 8, 4)).amax(dim=2)
L = lens[seg_tile]
for lo, hi in ((1, 128), (128, 384), (384, 1024), (1024, 1 << 30)):
    m = (L >= lo) & (L < hi)
    if not bool(m.any()): continue
    a, b, c = wp_today[m], wp_sorted[m], wp_lpt8[m]
    print(f"  tiles with {lo:5d} <= n < {hi if hi < 1 << 30 else 'inf':>5}: {int(torch.unique(seg_tile[m]).numel()):5d} tiles, {int(m.sum()):5d} rounds | "
          f"today passes {int(a.sum()):7d} PE {int((a.amax(1) * 16).sum()):7d} | sorted passes {int(b.sum()):7d} PE {int((b.amax(1) * 16).sum()):7d} | "
          f"lpt8 passes {int(c.sum()):7d} PE {int((c.amax(1) * 8).sum()):7d}, slowest-wave passes {int(c.amax(1).sum()):6d} vs today {int(a.amax(1).sum()):6d}")
t0 = res["today128"]
print("relative to today (128-entry chunks):")
for k, (s_, p_) in res.items():
    print(f"  {k:14s} slots {s_ / t0[0]:.3f}   pass-equivalents {p_ / t0[1]:.3f}")

# ---- static block pairings: NW waves per tile, wave w walks blocks w, w + NW, ... one after the other (today's block-bound rows and chunks) ----
print("static block-serial variants (today's lists; wave w walks blocks w, w + NW, ...):")
bp = c4(tt4).amax(dim=1).view(nseg, 16, nchunk).sum(dim=2)        # passes per block and (tile, round), 128-entry chunks
for nw in (16, 8, 4):
    wpn = bp.view(nseg, 16 // nw, nw).sum(dim=1)
    s_, p_ = report(f"{nw} waves per tile, blocks w + {nw} k", wpn, nw)
    print(f"      slowest wave's passes summed over (tile, round): {int(wpn.amax(1).sum())}   relative PE {p_ / t0[1]:.3f}")
# 8 waves, block b paired with 15 - b (corner with corner of the other side, centre with centre)
idx = torch.arange(8)
wpn = bp[:, idx] + bp[:, 15 - idx]
s_, p_ = report("8 waves per tile, block b with 15 - b", wpn, 8)
print(f"      slowest wave's passes: {int(wpn.amax(1).sum())}   relative PE {p_ / t0[1]:.3f}")
# 8 waves, blocks paired per round by their list length (longest with shortest)
bl = ab.long()
blen = torch.zeros(nseg, 16, dtype=torch.long); blen.index_add_(0, so, bl)
o = torch.argsort(blen, dim=1, descending=True)
bps = torch.gather(bp, 1, o)
wpn = bps[:, :8] + bps.flip(dims=(1,))[:, :8]
s_, p_ = report("8 waves per tile, blocks paired by list length per round", wpn, 8)
print(f"      slowest wave's passes: {int(wpn.amax(1).sum())}   relative PE {p_ / t0[1]:.3f}")
ld = lpt(bps, 4)
s_, p_ = report("4 waves per tile, blocks by LPT on list length", ld, 4)
print(f"      slowest wave's passes: {int(ld.amax(1).sum())}   relative PE {p_ / t0[1]:.3f}")

# ---- forward only: how many of a tile's staged entries reach each QUARTER (k_render_fwd stages the whole list in every quarter's workgroup) ----
if FORWARD:
    AB16 = AQ.view(R, 16, 4).any(dim=2)                       # [R, block]
    blk_of_quarter = torch.tensor([[0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]])
    AQr = torch.stack([AB16[:, blk_of_quarter[q]].any(dim=1) for q in range(4)], dim=1) & inr[:, None]   # [R, quarter]: entry reaches the quarter (and the quarter's walk has not ended)
    heavy = lens[tile_of] >= 128
    staged = int((inr & heavy).sum()) * 4
    reach = int(AQr[heavy].sum())
    # rounds per quarter today (all entries up to the tile's deepest position, 256 per round) against compacted staging (256 REACHING entries per round)
    tq_heavy = torch.unique(tile_of[heavy])
    r_today = int(((tile_qmax[tq_heavy] + BCH - 1) // BCH).sum()) * 4
    per_q = torch.zeros(T, 4, dtype=torch.long); per_q.index_add_(0, tile_of[heavy], AQr[heavy].long())
    r_comp = int(((per_q[tq_heavy] + BCH - 1) // BCH).clamp(min=1).sum())
    print(f"forward, tiles with >= 128 entries ({tq_heavy.numel()}): entries staged by the four quarters {staged}, of which reach the quarter {reach} ({reach / max(staged, 1):.3f}); "
          f"quarter-rounds today {r_today}, with staging compacted per quarter {r_comp}")
