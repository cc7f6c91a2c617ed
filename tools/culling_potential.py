"""How much of the render kernels' lane work is useful, by culling granularity (config 3, view 0): for every tile instance
count the pixels with alpha >= 1/255, and the 4x4 blocks / 2x2 quadrants that contain at least one such pixel."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from youreditableavatar_amd import scenes
from diff_gaussian_rasterization import _C
dev = torch.device("cuda", 0)
cfg = scenes.CONFIGS[3]; P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
cloud = scenes.make_cloud(P, D, cfg["seed"])
g = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
e = torch.Tensor([])
c = scenes.orbit_camera(W, H, azimuth_deg=0.0)
R, color, radii, geom, binning, img = _C.rasterize_gaussians(g(c.bg), g(cloud["means3D"]), e, g(cloud["opacities"]), g(cloud["scales"]), g(cloud["rotations"]), 1.0, e,
    g(c.viewmatrix), g(c.projmatrix), c.tanfovx, c.tanfovy, H, W, g(cloud["shs"]), D, g(c.campos), False, False)
f = lambda n: _C.state_field(n, P, W, H, R, True, True, geom, binning, img)
pl = f("point_list").long(); rg = f("ranges").view(-1, 2).long(); m2 = f("means2D").view(-1, 2); co = f("conic_opacity").view(-1, 4)
gx = (W + 15) // 16
lens = rg[:, 1] - rg[:, 0]
tile_of = torch.repeat_interleave(torch.arange(rg.shape[0], device=dev), lens)          # instances are stored tile-major
px = torch.arange(16, device=dev, dtype=torch.float32)
tot_px = tot_blk = tot_quad = 0
T = rg.shape[0]
nb = torch.zeros(T, 16, device=dev, dtype=torch.long); nq = torch.zeros(T, 64, device=dev, dtype=torch.long)
for s in range(0, R, 1 << 18):
    ids, t = pl[s:s + (1 << 18)], tile_of[s:s + (1 << 18)]
    x0, y0 = ((t % gx) * 16).float(), ((t // gx) * 16).float()
    dx = m2[ids, 0, None, None] - (x0[:, None, None] + px[None, None, :])
    dy = m2[ids, 1, None, None] - (y0[:, None, None] + px[None, :, None])
    q = co[ids]
    power = -0.5 * (q[:, 0, None, None] * dx * dx + q[:, 2, None, None] * dy * dy) - q[:, 1, None, None] * dx * dy
    alive = (power <= 0) & (torch.minimum(torch.tensor(0.99, device=dev), q[:, 3, None, None] * torch.exp(power)) >= 1.0 / 255.0)   # [n,16(y),16(x)]
    tot_px += int(alive.sum())
    tot_blk += int(alive.view(-1, 4, 4, 4, 4).any(dim=4).any(dim=2).sum())
    tot_quad += int(alive.view(-1, 8, 2, 8, 2).any(dim=4).any(dim=2).sum())
    nb.index_add_(0, t, alive.view(-1, 4, 4, 4, 4).any(dim=4).any(dim=2).view(-1, 16).long())
    # quadrant index = block * 4 + quadrant-in-block
    qa = alive.view(-1, 4, 2, 2, 4, 2, 2).any(dim=6).any(dim=3)             # [n, by, qy, bx, qx]
    nq.index_add_(0, t, qa.permute(0, 1, 3, 2, 4).reshape(-1, 64).long())
print(f"instances {R}; alive (pixel, instance) pairs {tot_px}")
print(f"4x4 blocks touched {tot_blk} -> {16 * tot_blk} lane-pairs processed, {tot_px / (16 * tot_blk) * 100:.1f} % useful")
print(f"2x2 quadrants touched {tot_quad} -> {4 * tot_quad} lane-pairs, {tot_px / (4 * tot_quad) * 100:.1f} % useful; ratio 4x4 / 2x2 = {16 * tot_blk / (4 * tot_quad):.2f}")

# wave passes of the render kernels (4 entries per pass; single staging round assumed, early termination ignored)
c4 = lambda n: (n + 3) // 4
cur = c4(nb)                                              # [T,16] passes of each wave now
new = c4(nq).view(T, 16, 4).max(dim=2).values            # each 16-lane row walks its own quadrant's list
print(f"wave passes: block lists {int(cur.sum())}, quadrant lists {int(new.sum())}  ({cur.sum() / new.sum():.2f}x)")
crit_cur, crit_new = cur.max(dim=1).values, new.max(dim=1).values
print(f"critical path per tile (slowest wave), summed: {int(crit_cur.sum())} -> {int(crit_new.sum())}  ({crit_cur.sum() / crit_new.sum():.2f}x)")
top = torch.argsort(lens, descending=True)[:256]
print(f"256 longest tiles: passes {int(cur[top].sum())} -> {int(new[top].sum())} ({cur[top].sum() / new[top].sum():.2f}x); slowest wave {int(crit_cur[top].sum())} -> {int(crit_new[top].sum())} ({crit_cur[top].sum() / crit_new[top].sum():.2f}x)")
print(f"sum_q n_q / n_b over blocks: {nq.sum() / nb.sum():.2f}; blocks whose 4 quadrant lists exceed 2x512 entries: {int((nq.view(T,16,4).sum(2) > 1000).sum())}")

# ---- backward: how many list entries lie behind the last contributor of a block / quadrant (they are walked up to the tile's qmax today)
nc = f("n_contrib").view(H, W).long()
Hp, Wp = (H + 15) // 16 * 16, gx * 16
ncp = torch.zeros(Hp, Wp, device=dev, dtype=torch.long); ncp[:H, :W] = nc
tq = ncp.view(Hp // 16, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 16, 16)               # [T, y, x]
tile_qmax = tq.reshape(T, -1).max(dim=1).values
blk_max = tq.view(T, 4, 4, 4, 4).permute(0, 1, 3, 2, 4).reshape(T, 16, 16).max(dim=2).values           # [T, block]
quad_max = tq.view(T, 8, 2, 8, 2).permute(0, 1, 3, 2, 4).reshape(T, 64, 4).max(dim=2).values           # [T, 8*row + col]
pos_in_tile = torch.arange(R, device=dev) - rg[tile_of, 0]                                             # list position of every instance
cnt_tile = cnt_blk = cnt_quad_tile = cnt_quad = 0
for s in range(0, R, 1 << 18):
    ids, t, pos = pl[s:s + (1 << 18)], tile_of[s:s + (1 << 18)], pos_in_tile[s:s + (1 << 18)]
    x0, y0 = ((t % gx) * 16).float(), ((t // gx) * 16).float()
    dx = m2[ids, 0, None, None] - (x0[:, None, None] + px[None, None, :])
    dy = m2[ids, 1, None, None] - (y0[:, None, None] + px[None, :, None])
    q = co[ids]
    power = -0.5 * (q[:, 0, None, None] * dx * dx + q[:, 2, None, None] * dy * dy) - q[:, 1, None, None] * dx * dy
    alive = (power <= 0) & (torch.minimum(torch.tensor(0.99, device=dev), q[:, 3, None, None] * torch.exp(power)) >= 1.0 / 255.0)
    ab = alive.view(-1, 4, 4, 4, 4).any(dim=4).any(dim=2).view(-1, 16)                      # [n, block]
    aq = alive.view(-1, 8, 2, 8, 2).any(dim=4).any(dim=2).view(-1, 64)                      # [n, quadrant (raster)]
    in_tile = (pos < tile_qmax[t])[:, None]
    cnt_tile += int((ab & in_tile).sum()); cnt_blk += int((ab & (pos[:, None] < blk_max[t])).sum())
    cnt_quad_tile += int((aq & in_tile).sum()); cnt_quad += int((aq & (pos[:, None] < quad_max[t])).sum())
print(f"backward, (block, entry) pairs in front of the tile's qmax {cnt_tile}, in front of the block's own last contributor {cnt_blk} ({cnt_blk / cnt_tile:.3f})")
print(f"backward, (quadrant, entry) pairs in front of the tile's qmax {cnt_quad_tile}, in front of the quadrant's own last contributor {cnt_quad} ({cnt_quad / cnt_quad_tile:.3f})")

# ---- passes of the forward kernel as built (512-entry rounds, 64-entry chunks of the block list) against other chunk sizes
AB = torch.zeros(R, 16, dtype=torch.bool, device=dev); AQ = torch.zeros(R, 64, dtype=torch.bool, device=dev)
for s in range(0, R, 1 << 18):
    ids, t = pl[s:s + (1 << 18)], tile_of[s:s + (1 << 18)]
    x0, y0 = ((t % gx) * 16).float(), ((t // gx) * 16).float()
    dx = m2[ids, 0, None, None] - (x0[:, None, None] + px[None, None, :])
    dy = m2[ids, 1, None, None] - (y0[:, None, None] + px[None, :, None])
    q = co[ids]
    power = -0.5 * (q[:, 0, None, None] * dx * dx + q[:, 2, None, None] * dy * dy) - q[:, 1, None, None] * dx * dy
    alive = (power <= 0) & (torch.minimum(torch.tensor(0.99, device=dev), q[:, 3, None, None] * torch.exp(power)) >= 1.0 / 255.0)
    AB[s:s + (1 << 18)] = alive.view(-1, 4, 4, 4, 4).any(dim=4).any(dim=2).view(-1, 16)
    AQ[s:s + (1 << 18)] = alive.view(-1, 4, 2, 2, 4, 2, 2).any(dim=6).any(dim=3).permute(0, 1, 3, 2, 4).reshape(-1, 64)    # block*4 + quadrant
rnd = pos_in_tile // 512
seg = tile_of * 8 + rnd                                   # (tile, round) segment of every instance (rounds < 8 here)
assert int(rnd.max()) < 8
cs = torch.cumsum(AB.long(), dim=0)                       # running count per block over all instances
first = torch.zeros(int(seg.max()) + 2, dtype=torch.long, device=dev)
segstart = torch.ones(R, dtype=torch.bool, device=dev); segstart[1:] = seg[1:] != seg[:-1]
start_idx = torch.nonzero(segstart).squeeze(1)
base = torch.zeros(R, 16, dtype=torch.long, device=dev)
base_vals = cs[start_idx] - AB[start_idx].long()          # count before the segment's first instance
seg_id = torch.cumsum(segstart.long(), 0) - 1
rank = cs - AB.long() - base_vals[seg_id]                 # rank of the instance in its block's list of this round
for ch in (64, 128, 4096):
    chunk = rank // ch                                    # [R,16]
    nchunk = int(chunk.max()) + 1
    key = (seg_id[:, None] * 16 + torch.arange(16, device=dev)[None, :]) * nchunk + chunk          # [R,16]
    tot = torch.zeros((int(seg_id.max()) + 1) * 16 * nchunk, 4, dtype=torch.long, device=dev)
    for qd in range(4):
        m = AQ.view(R, 16, 4)[:, :, qd] & AB
        tot[:, qd].index_add_(0, key[m], torch.ones(int(m.sum()), dtype=torch.long, device=dev))
    passes = ((tot + 3) // 4).max(dim=1).values.sum()
    print(f"forward passes with {ch}-entry chunks of the block list: {int(passes)}")
print(f"ideal (every row always busy): {int((AQ.view(R,16,4) & AB[:, :, None]).sum()) // 16}")
