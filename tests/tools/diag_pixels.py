#!/usr/bin/env python3
"""Which pixels carry the difference between the default and the fixed-order per-pixel backward of one fuzz scene?  The backward is linear
in dL/d image, so the difference is bisected over image rectangles.   python tests/tools/diag_pixels.py <seed> <scene>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import util, fuzz

seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(want + 1):
    desc, inp, dL = fuzz.random_scene(rng, it)
H, W = int(inp["image_height"]), int(inp["image_width"])
key = "dL_dconic"

def diff(mask):
    d = dL * mask[None]
    a = util.hip_run(inp, d, introspect=False)[key]; b = util.hip_run(inp, d, introspect=False, deterministic=True)[key]
    return float(np.abs(np.asarray(a) - np.asarray(b)).sum()), float(np.abs(np.asarray(b)).sum())

full = np.ones((H, W), np.float32)
e, tot = diff(full)
print(f"scene {desc}: sum|default - fixed order| of {key} = {e:.4e} (sum|.| {tot:.4e})")
x0, x1, y0, y1 = 0, W, 0, H
while (x1 - x0) * (y1 - y0) > 1:
    if x1 - x0 >= y1 - y0:
        xm = (x0 + x1) // 2; boxes = [(x0, xm, y0, y1), (xm, x1, y0, y1)]
    else:
        ym = (y0 + y1) // 2; boxes = [(x0, x1, y0, ym), (x0, x1, ym, y1)]
    es = []
    for (a, b, c, d) in boxes:
        m = np.zeros((H, W), np.float32); m[c:d, a:b] = 1
        es.append(diff(m)[0])
    k = int(np.argmax(es))
    print(f"  box x [{boxes[k][0]},{boxes[k][1]}) y [{boxes[k][2]},{boxes[k][3]}): {es[k]:.4e}   (other half {es[1 - k]:.4e})")
    x0, x1, y0, y1 = boxes[k]
print(f"pixel ({x0}, {y0})  tile ({x0 // 16}, {y0 // 16})")
mine = util.hip_run(inp, dL)
ref = util.oracle_run(inp, dL)
print("n_contrib hip / oracle:", int(mine["n_contrib"][y0, x0]), int(ref["n_contrib"][y0, x0]), " final_T hip / oracle:", float(mine["final_T"][y0, x0]), float(ref["final_T"][y0, x0]))
t = (y0 // 16) * ((W + 15) // 16) + x0 // 16
r = mine["ranges"][t]
print("tile list length", int(r[1] - r[0]))
