"""How many distinct radix digits does a 64-key row of the sort's input hold?  (decides between the ballot match-any over
all 8 digit bits and a leader loop over the distinct digits of a row)   python tools/digit_rows.py [workload]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from forma_amd import api, scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "paris-like-30k-4k"
build, W, H = scenes.WORKLOADS[wl]
r = api.Renderer(0)
img = np.zeros(W * H * 4, np.uint8)
r.render(build(), api.BufferBuilder(img, api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
v = r._ctx.segments(0)
n = len(v) // 64 * 64
tx = ((v >> np.uint64(41)) & np.uint64(0xFF)).astype(np.int32)
ty = ((v >> np.uint64(53)) & np.uint64(0xFF)).astype(np.int32)
def stats(name, d):
    rows = np.sort(d[:n].reshape(-1, 64), axis=1)
    distinct = 1 + (np.diff(rows, axis=1) != 0).sum(axis=1)
    h = np.bincount(distinct, minlength=65)
    cum = np.cumsum(h) / h.sum()
    print(f"{name}: mean distinct {distinct.mean():.2f}; rows with <=1: {cum[1]:.3f} <=2: {cum[2]:.3f} <=3: {cum[3]:.3f} <=4: {cum[4]:.3f} <=6: {cum[6]:.3f} <=8: {cum[8]:.3f} <=16: {cum[16]:.3f}")
stats("pass 1 (tile_x digit, rasterizer order)", tx)
order = np.argsort(tx, kind="stable")
stats("pass 2 (tile_y digit, after pass 1)", ty[order])
