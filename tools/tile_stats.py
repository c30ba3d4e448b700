#!/usr/bin/env python3
"""How deep are the tiles of a workload?  Renders one frame, reads the sorted stream back and prints the distribution of
(tile, layer) runs and pixel segments per 16x16 tile — what decides how long the painter's slowest wavefront lives.

    python tools/tile_stats.py [workload]        (after tools/ab_fast.py parked the scene in /tmp)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import forma_amd                                       # noqa: E402
from forma_amd import scenes                            # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "paris-like-30k-4k"
t = np.load("/tmp/ab_fast_scene_%s.npz" % w)
_, W, H = scenes.WORKLOADS[w]
c = forma_amd.Context(0)
c.set_geometry(t["x"], t["y"], t["line_slot"]); c.set_geoms(t["geoms"])
c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
c.render(W, H, clear=(1, 1, 1, 1), device_only=True)
s = c.segments(1)
tile = (s >> np.uint64(41)).astype(np.int64)            # tile_y + 1 (11) | tile_x + 1 (12)
tl = (s >> np.uint64(20)).astype(np.int64)              # ... | layer (21)
ty, tx = (tile >> 12) - 1, (tile & 4095) - 1
ok = (ty >= 0) & (ty < (H + 15) // 16) & (tx >= 0) & (tx < (W + 15) // 16)
tile, tl = tile[ok], tl[ok]
heads = np.concatenate([[True], tl[1:] != tl[:-1]])
ut, segs = np.unique(tile, return_counts=True)
_, runs = np.unique(tile[heads], return_counts=True)
q = [50, 90, 99, 99.9, 100]
print(w, "tiles with segments", len(ut), "of", ((H + 15) // 16) * ((W + 15) // 16))
print("runs per tile     mean %.1f  " % runs.mean() + " ".join("p%g %d" % (p, np.percentile(runs, p)) for p in q))
print("segments per tile mean %.1f  " % segs.mean() + " ".join("p%g %d" % (p, np.percentile(segs, p)) for p in q))
c.close()
