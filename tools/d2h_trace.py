#!/usr/bin/env python3
"""a few frames into a registered caller buffer, for `rocprofv3 --kernel-trace --memory-copy-trace`: do the band copies overlap the painters?"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import forma_amd
from forma_amd import scenes
wl = "paris-like-30k-4k"
t = np.load("/tmp/ab_fast_scene_%s.npz" % wl)
_, W, H = scenes.WORKLOADS[wl]
c = forma_amd.Context(0)
c.set_geometry(t["x"], t["y"], t["line_slot"]); c.set_geoms(t["geoms"])
c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
img = np.zeros((H, W * 4), np.uint8)
c.register_buffer(img)
for _ in range(12):
    c.render(W, H, clear=(1, 1, 1, 1), dst=img)
c.close()
