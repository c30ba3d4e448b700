#!/usr/bin/env python3
"""Band frames only, for `rocprofv3 --kernel-trace --stats`: what one device of an 8-GPU context runs per frame.

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python tools/band_trace.py [--workload W] [--band r0,r1] [--frames K] [--debug SWITCHES]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="paris-like-30k-4k")
ap.add_argument("--band", default=None)
ap.add_argument("--frames", type=int, default=200)
a = ap.parse_args()
import forma_amd                                       # noqa: E402
from forma_amd import scenes                            # noqa: E402
t = np.load("/tmp/ab_fast_scene_%s.npz" % a.workload)
_, W, H = scenes.WORKLOADS[a.workload]
th = (H + 15) // 16
r0, r1 = (int(v) for v in a.band.split(",")) if a.band else (th * 7 // 16, th * 7 // 16 + (th + 7) // 8)
c = forma_amd.Context(0)
c.set_geometry(t["x"], t["y"], t["line_slot"]); c.set_geoms(t["geoms"])
c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
c.set_band(r0, r1)
crop = (0, W, r0 * 16, min(r1 * 16, H))
for _ in range(a.frames):
    c.render(W, H, clear=(1, 1, 1, 1), crop=crop, device_only=True)
c.sync()
c.close()
