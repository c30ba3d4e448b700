#!/usr/bin/env python3
"""Thread sweep of the CPU oracle (bench.py's `cpu_baseline` leg alone): per-stage milliseconds of one frame of a workload for
each thread count.  The scene tables come from /tmp/ab_fast_scene_<workload>.npz (tools/ab_fast.py parks them).

    python tools/cpu_sweep.py [workload] [threads,threads,...]"""
import os
import sys

os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from forma_amd import scenes                               # noqa: E402
from oracle import oracle as orc                           # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "paris-like-30k-4k"
t = np.load("/tmp/ab_fast_scene_%s.npz" % w)
_, W, H = scenes.WORKLOADS[w]
hw = orc.lib().oracle_max_threads()
cands = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [c for c in (1, 8, 16, 32, 48, 64, 96, 128, 192, 256) if c <= hw]
o = orc.Oracle(threads=1)
o.set_geometry(t["x"], t["y"], t["line_slot"]); o.set_geoms(t["geoms"])
o.set_styles(t["style_offsets"], t["style_words"], None); o.set_images(t["images"], t["texels"])
print("hardware threads", hw)
for c in cands:
    o.set_threads(c)
    o.time_frame(W, H, 1)
    tm = o.time_frame(W, H, 3)
    print("%4d threads: %7.1f ms  " % (c, sum(tm.values()) * 1e3) + "  ".join("%s %.1f" % (k, v * 1e3) for k, v in tm.items()), flush=True)
