"""Phase breakdown of k_onesweep on a REAL frame's keys (tools/sort_prof.py sorts random digits; needs a -DSORT_PROF build in
FORMA_HIP_LIB and the scene parked by tools/ab_fast.py --rounds 0).  python tools/sort_prof_frame.py [workload]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import forma_amd
from forma_amd import scenes, _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "paris-like-30k-4k"
t = np.load("/tmp/ab_fast_scene_%s.npz" % wl)
_, W, H = scenes.WORKLOADS[wl]
c = forma_amd.Context(0)
c.set_geometry(t["x"], t["y"], t["line_slot"]); c.set_geoms(t["geoms"])
c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
for _ in range(4):
    c.render(W, H, clear=(1, 1, 1, 1), device_only=True)
L = _lib.lib()
buf = (C.c_ulonglong * 16)()
L.forma_hip_debug_sort_prof(buf, 1)
N = 10
for _ in range(N):
    c.render(W, H, clear=(1, 1, 1, 1), device_only=True)
c.sync()
L.forma_hip_debug_sort_prof(buf, 0)
tiles = buf[15]
names = ["0 ticket+clear+barrier", "1 key loads", "2 rank+barrier", "3 totals/scan/bases", "4 staging issue", "5 look-back+barrier", "", "7 scatter+barrier"]
tot = sum(buf[i] for i in range(8))
print(f"{wl}: {tiles} tiles over {N} frames")
for i, nm in enumerate(names):
    if nm:
        print(f"  {nm:24s} {buf[i] / tiles:8.0f} clocks/tile  {100 * buf[i] / tot:5.1f}%")
print(f"  total {tot / tiles:.0f} clocks per tile")
