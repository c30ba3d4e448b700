"""Phase breakdown of k_rasterize (needs a -DRAS_PROF build: tools/build_variants.sh rprof:"-DRAS_PROF", copied over libforma_hip.so)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from forma_amd import api, scenes, _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "paris-like-30k-4k"
build, W, H = scenes.WORKLOADS[wl]
r = api.Renderer(0)
img = np.zeros(W * H * 4, np.uint8)
r.render(build(), api.BufferBuilder(img, api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
L = _lib.lib()
buf = (C.c_ulonglong * 8)()
for _ in range(3):
    r._ctx.render(W, H, clear=(1, 1, 1, 1), device_only=True)
L.forma_hip_debug_ras_prof(buf, 1)
N = 10
for _ in range(N):
    r._ctx.render(W, H, clear=(1, 1, 1, 1), device_only=True)
L.forma_hip_debug_ras_prof(buf, 0)
wg = buf[7]
names = ["0 lines staged (loads, f64 constants, barrier)", "1 binary search", "2 eight segments per lane", "3 stores + mask reduction"]
tot = sum(buf[i] for i in range(4))
print(f"{wl}: {wg / N:.0f} workgroups per frame")
for i, n in enumerate(names):
    print(f"  {n:48s} {buf[i] / wg:8.0f} clocks/workgroup  {100 * buf[i] / tot:5.1f}%")
print(f"  total {tot / wg:.0f} clocks per workgroup (thread 0)")
