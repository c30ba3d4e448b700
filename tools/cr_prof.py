"""Phase breakdown of k_carry_rows (needs a -DCR_PROF build: tools/build_variants.sh crprof:"-DCR_PROF", copied over libforma_hip.so)."""
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
L.forma_hip_debug_cr_prof(buf, 1)
N = 10
for _ in range(N):
    r._ctx.render(W, H, clear=(1, 1, 1, 1), device_only=True)
L.forma_hip_debug_cr_prof(buf, 0)
rows = buf[7]
names = ["0 prologue (counts, row prefix)", "1 run keys -> LDS", "2 in-LDS sort: digit totals, bases, scatter", "3 pieces: waiting for the gathers", "4 pieces: scan, carry, spans, stores",
         "5 in-LDS sort: clearing the counters", "6 in-LDS sort: ranking"]
tot = sum(buf[i] for i in range(7))
print(f"{wl}: {rows / N:.0f} rows per frame")
for i, n in enumerate(names):
    print(f"  {n:44s} {buf[i] / rows:9.0f} clocks/row  {100 * buf[i] / tot:5.1f}%")
print(f"  total {tot / rows:.0f} clocks per row (thread 0; the last piece's tail is not stamped)")
