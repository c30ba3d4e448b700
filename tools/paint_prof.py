"""Phase breakdown of k_paint_wave (needs a -DPAINT_PROF build of libforma_hip.so: tools/build_variants.sh prof:"-DPAINT_PROF")."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from forma_amd import api, scenes, _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "paris-like-30k-4k"
build, W, H = scenes.WORKLOADS[wl]
comp = build()
r = api.Renderer(0)
img = np.zeros(W * H * 4, np.uint8)
r.render(comp, api.BufferBuilder(img, api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
L = _lib.lib()
buf = (C.c_ulonglong * 24)()
for _ in range(3):
    r._ctx.render(W, H, clear=(1, 1, 1, 1), device_only=True)
L.forma_hip_debug_paint_prof(buf, 1)
N = 10
for _ in range(N):
    r._ctx.render(W, H, clear=(1, 1, 1, 1), device_only=True)
L.forma_hip_debug_paint_prof(buf, 0)
v = [x / N for x in buf]
names = ["0 find runs+spans", "1 merge+flags", "2 passes", "3 solid fold+store", "4 fold failed", "5 painted list", "6 batch staging",
         "7 seg accumulate", "8 cover/blend solid+Over", "9 srgb+store", "10 cover/fill/blend other"]
tiles = v[16]
tot = sum(v[:11])
print(f"{wl}: tiles {tiles:.0f}, entries/tile {v[17]/tiles:.1f}, row spans/tile {v[18]/tiles:.1f}, solid tiles {v[19]:.0f}, "
      f"painted entries/tile {v[20]/max(tiles - v[19], 1):.1f}, segments accumulated/tile {v[21]/max(tiles - v[19], 1):.1f}")
for i, n in enumerate(names):
    print(f"  {n:22s} {v[i]/tiles:9.0f} cycles/tile  {100*v[i]/tot:5.1f}%")
print(f"  total {tot/tiles:.0f} cycles per tile (wave-serial)")
print(f"  solid/Over layers painted {v[23]:.0f}: {v[8]/max(v[23],1):.0f} cycles each;  other layers {v[22]:.0f}: {v[10]/max(v[22],1):.0f} cycles each")
