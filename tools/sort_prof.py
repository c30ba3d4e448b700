"""Phase breakdown of k_onesweep (needs a -DSORT_PROF build: tools/build_variants.sh sprof:"-DSORT_PROF", copy it over libforma_hip.so)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import forma_amd
from forma_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13762560
c = forma_amd.Context(0)
rng = np.random.default_rng(1)
v = (rng.integers(0, 1 << 16, n, dtype=np.uint64) << np.uint64(20)) | rng.integers(0, 1 << 20, n, dtype=np.uint64)
L = _lib.lib()
buf = (C.c_ulonglong * 16)()
c.sort_array(v)
L.forma_hip_debug_sort_prof(buf, 1)
R = 4
for _ in range(R):
    c.sort_array(v)
L.forma_hip_debug_sort_prof(buf, 0)
tiles = buf[15]
names = ["0 ticket+clear+barrier", "1 key loads", "2 rank+barrier", "3 totals/scan/bases", "4 staging issue", "5 look-back+barrier", "", "7 scatter+barrier"]
tot = sum(buf[i] for i in range(8))
print(f"N {n}: {tiles} tiles over {2 * R} passes")
for i, nm in enumerate(names):
    if nm:
        print(f"  {nm:24s} {buf[i] / tiles:8.0f} clocks/tile  {100 * buf[i] / tot:5.1f}%")
print(f"  total {tot / tiles:.0f} clocks per tile")
