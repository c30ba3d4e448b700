"""Pass time of the radix sort against the number of tiles (start-up of the look-back chain vs steady state):
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/sort_sweep.py run
    python tools/sort_sweep.py parse OUT
Keys: 16 live key bits (two 8-bit passes), uniformly random."""
import csv, glob, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SIZES = [1 << 20, 2 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 13762560, 16 << 20, 24 << 20, 32 << 20]
REPS = 4

if sys.argv[1] == "run":
    import numpy as np
    import forma_amd
    c = forma_amd.Context(0)
    rng = np.random.default_rng(1)
    for n in SIZES:
        v = (rng.integers(0, 1 << 16, n, dtype=np.uint64) << np.uint64(20)) | rng.integers(0, 1 << 20, n, dtype=np.uint64)
        for _ in range(REPS):
            c.sort_array(v)
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_onesweep" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    per = 2 * REPS
    for i, n in enumerate(SIZES):
        x = sorted(d[i * per + 2:(i + 1) * per])            # skip the first repetition
        tiles = (n + 16383) // 16384
        print("N %9d tiles %5d rounds %5.2f  pass us min %6.1f med %6.1f  -> %5.2f TB/s" % (n, tiles, tiles / 256, x[0], x[len(x) // 2], 16 * n / x[len(x) // 2] / 1e6))
