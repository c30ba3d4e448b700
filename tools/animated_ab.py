#!/usr/bin/env python3
"""bench.py's animated leg (BASELINE config 5: the deterministic spaceship at 4K, with and without a buffer-layer cache) alone, under
the FORMA_HIP_DEBUG strings given (same box):   python tools/animated_ab.py "" paint_split=0 paint_split=2"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT)
    import bench
    print(json.dumps(bench.animated_leg(0, 300)))
else:
    for rd in range(2):
        for sw in (sys.argv[1:] or [""]):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, FORMA_HIP_DEBUG=sw), capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1]) if line else {"err": p.stderr[-300:]}
            print("%-16s no_cache %s with_cache %s" % (sw or "(default)", d.get("fps_no_cache"), d.get("fps_with_cache") or d), flush=True)
