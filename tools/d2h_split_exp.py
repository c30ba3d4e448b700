#!/usr/bin/env python3
"""frames/s INTO (registered) CALLER MEMORY of synchronous calls against the painter's band split (api.cpp split_plan): one launch,
the policy's two bands with the first at P percent of the rows, N equal bands.   python tools/d2h_split_exp.py [workload] split_first=17 split_first=25 paint_split=4 ..."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENE = "/tmp/ab_fast_scene_%s.npz"
def child(wl):
    import forma_amd
    from forma_amd import scenes
    t = np.load(SCENE % wl)
    _, W, H = scenes.WORKLOADS[wl]
    c = forma_amd.Context(0)
    c.set_geometry(t["x"], t["y"], t["line_slot"]); c.set_geoms(t["geoms"])
    c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
    img = np.zeros((H, W * 4), np.uint8)
    c.register_buffer(img)
    def loop(n):
        for _ in range(5): c.render(W, H, clear=(1, 1, 1, 1), dst=img)
        c.sync(); t0 = time.perf_counter()
        for _ in range(n): c.render(W, H, clear=(1, 1, 1, 1), dst=img)
        c.sync(); return n / (time.perf_counter() - t0)
    r = [loop(100) for _ in range(3)]
    print(json.dumps({"fps": [round(x, 1) for x in r], "crc": int(np.bitwise_xor.reduce(img.view(np.uint32).reshape(-1)))}))
    c.unregister_buffer(img); c.close()
if "--child" in sys.argv:
    child(sys.argv[1])
else:
    wl = sys.argv[1]
    if not os.path.exists(SCENE % wl):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_fast.py"), "--workload", wl, "--rounds", "0"])
    for rd in range(2):
        for name, env in [("one launch", "paint_split=0")] + [(v, v) for v in (sys.argv[2:] or ["split_first=25"])]:
            env = {"FORMA_HIP_DEBUG": env}
            p = subprocess.run([sys.executable, os.path.abspath(__file__), wl, "--child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            print("%-12s" % name, [l for l in p.stdout.splitlines() if l.startswith("{")] or p.stderr[-400:], flush=True)
