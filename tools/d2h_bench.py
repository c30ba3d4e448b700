#!/usr/bin/env python3
"""frames/s of Renderer::render INTO CALLER MEMORY from one context and one host thread (the PCIe-inclusive rate):
pageable and registered destinations, and forma_hip_render_enqueue over three registered buffers (two runs: PCIe rates wobble).   python tools/d2h_bench.py [workload]"""
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
    out = {}
    img = np.zeros((H, W * 4), np.uint8)

    def loop(n, fn):
        for _ in range(5):
            fn()
        c.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        c.sync()
        return round(n / (time.perf_counter() - t0), 1)
    out["device_only_fps"] = loop(100, lambda: c.render(W, H, clear=(1, 1, 1, 1), device_only=True))
    out["pageable_fps"] = loop(100, lambda: c.render(W, H, clear=(1, 1, 1, 1), dst=img))
    c.register_buffer(img)
    out["registered_fps"] = loop(100, lambda: c.render(W, H, clear=(1, 1, 1, 1), dst=img))
    c.unregister_buffer(img)
    c.set_frames_in_flight(2)
    bufs = [np.zeros((H, W * 4), np.uint8) for _ in range(3)]
    for b in bufs:
        c.register_buffer(b)
    k = [0]

    def enq():
        c.render_enqueue(W, H, bufs[k[0] % 3], clear=(1, 1, 1, 1)); k[0] += 1
    out["enqueue_3_registered_buffers_fps"] = loop(150, enq)
    for b in bufs:
        c.unregister_buffer(b)
    c.close()
    print(json.dumps(out))


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "paris-like-30k-4k"
    if "--child" in sys.argv:
        child(wl)
    else:
        if not os.path.exists(SCENE % wl):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_fast.py"), "--workload", wl, "--rounds", "0"])
        for name, env in (("run 1", {}), ("run 2", {})):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), wl, "--child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            print(name, [l for l in p.stdout.splitlines() if l.startswith("{")] or p.stderr[-600:])
