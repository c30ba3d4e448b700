#!/usr/bin/env python3
"""Feasibility probe: a frame into caller memory as K BAND frames (K contexts on one GPU, each culled to a band of tile rows with
forma_hip_set_band, each writing its rows of the SAME registered destination from its own host thread), against the one-context
frame: does a first band's copy that starts long before the whole frame is painted pay for the repeated line work?
   python tools/d2h_ctx_bands_exp.py [workload] [cuts ...]       a cut list = percentages of the tile rows, e.g. 30 or 20,50"""
import json, os, subprocess, sys, threading, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENE = "/tmp/ab_fast_scene_%s.npz"


def child(wl, cuts):
    import forma_amd
    from forma_amd import scenes
    t = np.load(SCENE % wl)
    _, W, H = scenes.WORKLOADS[wl]
    rows = (H + 15) // 16
    edges = [0] + [max(1, min(rows - 1, round(rows * c / 100))) for c in cuts] + [rows]
    img = np.zeros((H, W * 4), np.uint8)
    ctxs = []
    for k in range(len(edges) - 1):
        c = forma_amd.Context(0)
        c.set_geometry(t["x"], t["y"], t["line_slot"]); c.set_geoms(t["geoms"])
        c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
        if len(edges) > 2:
            c.set_band(edges[k], edges[k + 1])
        if k == 0:
            c.register_buffer(img)
        ctxs.append((c, None if len(edges) == 2 else (0, W, edges[k] * 16, min(edges[k + 1] * 16, H))))
    go = threading.Barrier(len(ctxs)); done = threading.Barrier(len(ctxs))
    stop = [False]

    def work(c, crop):
        while True:
            go.wait()
            if stop[0]:
                return
            c.render(W, H, clear=(1, 1, 1, 1), dst=img, crop=crop)
            done.wait()
    th = [threading.Thread(target=work, args=cc, daemon=True) for cc in ctxs[1:]]
    for x in th:
        x.start()

    def frame():
        go.wait()
        ctxs[0][0].render(W, H, clear=(1, 1, 1, 1), dst=img, crop=ctxs[0][1])
        done.wait()
    fps = []
    for _ in range(3):
        for _ in range(8):
            frame()
        t0 = time.perf_counter()
        for _ in range(100):
            frame()
        fps.append(round(100 / (time.perf_counter() - t0), 1))
    stop[0] = True; go.wait()
    print(json.dumps({"fps": fps, "crc": zlib.crc32(img.tobytes()) & 0xFFFFFF, "edges": edges}))
    ctxs[0][0].unregister_buffer(img)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    wl = args[0] if args and not args[0][0].isdigit() else "paris-like-30k-4k"
    cutlists = [a for a in args if a[0].isdigit()]
    if "--child" in sys.argv:
        child(wl, [float(c) for c in cutlists[0].split(",")] if cutlists and cutlists[0] != "0" else [])
    else:
        if not os.path.exists(SCENE % wl):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_fast.py"), "--workload", wl, "--rounds", "0"])
        for rd in range(2):
            for cl in ["0"] + (cutlists or ["30", "20,50"]):
                p = subprocess.run([sys.executable, os.path.abspath(__file__), wl, cl, "--child"], capture_output=True, text=True, timeout=600)
                print("%-12s" % (cl if cl != "0" else "one frame"), [l for l in p.stdout.splitlines() if l.startswith("{")] or p.stderr[-600:], flush=True)
