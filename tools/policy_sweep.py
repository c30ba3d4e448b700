#!/usr/bin/env python3
"""Policy sweep: is every default choice of the library the best (or within a few percent of the best) AWAY from the four bench
workloads it was measured on?  (VERDICT r5 #7 / "What's weak" #12: each heuristic was a measured point, none a measured curve.)

    python tools/policy_sweep.py [--canvases 720p,1080p,1440p,4k,8k] [--families cubics,paris,triangles,circles]
                                 [--slots 1,2,3,4] [--frames K] [--out profiles/r06_policy_sweep.json]

Scene families = the four bench scenes' flat tables (tools/ab_fast.py parks them in /tmp; built here if missing) with their
coordinates scaled to the canvas (the styles stay: a scaled gradient is still a gradient), so every family exists at every
canvas: cubics (1 000 opaque cubics), paris (30 000 mixed layers), triangles (19 400 small opaque triangles), circles (20 000
translucent discs).  Per (canvas, family, slots) cell every schedule switch of csrc/debug.h is forced both ways in a context of
its own (FORMA_HIP_DEBUG is read at context creation) and K device-resident frames are timed after the set-up frames:
    default | strip_tiles=0 / =huge | paint_quad=0 / =2 | no_order / order_thr=32768 | runs_chain=0 / =1 | runs_blk=0 / =1 | sort_cus=0 / =128 |
    carry_half=0 / =2 | carry_covl=0
`loss` of a cell = 1 - fps(default) / max over all settings; the report lists the cells where the default loses more than 5 %.
Images are checked equal across the settings of a cell (xor-fold of the last frame)."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENE = "/tmp/ab_fast_scene_%s.npz"
CANVAS = {"720p": (1280, 720), "1080p": (1920, 1080), "1440p": (2560, 1440), "4k": (3840, 2160), "8k": (8192, 8192)}
FAMILY = {"cubics": "cubics-1080p", "paris": "paris-like-30k-4k", "triangles": "triangles-10m-8k", "circles": "circles-20k"}
SETTINGS = ["", "strip_tiles=0", "strip_tiles=100000000", "paint_quad=0", "paint_quad=2", "no_order", "order_thr=32768",
            "runs_chain=0", "runs_chain=1", "runs_blk=0", "runs_blk=1", "sort_cus=0", "sort_cus=128", "carry_half=0", "carry_half=2", "carry_covl=0"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--canvases", default="720p,1080p,1440p,4k,8k")
    ap.add_argument("--families", default="cubics,paris,triangles,circles")
    ap.add_argument("--slots", default="1,2,3,4")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--settings", default=None, help="comma-free list separated by ';' (default: all)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import ctypes as C
    import forma_amd
    from forma_amd import scenes
    settings = a.settings.split(";") if a.settings is not None else SETTINGS
    cells = []
    t_all = time.perf_counter()
    for fam in a.families.split(","):
        wl = FAMILY[fam]
        if not os.path.exists(SCENE % wl):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_fast.py"), "--workload", wl, "--rounds", "0"])
        t = np.load(SCENE % wl)
        _, W0, H0 = scenes.WORKLOADS[wl]
        for cv in a.canvases.split(","):
            W, H = CANVAS[cv]
            x = (t["x"] * np.float32(W / W0)).astype(np.float32); y = (t["y"] * np.float32(H / H0)).astype(np.float32)
            for slots in (int(v) for v in a.slots.split(",")):
                row = {"family": fam, "canvas": cv, "slots": slots, "fps": {}}
                crcs = set()
                for st in settings:
                    os.environ["FORMA_HIP_DEBUG"] = st
                    c = forma_amd.Context(0)
                    try:
                        c.set_geometry(x, y, t["line_slot"]); c.set_geoms(t["geoms"])
                        c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
                        L, h = c._L, c._h
                        ch = np.asarray((0, 1, 2, 3), np.uint8); cl = np.asarray((1, 1, 1, 1), np.float32)
                        args = (h, None, W, H, W * 4, ch.ctypes.data_as(C.c_void_p), cl.ctypes.data_as(C.c_void_p), None, -1, None)
                        if slots > 1:
                            c.set_frames_in_flight(slots)
                        for _ in range(3 * slots + 4):
                            rc = L.forma_hip_render(*args)
                            assert rc == 0, (rc, st, fam, cv)
                        c.sync()
                        best = 0.0
                        for _ in range(2):
                            t0 = time.perf_counter()
                            for _ in range(a.frames):
                                L.forma_hip_render(*args)
                            c.sync()
                            best = max(best, a.frames / (time.perf_counter() - t0))
                        row["fps"][st or "default"] = round(best, 1)
                        if "n_segments" not in row:
                            _, tm = c.render(W, H, clear=(1, 1, 1, 1), device_only=True, timings=True)
                            row["n_segments"] = int(tm["n_segments"])
                        img = c.read_image(W, H)
                        crcs.add(int(np.bitwise_xor.reduce(img.view(np.uint32).reshape(-1))))
                    finally:
                        c.close()
                best_st = max(row["fps"], key=row["fps"].get)
                row["best"] = best_st
                row["default_loss"] = round(1.0 - row["fps"]["default"] / row["fps"][best_st], 4) if "default" in row["fps"] else None
                row["images_identical"] = len(crcs) == 1
                cells.append(row)
                print(json.dumps(row), flush=True)
    os.environ.pop("FORMA_HIP_DEBUG", None)
    bad = [c for c in cells if c["default_loss"] is not None and c["default_loss"] > 0.05]
    out = {"what": "frames/s of K device-resident frames per (family, canvas, frame slots) cell with every schedule switch forced both ways; "
                   "default_loss = 1 - fps(default) / best", "frames": a.frames, "settings": settings, "cells": cells,
           "cells_where_default_loses_more_than_5_percent": [{k: c[k] for k in ("family", "canvas", "slots", "best", "default_loss")} for c in bad],
           "worst_default_loss": max((c["default_loss"] or 0.0) for c in cells) if cells else None,
           "all_images_identical": all(c["images_identical"] for c in cells), "seconds": round(time.perf_counter() - t_all, 1)}
    print(json.dumps({k: out[k] for k in ("cells_where_default_loses_more_than_5_percent", "worst_default_loss", "all_images_identical", "seconds")}))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
