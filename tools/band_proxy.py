#!/usr/bin/env python3
"""The multi-GPU band frame on ONE GPU (profiles/r04_band_proxy.json): a 1/8 band of a workload (`forma_hip_set_band`: the line
kernels cull to the band, the painter paints it) rendered with F frames in flight, against the full frame of the same build.

    python tools/band_proxy.py [--workload W] [--band r0,r1] [--slots 1,2,3] [--frames K] [--out FILE]

The scene tables come from /tmp/ab_fast_scene_<workload>.npz (tools/ab_fast.py parks them; built here if missing).  The render
loop calls forma_hip_render through ctypes with prebuilt arguments: what is timed is the library, not numpy.  Per configuration:
us per frame of K pipelined device-resident frames, the per-stage device times of one frame in flight, and — the frames of a
one-row band, where the device never is the limit — the HOST time of a render call."""
import argparse
import ctypes as C
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENE = "/tmp/ab_fast_scene_%s.npz"


def run(ctx, W, H, crop, frames, slots):
    from forma_amd._lib import RectT
    L, h = ctx._L, ctx._h
    ch = np.asarray((0, 1, 2, 3), np.uint8); cl = np.asarray((1, 1, 1, 1), np.float32)
    rect = RectT(*crop) if crop else None
    args = (h, None, W, H, W * 4, ch.ctypes.data_as(C.c_void_p), cl.ctypes.data_as(C.c_void_p), C.addressof(rect) if rect else None, -1, None)
    ctx.set_frames_in_flight(slots)
    for _ in range(3 * slots + 3):
        assert L.forma_hip_render(*args) == 0
    ctx.sync()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(frames):
            L.forma_hip_render(*args)
        ctx.sync()
        dt = (time.perf_counter() - t0) / frames * 1e6
        best = dt if best is None else min(best, dt)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="paris-like-30k-4k")
    ap.add_argument("--band", default=None, help="r0,r1 (default: the middle eighth of the tile rows)")
    ap.add_argument("--slots", default="1,2,3,4")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import forma_amd
    from forma_amd import scenes
    if not os.path.exists(SCENE % a.workload):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_fast.py"), "--workload", a.workload, "--rounds", "0"])
    t = np.load(SCENE % a.workload)
    _, W, H = scenes.WORKLOADS[a.workload]
    tiles_h = (H + 15) // 16
    r0, r1 = (int(v) for v in a.band.split(",")) if a.band else (tiles_h * 7 // 16, tiles_h * 7 // 16 + (tiles_h + 7) // 8)
    out = {"workload": a.workload, "canvas": [W, H], "band_rows": [r0, r1], "frames": a.frames, "runs": []}

    def fresh():
        c = forma_amd.Context(0)
        c.set_geometry(t["x"], t["y"], t["line_slot"]); c.set_geoms(t["geoms"])
        c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
        return c

    for what, band in (("full frame", None), ("band", (r0, r1)), ("one-row band (host cost of a call)", (r0, r0 + 1))):
        c = fresh()
        crop = None
        if band:
            c.set_band(*band)
            crop = (0, W, band[0] * 16, min(band[1] * 16, H))
        for _ in range(4):
            c.render(W, H, clear=(1, 1, 1, 1), crop=crop, device_only=True)
        acc, kacc = {}, {}
        for _ in range(30):
            _, tm = c.render(W, H, clear=(1, 1, 1, 1), crop=crop, device_only=True, timings=True)
            for k, v in tm.items():
                acc.setdefault(k, []).append(v)
            per = {}
            for name, _st, _t0, us in c.kernel_times():       # every kernel's own launch events (forma_hip_kernel_times)
                per[name] = per.get(name, 0.0) + us
            for k, v in per.items():
                kacc.setdefault(k, []).append(v)
        st = {k[:-3]: round(statistics.median(acc[k]), 1) for k in ("prepare_us", "rasterize_us", "sort_us", "carry_us", "paint_us", "total_us")}
        row = {"what": what, "rows": list(band) if band else [0, tiles_h], "n_segments": int(statistics.median(acc["n_segments"])),
               "stages_us_one_in_flight": st, "kernels_us_one_in_flight": {k: round(statistics.median(v), 1) for k, v in kacc.items()},
               "us_per_frame": {}}
        row["gaps_us_one_in_flight"] = round(st["total"] - sum(row["kernels_us_one_in_flight"].values()), 1)
        for s in (int(v) for v in a.slots.split(",")):
            row["us_per_frame"]["F=%d" % s] = round(run(c, W, H, crop, a.frames, s), 1)
        out["runs"].append(row)
        print(json.dumps(row), flush=True)
        c.close()
    full = out["runs"][0]["us_per_frame"]; band = out["runs"][1]["us_per_frame"]
    out["band_over_full"] = {k: round(band[k] / full[k], 3) for k in band}
    out["best"] = {"full_us": min(full.values()), "band_us": min(band.values()), "ratio": round(min(band.values()) / min(full.values()), 3)}
    print(json.dumps(out["best"]))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
