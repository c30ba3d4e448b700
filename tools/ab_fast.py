#!/usr/bin/env python3
"""Same-box A/B of builds of libforma_hip.so in seconds per variant (boxes of the pool differ by 10-25 %, so numbers from two
gpurun calls are not comparable; bench.py spends most of a short run building the scene in Python).

    python tools/ab_fast.py [--workload W] [--rounds R] [--frames K] base.bin new.bin new.bin@ENV=1 ...
                                                                   (files under forma_amd/csrc/variants/, optional env switches)

The parent builds the scene once through the product API and parks its flat tables in /tmp; every (round, variant) is a
child process that loads ONE build (FORMA_HIP_LIB), uploads the tables and reports: per-stage device times and the radix
pass (HIP events, one frame in flight, median of K frames), frames/s of K synchronous render calls and of K calls with three
frames in flight.  No torch, no oracle: a child takes about three seconds."""
import argparse
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


def child(args):
    import forma_amd
    from forma_amd import scenes
    t = np.load(SCENE % args.workload)
    _, W, H = scenes.WORKLOADS[args.workload]
    c = forma_amd.Context(0)
    c.set_geometry(t["x"], t["y"], t["line_slot"]); c.set_geoms(t["geoms"])
    c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
    clear = (1.0, 1.0, 1.0, 1.0)
    crop = None
    if os.environ.get("AB_BAND"):                         # "r0,r1": a multi-GPU band on one GPU (foreign segments culled, band painted)
        r0, r1 = (int(v) for v in os.environ["AB_BAND"].split(","))
        c.set_band(r0, r1)
        crop = (0, W, r0 * 16, min(r1 * 16, H))
    _render = c.render
    c.render = lambda *a, **k: _render(*a, crop=crop, **k)
    for _ in range(4):
        c.render(W, H, clear=clear, device_only=True)
    acc, kacc = {}, {}
    for _ in range(args.frames):
        _, tm = c.render(W, H, clear=clear, device_only=True, timings=True)
        for k, v in tm.items():
            acc.setdefault(k, []).append(v)
        per = {}
        for name, _st, _t0, us in c.kernel_times():           # every kernel's own launch events, summed per name and frame
            per[name] = per.get(name, 0.0) + us
        for k, v in per.items():
            kacc.setdefault(k, []).append(v)
    med = {k: statistics.median(v) for k, v in acc.items()}
    kern = {k[2:] if k.startswith("k_") else k: round(statistics.median(v), 1) for k, v in kacc.items()}
    t0 = time.perf_counter()
    for _ in range(args.frames):
        c.render(W, H, clear=clear, device_only=True)
    c.sync()
    fps1 = args.frames / (time.perf_counter() - t0)
    slots = int(os.environ.get("AB_INFLIGHT", "3"))                   # (AB_INFLIGHT: slots for the pipelined rate)
    c.set_frames_in_flight(slots)
    for _ in range(3 * slots + 3):                                    # every slot: its synchronous frame, its first read-back-free one, one more
        c.render(W, H, clear=clear, device_only=True)
    c.sync()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        c.render(W, H, clear=clear, device_only=True)
    c.sync()
    fps3 = args.frames / (time.perf_counter() - t0)
    img = c.read_image(W, H)
    print(json.dumps({"fps1": round(fps1, 1), "fps3": round(fps3, 1), "pass_us": round(med["sort_pass_us"], 1),
                      "stages": {k[:-3]: round(med[k], 1) for k in ("prepare_us", "rasterize_us", "sort_us", "carry_us", "paint_us", "total_us")},
                      "crc": int(np.bitwise_xor.reduce(img.view(np.uint32).reshape(-1))) & 0xFFFFFFFF,
                      "n": int(med["n_segments"]), "kern": kern}))
    c.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="paris-like-30k-4k")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()
    if args.child:
        return child(args)
    if not os.path.exists(SCENE % args.workload):
        from forma_amd import api, scenes
        fn, W, H = scenes.WORKLOADS[args.workload]
        r = api.Renderer(0)
        r.render(fn(), api.BufferBuilder(np.zeros(W * H * 4, np.uint8), api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
        np.savez(SCENE % args.workload, **r.host_tables)
        r._ctx.close()
    vdir = os.path.join(ROOT, "forma_amd", "csrc", "variants")
    rows = {}
    for rd in range(args.rounds):
        for v in args.variants:
            lib, *sets = v.split("@")                          # "build.bin@ENV=VALUE@ENV2=VALUE2": environment switches of one build
            env = dict(os.environ, FORMA_HIP_LIB=os.path.join(vdir, lib))
            for kv in sets:
                k, _, val = kv.partition("=")
                env[k] = val
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--workload", args.workload, "--frames", str(args.frames)],
                               env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("%-18s FAILED %s" % (v, (p.stderr or p.stdout)[-300:]))
                continue
            d = json.loads(line[-1])
            rows.setdefault(v, []).append(d)
            print("%-18s fps1 %7.1f fps3 %7.1f pass %6.1f  %s crc %08x" % (v, d["fps1"], d["fps3"], d["pass_us"], d["stages"], d["crc"]), flush=True)
            if os.environ.get("AB_KERNELS"):
                print("    kernels: " + " ".join("%s %.1f" % kv for kv in d.get("kern", {}).items()), flush=True)
    print("---- medians")
    for v, ds in rows.items():
        print("%-18s fps1 %7.1f fps3 %7.1f pass %6.1f total %6.1f" % (v, statistics.median(x["fps1"] for x in ds), statistics.median(x["fps3"] for x in ds),
                                                                    statistics.median(x["pass_us"] for x in ds), statistics.median(x["stages"]["total"] for x in ds)))
    crcs = {ds[0]["crc"] for ds in rows.values()}
    print("images identical across variants:", len(crcs) == 1)


if __name__ == "__main__":
    main()
