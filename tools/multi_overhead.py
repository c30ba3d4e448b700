#!/usr/bin/env python3
"""What the multi-device frame costs over the plain one on ONE GPU (world of one, where all of it is overhead): the plain
context, the exchange path without a collective (FORMA_HIP_DEBUG=xchg=copy) and with RCCL (ncclAllToAll with itself).
    python tools/multi_overhead.py [workload]        (scene tables: /tmp/ab_fast_scene_<workload>.npz, built by tools/ab_fast.py)"""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "paris-like-30k-4k"

def child(kind):
    import torch  # noqa: F401
    import forma_amd
    from forma_amd import scenes
    t = np.load("/tmp/ab_fast_scene_%s.npz" % wl)
    _, W, H = scenes.WORKLOADS[wl]
    c = forma_amd.Context(0) if kind == "plain" else forma_amd.Context(devices=[0] if kind != "two" else [0, 0])
    c.set_geometry(t["x"], t["y"], t["line_slot"]); c.set_geoms(t["geoms"])
    c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])
    for _ in range(6):
        c.render(W, H, clear=(1, 1, 1, 1), device_only=True)
    n = 60
    t0 = time.perf_counter()
    for _ in range(n):
        c.render(W, H, clear=(1, 1, 1, 1), device_only=True)
    dt = (time.perf_counter() - t0) / n
    _, tm = c.render(W, H, clear=(1, 1, 1, 1), device_only=True, timings=True)
    print(json.dumps({"kind": kind, "us_per_frame": round(dt * 1e6, 1), "stages_total_us": round(tm["total_us"], 1), "exchange_us": round(tm["exchange_us"], 1)}))
    c.close()

if "--child" in sys.argv:
    child(sys.argv[sys.argv.index("--child") + 1])
else:
    for kind, env in (("plain", {}), ("copy1", {"FORMA_HIP_DEBUG": "force_exchange,xchg=copy"}), ("rccl1", {"FORMA_HIP_DEBUG": "force_exchange"}),
                      ("two", {})):
        e = dict(os.environ, **env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), wl, "--child", kind], env=e, capture_output=True, text=True, timeout=300)
        print([l for l in p.stdout.splitlines() if l.startswith("{")] or p.stderr[-400:])
