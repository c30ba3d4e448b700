#!/usr/bin/env python3
"""What ONE of G ranks executes per frame of the multi-GPU layout, on ONE GPU and through the multi-device plumbing
(VERDICT r5 #1b: tools/band_proxy.py times a band through `forma_hip_set_band` — the single-device plumbing — and so leaves the
exchange's bucketing kernels and the chunk-mapped sort out).

    python tools/rank_proxy.py [--workload W] [--ranks 8] [--rank R] [--frames K] [--out FILE]

The rank's frame, exactly the calls `forma_amd/sharding.py::ExchangeFrame` and `csrc/multi.cpp` make per device:
  forma_hip_rasterize_bucket_frame     its share of the LINES (1/G of the pixel segments): k_line_len, k_line_compact, k_rasterize,
                                       then k_owner_count / k_owner_scan / k_owner_scatter -> G send buckets
  [the all-to-all: NOT here — one GPU]  the receive buckets are filled ONCE, before the timed region, with what the G ranks
                                       would send this rank (a stable filter of each line share's stream by owner: what
                                       k_owner_scatter produces; checked against this rank's own send bucket)
  forma_hip_gather_sort_paint_frame    chunk-mapped digit passes over the received buckets, runs, carry, paint of its band

Reported: us per frame with one frame in flight, with three contexts on three host threads (the device-side equivalent of three
frame slots: a slot IS a context that borrows the scene), the per-stage times incl. `exchange_us` (= the bucketing kernels) and
every kernel's own launch events; the band's image is compared with the same rows of the single-GPU frame (bit-identical)."""
import argparse
import json
import os
import statistics
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENE = "/tmp/ab_fast_scene_%s.npz"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="paris-like-30k-4k")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=-1, help="default: the middle rank")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--contexts", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    import forma_amd
    from forma_amd import scenes, sharding
    if not os.path.exists(SCENE % a.workload):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_fast.py"), "--workload", a.workload, "--rounds", "0"])
    t = np.load(SCENE % a.workload)
    _, W, H = scenes.WORKLOADS[a.workload]
    tiles_h = (H + 15) // 16
    G = a.ranks
    r = a.rank if a.rank >= 0 else G // 2
    clear = (1.0, 1.0, 1.0, 1.0)

    def scene(c, x, y, ls):
        c.set_geometry(x, y, ls); c.set_geoms(t["geoms"])
        c.set_styles(t["style_offsets"], t["style_words"], None); c.set_images(t["images"], t["texels"])

    # the single-GPU frame: the plan (bands of equal segment counts, line shares) and the image to compare with
    full = forma_amd.Context(0)
    scene(full, t["x"], t["y"], t["line_slot"])
    want = full.render(W, H, clear=clear)
    segs = full.segments(0)
    lens = full.prepare_lines(W, H)["lengths"].astype(np.int64)
    acc = []
    for _ in range(20):
        acc.append(full.render(W, H, clear=clear, device_only=True, timings=True)[1]["total_us"])
    def full_loop():                                         # best of three blocks, like the rank's loops below (clocks ramp up)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(a.frames):
                full.render(W, H, clear=clear, device_only=True)
            full.sync()
            d = (time.perf_counter() - t0) / a.frames * 1e6
            best = d if best is None else min(best, d)
        return best
    full_us = full_loop()
    full.set_frames_in_flight(3)
    for _ in range(12):
        full.render(W, H, clear=clear, device_only=True)
    full.sync()
    full3_us = full_loop()
    full.close()
    edges = sharding.band_edges(sharding.row_histogram(segs, tiles_h), G)
    cuts = sharding.line_shares(lens, G)
    seg_lo = [int(lens[c - 1]) if c > 0 else 0 for c in cuts]                 # line share q = segments [seg_lo[q], seg_lo[q + 1])
    ty = (segs >> np.uint64(53)).astype(np.int64) - 1
    keep = (ty >= edges[0]) & (ty < edges[-1])
    owner = np.where(keep, np.searchsorted(np.asarray(edges[1:-1], np.int64), ty, side="right"), -1)
    parts, mx = [], 0
    for q in range(G):
        o = owner[seg_lo[q]:seg_lo[q + 1]]
        mx = max(mx, int(np.bincount(o[o >= 0], minlength=G).max()) if (o >= 0).any() else 0)
        parts.append(segs[seg_lo[q]:seg_lo[q + 1]][o == r])                    # what rank q sends rank r: stable, line order
    cap = sharding.pair_capacity(mx)
    recv_host = np.zeros(G * (cap + 1), np.uint64)
    for q, p in enumerate(parts):
        recv_host[q * (cap + 1): q * (cap + 1) + len(p)] = p
        recv_host[q * (cap + 1) + cap] = len(p)                                # header {count | overflow << 32}
    crop = sharding.band_crop(edges, r, W, H)
    gx, gy, gl = sharding.slice_geometry(t["x"], t["y"], t["line_slot"], cuts[r], cuts[r + 1])

    def rank_context():
        c = forma_amd.Context(0)
        scene(c, gx, gy, gl)
        c.exchange_plan(edges, cap)
        send, recv, wpp = c.exchange_views()
        assert wpp == cap + 1
        recv.copy_(torch.from_numpy(recv_host.view(np.int64)))
        torch.cuda.synchronize()
        return c, send

    def frame(c, timings=False, dst=False):
        t1 = c.rasterize_bucket_frame(W, H, timings=timings)
        r2 = c.gather_sort_paint_frame(W, H, clear=clear, crop=crop, timings=timings, device_only=not dst)
        return t1, r2

    c, send = rank_context()
    _, img = frame(c, dst=True)
    own = send[r * (cap + 1): r * (cap + 1) + len(parts[r])].cpu().numpy().view(np.uint64)
    assert np.array_equal(own, parts[r]), "k_owner_scatter's bucket differs from the stable filter of the line share"
    y0, y1 = crop[2], crop[3]
    assert np.array_equal(img[y0:y1], want[y0:y1]), "the rank's band differs from the single-GPU frame"
    for _ in range(6):
        frame(c)
    st, kacc = {}, {}
    for _ in range(30):
        t1, (_, t2) = frame(c, timings=True)
        per = {}
        for name, _s, _t, us in c.kernel_times():
            per[name] = per.get(name, 0.0) + us
        row = {"prepare": t1["prepare_us"], "rasterize": t1["rasterize_us"], "bucketing": t1["exchange_us"], "gather": t2["exchange_us"],
               "sort": t2["sort_us"], "carry": t2["carry_us"], "paint": t2["paint_us"]}
        for k, v in row.items():
            st.setdefault(k, []).append(v)
        for k, v in per.items():
            kacc.setdefault(k, []).append(v)
    stages = {k: round(statistics.median(v), 1) for k, v in st.items()}
    stages["kernels_total"] = round(sum(stages.values()), 1)

    def loop(cc, n):
        for _ in range(n):
            frame(cc)
    best1 = None
    for _ in range(3):
        t0 = time.perf_counter(); loop(c, a.frames)
        d = (time.perf_counter() - t0) / a.frames * 1e6
        best1 = d if best1 is None else min(best1, d)
    # (frame slots of a multi-device context run their digit passes on half the CUs — api.cpp sort_workgroups; three independent
    #  contexts do not know of each other: the switch gives them what the slots' policy would)
    c.close()
    os.environ["FORMA_HIP_DEBUG"] = "sort_cus=128"
    pool = [rank_context()[0] for _ in range(a.contexts)]
    os.environ.pop("FORMA_HIP_DEBUG", None)
    for cc in pool:
        for _ in range(6):
            frame(cc)
    bestF = None
    for _ in range(3):
        ths = [threading.Thread(target=loop, args=(cc, a.frames)) for cc in pool]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        d = (time.perf_counter() - t0) / (a.frames * len(pool)) * 1e6
        bestF = d if bestF is None else min(bestF, d)
    out = {"workload": a.workload, "canvas": [W, H], "ranks": G, "rank": r, "band_rows": [edges[r], edges[r + 1]], "pair_capacity": cap,
           "n_segments_line_share": seg_lo[r + 1] - seg_lo[r], "n_segments_band": int(sum(len(p) for p in parts)),
           "image": "band rows bit-identical to the single-GPU frame", "stages_us_one_in_flight": stages,
           "exchange_us": round(stages["bucketing"] + stages["gather"], 1),
           "exchange_us_what": "k_owner_count + k_owner_scan + k_owner_scatter on the line share (+ k_gather_chunks when the received buckets are "
                               "materialised; default: sorted where they lie through the chunk map) — the collective itself is NOT in this proxy",
           "kernels_us_one_in_flight": {k: round(statistics.median(v), 1) for k, v in kacc.items()},
           "kernels_us_note": "forma_hip_kernel_times of the owner's half (the last timed call); the line share's kernels are in the stages",
           "rank_frame_us": {"F=1": round(best1, 1), "%d contexts" % len(pool): round(bestF, 1)},
           "full_frame_us": {"F=1": round(full_us, 1), "F=3": round(full3_us, 1), "kernels": round(statistics.median(acc), 1)},
           "speedup_vs_full": {"F=1": round(full_us / best1, 2), "pipelined": round(full3_us / bestF, 2)}, "frames": a.frames}
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
    for cc in pool:
        cc.close()


if __name__ == "__main__":
    main()
