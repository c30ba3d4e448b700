#!/usr/bin/env python3
"""bench.py — frames/s and Mpixel-segments/s of the forma raster hot path on MI355X.

A "step" is one `Renderer::render` frame (prepare lines -> rasterize -> radix sort -> carry pre-pass
-> per-tile paint) of a synthetic scene whose geometry, layer table and styles are already resident
in HBM; the image stays device-resident (the PCIe-inclusive rate is reported separately, never as
`value`).  One process per GPU.  For N > 1 the default (`--mode frames`) is frame-parallel: the unit of
work is a frame, every GPU renders whole frames, per-GPU work is fixed (weak scaling) and nothing is
exchanged on the data path; `--mode bands` splits ONE frame into tile-row bands (SURVEY.md §8e, strong
scaling).  `value` is the whole-job aggregate: frames completed by all GPUs / max-over-ranks wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME | --svg FILE [--svg-scale S]] [--animated]

`--svg` renders a real SVG document (e.g. the reference's paris-30k.svg, which is not in its checkout) through
forma_amd/svg.py instead of the labelled stand-in; `--animated` adds the BASELINE config-5 leg (spaceship-like scene with
and without the buffer-layer cache) under "animated".  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="paris-like-30k-4k")
    ap.add_argument("--mode", default="frames", choices=["frames", "bands", "exchange"],
                    help="N > 1: 'frames' = every GPU renders whole frames (weak scaling, no exchange); "
                         "'bands' = ONE frame split into tile-row bands across the GPUs, scene replicated (strong scaling, no "
                         "exchange); 'exchange' = ONE frame: every GPU rasterizes 1/N of the lines, an RCCL all-to-all moves the "
                         "pixel segments to the GPU that owns their tile row, which sorts and paints its band (strong scaling)")
    ap.add_argument("--svg", default=None, metavar="FILE",
                    help="render this SVG file (e.g. the real paris-30k.svg) on a 3840x2160 canvas instead of a synthetic "
                         "workload; loaded by forma_amd.svg like the reference demo's `svg` mode")
    ap.add_argument("--svg-scale", type=float, default=1.0)
    ap.add_argument("--animated", action="store_true",
                    help="also measure BASELINE config 5: the animated spaceship-like 4K scene, with and without the "
                         "buffer-layer cache (per-tile damage tracking); reported under \"animated\", never as `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        if rank == 0:
            print(f"WORLD_SIZE={world} does not match --gpus {args.gpus}", file=sys.stderr)
        args.gpus = world
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    import forma_amd
    from forma_amd import api, scenes, sharding

    if args.svg:
        from forma_amd import svg as svg_loader
        width, height = 3840, 2160
        comp = svg_loader.Svg(args.svg, args.svg_scale).compose(api.Composition())
        args.workload = "svg:" + os.path.basename(args.svg) + f" x{args.svg_scale:g}"
    else:
        build_fn, width, height = scenes.WORKLOADS[args.workload]
        comp = build_fn()
    renderer = api.Renderer(device=local)
    ctx = renderer._ctx
    tiles_h = (height + 15) // 16
    image = np.zeros((height, width * 4), np.uint8)
    layout = api.LinearLayout(width, width * 4, height)
    buf = api.BufferBuilder(image.reshape(-1), layout).build()
    clear = api.Color(1.0, 1.0, 1.0, 1.0)
    # first frame through the public API: flattens on the GPU, uploads tables, leaves everything resident
    renderer.render(comp, buf, api.RGBA, clear, None, timings=True)
    t_full = renderer.last_timings
    n_segments_full = t_full["n_segments"]

    crop = None
    row0, row1 = 0, tiles_h
    if world > 1 and args.mode == "bands":
        # tile-row bands balanced on the per-row pixel-segment histogram of the full frame
        hist = sharding.row_histogram(ctx.segments(0), tiles_h)
        edges = sharding.agree_on_bands(dist, hist, world, device="cuda")
        row0, row1 = edges[rank], edges[rank + 1]
        ctx.set_band(row0, row1)
        crop = sharding.band_crop(edges, rank, width, height)

    channels = api.RGBA
    clr = (clear.r, clear.g, clear.b, clear.a)

    def frame(timings=False):
        return ctx.render(width, height, channels=channels, clear=clr, crop=crop, device_only=True, timings=timings)

    step = frame
    if args.mode == "exchange":
        # line-sharded rasterization + all-to-all of pixel segments to their tile-row owners (sharding.exchange_segments)
        tab = renderer.host_tables
        hist = sharding.row_histogram(ctx.segments(0), tiles_h)
        edges = sharding.agree_on_bands(dist, hist, world, device="cuda") if world > 1 else [0, tiles_h]
        row0, row1 = edges[rank], edges[rank + 1]
        crop = sharding.band_crop(edges, rank, width, height)
        cuts = sharding.line_shares(ctx.prepare_lines(width, height)["lengths"], world)
        ctx.set_geometry(*sharding.slice_geometry(tab["x"], tab["y"], tab["line_slot"], cuts[rank], cuts[rank + 1]))
        empty = torch.empty(0, dtype=torch.int64, device=torch.device("cuda", local))

        def frame(timings=False):                           # noqa: F811
            t1 = ctx.rasterize_frame(width, height, timings=timings)
            seg = ctx.unsorted_view()
            recv = sharding.exchange_segments(dist, empty if seg is None else seg, edges, world, out_alloc=ctx.reserve_view)
            torch.cuda.synchronize()                        # the exchange ran on torch's stream; the context has its own
            r = ctx.sort_paint_frame(int(recv.numel()), width, height, channels=channels, clear=clr, crop=crop,
                                     device_only=True, timings=timings)
            if not timings:
                return r
            t2 = r[1]
            for k in ("prepare_us", "rasterize_us"):
                t2[k] = t1[k]
            t2["total_us"] = t2["total_us"] + t1["total_us"]
            return r[0], t2

        step = frame

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = sharding.max_over_ranks(dist, elapsed, device="cuda")
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_step = world if (world > 1 and args.mode == "frames") else 1      # frames mode: one whole frame per GPU per step
    fps = frames_per_step * args.steps / elapsed

    # ---- per-stage device times + roofline of the radix pass (HIP events on the context's stream) ----------
    stage = {}
    reps = 10
    for _ in range(reps):
        _, t = frame(timings=True)
        for k, v in t.items():
            stage[k] = stage.get(k, 0.0) + float(v) / reps
    n_local = int(round(stage["n_segments"]))                 # (exchange mode: the segments this rank received)
    passes = int(round(stage["n_sort_passes"]))
    pass_us = stage["sort_pass_us"]
    algo_bytes_per_pass = 16.0 * n_local                       # 8 B read + 8 B written per key per digit pass
    achieved = (algo_bytes_per_pass / (pass_us * 1e-6) / 1e9) if pass_us > 0 else 0.0
    # HBM bytes per launch from the PMC counters (collected in separate rocprofv3 passes and corrected as the MI355X guide
    # prescribes; bench.py cannot run under two profilers at once, so the committed summary of the same command is read)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01s_pmc_traffic.json")) as f:     # tools/pmc_traffic.py, current kernels
            pmc = json.load(f)
        if args.workload == "paris-like-30k-4k" and world == 1:
            traffic = pmc["kernels"][pmc["roofline_kernel"]]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": "k_onesweep: one radix digit pass (LSB, 8-bit digits over live key bits, u64 keys, chained scan)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes_per_pass, "avg_launch_us": round(pass_us, 2),
                "passes": passes}

    # PCIe-inclusive frame (image copied into caller memory) — reported, never `value`
    sync_all()
    t1 = time.perf_counter()
    if args.mode == "exchange":
        fps_d2h = 0.0                                       # (not measured in this mode)
    else:
        for _ in range(5):
            ctx.render(width, height, channels=channels, clear=clr, crop=crop, dst=image.reshape(-1), stride=width * 4)
        torch.cuda.synchronize()
        fps_d2h = 5 / (time.perf_counter() - t1)

    out = {
        "metric": "frames/sec (sorted+painted, device-resident) + Mpixel-segments/sec",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if (world > 1 and args.mode in ("bands", "exchange")) else "weak", "vs_baseline": None,
        "dtype": "u64 segments / f64+f32 rasterizer / f32 painter", "data": "synthetic",
        "mpixel_segments_per_s": round(n_segments_full * fps / 1e6, 1),
        "fps_including_d2h": round(fps_d2h, 2),
        "config": {"workload": args.workload + (" (labelled stand-in: paris-30k.svg is not in the reference checkout)"
                                                 if args.workload.startswith("paris") else ""),
                   "canvas": [width, height], "layers": len(comp), "pixel_segments": int(n_segments_full),
                   "sharding": "none" if world == 1 else (
                       f"tile-row bands x{world} of ONE frame, replicated scene, band culling, no data-path collective"
                       if args.mode == "bands" else
                       f"lines / {world} rasterized per GPU, RCCL all-to-all of pixel segments to tile-row owners, band-local sort + paint"
                       if args.mode == "exchange" else
                       f"frame-parallel x{world}: every GPU renders whole 3840x2160 frames of the workload (units = frames), no exchange"),
                   "band_rows": [row0, row1]},
        "stages_us": {k: round(stage[k], 1) for k in ("prepare_us", "rasterize_us", "sort_us", "carry_us", "paint_us", "total_us")},
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and args.animated:
        out["animated"] = animated_leg(local)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(renderer, width, height, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def animated_leg(local, frames=240):
    """BASELINE config 5 on one GPU: 3840x2160, 400 static + 121 moving layers, 60 Hz animation.  Every frame uploads the
    layer table (transforms) and the per-order `unchanged` bytes exactly like `Renderer::render` would, then renders
    device-resident with cache 0 (damage tracking) or without a cache (everything repainted)."""
    import torch
    from forma_amd import api, scenes
    comp, moving, state = scenes.spaceship()
    renderer = api.Renderer(device=local)
    W, H = state["size"]
    img = np.zeros(W * H * 4, np.uint8)
    renderer.render(comp, api.BufferBuilder(img, api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(0, 0, 0, 1), None)
    ctx = renderer._ctx
    t = renderer.host_tables
    geoms = t["geoms"].copy()
    slot_of_order = {int(geoms[i]["order"]): i for i in range(len(geoms)) if geoms[i]["order"] != 0xFFFFFFFF}
    slots = np.array([slot_of_order[o] for o in moving])
    unchanged = np.ones(len(t["style_offsets"]), np.uint8)
    unchanged[np.array(moving)] = 0
    res = {}
    for label, cache_id, unch in (("no_cache", -1, None), ("with_cache", 0, unchanged)):
        painted = []
        for i in range(frames + 10):
            if i == 10:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            geoms["flags"][slots] = 1
            geoms["xf"][slots] = scenes.spaceship_transforms(state, i / 60.0)
            ctx.set_geoms(geoms)
            ctx.set_styles(t["style_offsets"], t["style_words"], unch if i > 0 else None)
            ctx.render(W, H, clear=(0, 0, 0, 1), cache_id=cache_id, device_only=True)
        torch.cuda.synchronize()
        res[label] = round(frames / (time.perf_counter() - t0), 1)
    # damaged-tile fraction: tiles the cached frames actually rewrite, from a few frames rendered into a host buffer
    host = np.zeros((H, W * 4), np.uint8)
    damaged = []
    for i in range(frames + 10, frames + 18):
        geoms["xf"][slots] = scenes.spaceship_transforms(state, i / 60.0)
        ctx.set_geoms(geoms)
        ctx.set_styles(t["style_offsets"], t["style_words"], unchanged)
        _, tm = ctx.render(W, H, clear=(0, 0, 0, 1), cache_id=0, dst=host, timings=True)
        damaged.append(tm["n_tiles_written"])
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    return {"workload": "spaceship-like-4k (400 static + 121 moving layers, 60 Hz transforms)", "frames": frames,
            "fps_no_cache": res["no_cache"], "fps_with_cache": res["with_cache"], "unit": "frames/s, device-resident, "
            "including the per-frame layer-table upload", "damaged_tile_fraction": round(float(np.mean(damaged)) / tiles, 4)}


def cpu_baseline(renderer, width, height, budget_s):
    """The CPU oracle (C++ restatement of forma's CPU backend: OpenMP over lines / pixel segments / tile rows, parallel
    stable radix sort, parallel prefix sum) timed on the host cores on the SAME scene tables the GPU rendered.  The thread
    count is the fastest of a short sweep (all hardware threads is rarely the fastest on a many-core host: the sort and
    the scan are memory-bound).  Reported baseline only — a restatement, not forma's own Rayon/SIMD build (no Rust
    toolchain in this image)."""
    from oracle import oracle as orc
    hw = orc.lib().oracle_max_threads()
    t = renderer.host_tables
    o = orc.Oracle(threads=1)
    o.set_geometry(t["x"], t["y"], t["line_slot"]); o.set_geoms(t["geoms"])
    o.set_styles(t["style_offsets"], t["style_words"], None); o.set_images(t["images"], t["texels"])
    cands = sorted({c for c in (8, 16, 32, 48, 64, 96, 128, hw) if c <= hw} or {hw})
    t_start = time.perf_counter()
    sweep = {}
    for c in cands:
        o.set_threads(c)
        o.time_frame(width, height, 1)                      # first touch of this thread count's buffers / thread pool
        sweep[c] = sum(o.time_frame(width, height, 1).values())
        if time.perf_counter() - t_start > budget_s * 0.5:
            break
    best = min(sweep, key=sweep.get)
    o.set_threads(best)
    per = sweep[best]
    left = max(1.0, budget_s - (time.perf_counter() - t_start))
    iters = max(3, min(40, int(left / max(per, 1e-3))))
    tm = o.time_frame(width, height, iters)
    per = sum(tm.values())
    return {"value": round(1.0 / per, 3), "unit": "frames/s", "cores": best, "kind": "port", "host_threads": hw,
            "sample": f"{iters} full frames of the same workload (C++ restatement of the CPU backend, OpenMP on {best} of {hw} "
                      f"hardware threads = the fastest of the sweep {sorted(sweep)}; parallel stable radix sort and prefix sum)",
            "thread_sweep_ms": {str(k): round(v * 1e3, 1) for k, v in sweep.items()},
            "stages_ms": {k: round(v * 1e3, 2) for k, v in tm.items()}}


if __name__ == "__main__":
    main()
