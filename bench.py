#!/usr/bin/env python3
"""bench.py — frames/s and Mpixel-segments/s of the forma raster hot path on MI355X.

A "step" is one `Renderer::render` frame (prepare lines -> rasterize -> radix sort -> carry pre-pass -> per-tile paint)
of a synthetic scene whose geometry, layer table and styles are already resident in HBM; the image stays device-resident
(the PCIe-inclusive rate is reported separately, never as `value`).  One process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME | --svg FILE [--svg-scale S]]
                    [--in-flight F] [--mode exchange|bands|frames] [--animated] [--no-cpu-baseline]

N = 1: `value` = frames completed / wall time of exactly K frames with `--in-flight` F frames in flight (default 3): F
renderer contexts on the one GPU, each with its own HIP stream and buffers, each fed by its own host thread.  Every frame
does all of the work (nothing is shared between contexts but the scene description); the kernels of this path are
latency-bound, so frames that overlap fill the machine — what a frame server or an animation export does.  The rate of
ONE context rendering frame after frame (`fps_one_frame_in_flight`, with its per-frame latency and per-stage device times)
is always reported next to it, as is the spread over five more blocks of K frames.

N > 1 (default `--mode exchange`, strong scaling, the north star's layout): ONE frame is split — every GPU rasterizes 1/N
of the lines, HIP kernels bucket the pixel segments by tile-row owner, one RCCL all-to-all over xGMI moves them, the owner
sorts and paints its band.  `--mode bands` replicates the scene and culls by band (no exchange); `--mode frames` renders
whole frames on every GPU (weak scaling).  `value` is always the whole-job aggregate / max-over-ranks wall time.

`roofline` (the radix digit pass, HBM-bound) and `stages_us` come from a second timed region of the same run — K more
frames with ONE frame in flight and HIP events recorded at the stage boundaries on the context's stream: with several frames
in flight the kernels of different frames time-share the chip and a launch duration is no longer the kernel's own.
`roofline.traffic` is NOT measured in the run: it is read from profiles/r02_pmc_summary.json (separate rocprofv3 --pmc passes
of the same build, tools/pmc_round.py) and labelled so.  `cpu_baseline` = the C++ oracle on the same scene tables, N = 1 only.

Rehearsal switches for single-GPU boxes (never set by the driver): FORMA_BENCH_MODE_AT_1=1 runs the sharded mode with one
rank; FORMA_BENCH_BACKEND=gloo + FORMA_BENCH_ONE_DEVICE=1 run N ranks on one GPU without RCCL.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2   # wave64 VALU instructions/ns the chip can issue: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles
PMC_FILE = os.path.join("profiles", "r02_pmc_summary.json")   # tools/pmc_round.py: separate rocprofv3 --pmc passes of this command


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--workload", default="paris-like-30k-4k")
    ap.add_argument("--in-flight", type=int, default=3, help="N = 1: renderer contexts rendering concurrently (frames in flight)")
    ap.add_argument("--mode", default="exchange", choices=["exchange", "bands", "frames"],
                    help="N > 1: 'exchange' = ONE frame, lines / N rasterized per GPU, all-to-all of pixel segments to the "
                         "tile-row owners (strong scaling); 'bands' = ONE frame, scene replicated, band culling, no exchange "
                         "(strong scaling); 'frames' = every GPU renders whole frames (weak scaling)")
    ap.add_argument("--svg", default=None, metavar="FILE", help="render this SVG (e.g. the real paris-30k.svg) instead of a synthetic workload")
    ap.add_argument("--svg-scale", type=float, default=1.0)
    ap.add_argument("--animated", action="store_true", help="also measure BASELINE config 5 (deterministic spaceship, 4K, damage cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
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
    # Rehearsal switches (never set by the driver): FORMA_BENCH_BACKEND=gloo runs the whole multi-process flow without RCCL
    # (collectives on CPU tensors, the all-to-all staged through host memory), FORMA_BENCH_ONE_DEVICE=1 puts every rank on
    # device 0 — together they let N ranks share the one GPU of a single-GPU box and exercise all the host logic of N > 1.
    backend = os.environ.get("FORMA_BENCH_BACKEND", "nccl")
    if os.environ.get("FORMA_BENCH_ONE_DEVICE"):
        local = 0
    cdev = "cuda" if backend == "nccl" else "cpu"                # where the small control tensors of the collectives live
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from forma_amd import api, scenes, sharding

    def measure(workload, primary=True, mode_req=None):
        if args.svg and primary:
            from forma_amd import svg as svg_loader
            width, height = 3840, 2160
            comp = svg_loader.Svg(args.svg, args.svg_scale).compose(api.Composition())
            workload = "svg:" + os.path.basename(args.svg) + f" x{args.svg_scale:g}"
        else:
            build_fn, width, height = scenes.WORKLOADS[workload]
            comp = build_fn()
        tiles_h = (height + 15) // 16
        clear = api.Color(1.0, 1.0, 1.0, 1.0)
        clr = (clear.r, clear.g, clear.b, clear.a)
        channels = api.RGBA

        def make_renderer():
            """a context with the scene resident: first frame through the public API (flattens on the GPU, uploads the tables)"""
            r = api.Renderer(device=local)
            img = np.zeros((height, width * 4), np.uint8)
            r.render(comp, api.BufferBuilder(img.reshape(-1), api.LinearLayout(width, width * 4, height)).build(), api.RGBA, clear, None, timings=True)
            return r, img

        renderer, image = make_renderer()
        ctx = renderer._ctx
        n_segments_full = renderer.last_timings["n_segments"]
        mode = (mode_req or args.mode) if (world > 1 or os.environ.get("FORMA_BENCH_MODE_AT_1")) else "single"   # (env: exercise a sharded mode on one GPU)
        in_flight = max(1, args.in_flight) if mode == "single" else 1     # (a sharded frame is one context per GPU)
        pool = [ctx] + [make_renderer()[0]._ctx for _ in range(in_flight - 1)]

        crop, row0, row1, xf = None, 0, tiles_h, None
        if mode == "bands":
            hist = sharding.row_histogram(ctx.segments(0), tiles_h)
            edges = sharding.agree_on_bands(dist, hist, world, device=cdev) if dist is not None else sharding.band_edges(hist, 1)
            row0, row1 = edges[rank], edges[rank + 1]
            ctx.set_band(row0, row1)
            crop = sharding.band_crop(edges, rank, width, height)
        elif mode == "exchange":
            tab = renderer.host_tables
            hist = sharding.row_histogram(ctx.segments(0), tiles_h)
            edges = sharding.agree_on_bands(dist, hist, world, device=cdev) if dist is not None else sharding.band_edges(hist, 1)
            row0, row1 = edges[rank], edges[rank + 1]
            cuts = sharding.line_shares(ctx.prepare_lines(width, height)["lengths"], world)
            ctx.set_geometry(*sharding.slice_geometry(tab["x"], tab["y"], tab["line_slot"], cuts[rank], cuts[rank + 1]))
            ctx.rasterize_frame(width, height)
            cap = sharding.pair_capacity(sharding.max_pair_count(dist, ctx.segments(0), edges, world, device=cdev))
            xf = sharding.ExchangeFrame(ctx, dist, rank, world, edges, width, height, cap)

        def frame(c=ctx, timings=False):
            if xf is not None:
                return xf.frame(channels=channels, clear=clr, timings=timings, device_only=True)
            return c.render(width, height, channels=channels, clear=clr, crop=crop, device_only=True, timings=timings)

        def sync_all():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        def run_block(steps, ctxs, acc=None):
            """exactly `steps` frames, len(ctxs) in flight: wall seconds (max over ranks).  With `acc` (a dict) every frame also
            records HIP events at its stage boundaries on its context's stream (forma_timings_t) and the per-frame values are
            summed into it: acc[key] = sum, acc["_frames"] = count."""
            share = [steps // len(ctxs) + (1 if i < steps % len(ctxs) else 0) for i in range(len(ctxs))]
            per = [dict() for _ in ctxs]

            def work(c, n, a):
                for _ in range(n):
                    if acc is None:
                        frame(c)
                    else:
                        _, t = frame(c, timings=True)
                        for k, v in t.items():
                            a[k] = a.get(k, 0.0) + float(v)
                        a["_frames"] = a.get("_frames", 0) + 1

            sync_all()
            t0 = time.perf_counter()
            if len(ctxs) == 1:
                work(ctxs[0], steps, per[0])
            else:
                ths = [threading.Thread(target=work, args=(c, n, a)) for c, n, a in zip(ctxs, share, per)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            sync_all()
            dt = time.perf_counter() - t0
            if acc is not None:
                for a in per:
                    for k, v in a.items():
                        acc[k] = acc.get(k, 0) + v
            return sharding.max_over_ranks(dist, dt, device=cdev) if dist is not None else dt

        for c in pool:
            for _ in range(max(1, args.warmup // len(pool))):
                frame(c)
        acc_one = {}
        elapsed = run_block(args.steps, pool)                       # THE timed region: exactly K frames, F in flight -> `value`
        frames_per_step = world if mode == "frames" else 1          # frames mode: one whole frame per GPU per step
        fps = frames_per_step * args.steps / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        # spread: five more blocks of K frames, pipelined; and five blocks of K frames with ONE frame in flight — the second
        # timed region of this run: a kernel's launch duration is its own only when nothing else shares the chip, so the
        # per-stage times and the roofline of the radix pass are taken there (HIP events on the context's stream)
        blocks = [frames_per_step * args.steps / run_block(args.steps, pool) for _ in range(5)]
        blocks1 = [frames_per_step * args.steps / run_block(args.steps, pool[:1]) for _ in range(5)] if in_flight > 1 else blocks
        run_block(args.steps, pool[:1], acc_one)                    # K more frames, one in flight, with the stage events

        def means(a):
            n = max(a.get("_frames", 0), 1)
            return {k: v / n for k, v in a.items() if k != "_frames"}

        stage = means(acc_one)
        n_local = int(round(stage["n_segments"]))                   # (exchange / bands: the segments this rank sorts)
        passes = int(round(stage["n_sort_passes"]))
        pass_us = stage["sort_pass_us"]
        algo_bytes_per_pass = 16.0 * n_local                        # 8 B read + 8 B written per key per digit pass
        achieved = (algo_bytes_per_pass / (pass_us * 1e-6) / 1e9) if pass_us > 0 else 0.0
        pmc = None
        try:
            with open(os.path.join(ROOT, PMC_FILE)) as f:
                pmc = json.load(f)
        except Exception:
            pmc = None
        use_pmc = pmc is not None and workload == "paris-like-30k-4k" and world == 1
        roofline = {"bound": "hbm", "kernel": "k_onesweep<8>: one radix digit pass (LSB, 8-bit digits over live key bits, u64 keys, chained scan)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": next((v.get("hbm_bytes_per_launch") for k, v in pmc["kernels"].items() if k.startswith("k_onesweep<8")), None) if use_pmc else None,
                    "traffic_source": (PMC_FILE + " — separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this command on "
                                       "the committed build (gfx950: FETCH_SIZE x 2), NOT measured in this run") if use_pmc else None,
                    "algorithmic_bytes_per_launch": algo_bytes_per_pass, "avg_launch_us": round(pass_us, 2), "passes": passes,
                    "measured": f"HIP events on the context's stream around every k_onesweep launch of {acc_one.get('_frames', 0)} frames with ONE "
                                "frame in flight (this run's second timed region; matches `rocprofv3 --kernel-trace --stats -- python "
                                "bench.py --in-flight 1`, profiles/r02_kernel_stats_inflight1.csv)"}
        if in_flight > 1:
            roofline["while_pipelined"] = ("with several frames in flight the kernels of different frames time-share the chip: rocprofv3 of the "
                                           "default command (profiles/r02_kernel_stats_default.csv) shows every kernel's average about "
                                           "1.5-2x its one-in-flight duration, while the frame rate rises; a launch duration is the kernel's "
                                           "own only with one frame in flight, which is where this roofline is measured")
        # the painter is not an HBM kernel: VALU issue and LDS bound it.  Live: its launch time; from the committed counters of
        # the same build: wave-level VALU instructions per launch and the LDS bank-conflict ratio.
        painter = {"kernel": "k_paint_wave (one wavefront per 16x16 tile)", "bound": "valu+lds", "avg_launch_us": round(stage["paint_us"], 1),
                   "hbm_algorithmic_bytes": 8.0 * n_local + 4.0 * width * height,
                   "hbm_achieved_GBs": round((8.0 * n_local + 4.0 * width * height) / max(stage["paint_us"], 1e-3) / 1e3, 1)}
        if use_pmc and "k_paint_wave" in pmc["kernels"]:
            k = pmc["kernels"]["k_paint_wave"]
            valu = k.get("SQ_INSTS_VALU")
            if valu:
                ach = valu / max(stage["paint_us"], 1e-3) / 1e3       # G wave-instructions / s
                painter.update({"valu_wave_instructions_per_launch": valu, "achieved": round(ach, 1), "peak": round(VALU_PEAK_GINST, 1),
                                "unit": "G wave64 VALU instructions/s", "frac": round(ach / VALU_PEAK_GINST, 4)})
            if k.get("SQ_LDS_IDX_ACTIVE"):
                painter["lds_bank_conflict_ratio"] = round(k.get("SQ_LDS_BANK_CONFLICT", 0) / k["SQ_LDS_IDX_ACTIVE"], 4)
            painter["counters_source"] = PMC_FILE + " (separate --pmc passes, NOT this run)"

        # PCIe-inclusive frames (image copied into caller memory) — reported, never `value`
        fps_d2h = None
        if mode in ("single", "frames", "bands"):
            imgs = [image] + [np.zeros_like(image) for _ in pool[1:]]

            def d2h_frames(c, img, n):
                for _ in range(n):
                    c.render(width, height, channels=channels, clear=clr, crop=crop, dst=img.reshape(-1), stride=width * 4)
            sync_all()
            t1 = time.perf_counter()
            ths = [threading.Thread(target=d2h_frames, args=(c, im, 6)) for c, im in zip(pool, imgs)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            fps_d2h = round(6 * len(pool) / (time.perf_counter() - t1), 2)

        sharding_txt = {
            "single": "none", "frames": f"frame-parallel x{world}: every GPU renders whole frames of the workload (units = frames), no exchange",
            "bands": f"tile-row bands x{world} of ONE frame, replicated scene, band culling, no data-path collective",
            "exchange": f"ONE frame: lines / {world} rasterized per GPU, HIP bucketing by tile-row owner, one RCCL all-to-all of pixel segments "
                        f"(padded equal split on the context's stream), band-local sort + paint"}[mode]
        out = {
            "metric": "frames/sec (sorted+painted, device-resident) + Mpixel-segments/sec",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if mode in ("single", "frames") else "strong", "vs_baseline": None,
            "dtype": "u64 segments / f64+f32 rasterizer / f32 painter", "data": "synthetic",
            **({"rehearsal_backend": backend} if backend != "nccl" else {}),
            "mpixel_segments_per_s": round(n_segments_full * fps / 1e6, 1),
            "frames_in_flight": in_flight,
            "fps_blocks": {"median": round(statistics.median(blocks), 1), "min": round(min(blocks), 1), "max": round(max(blocks), 1), "blocks": 5},
            "fps_one_frame_in_flight": {"median": round(statistics.median(blocks1), 1), "min": round(min(blocks1), 1), "max": round(max(blocks1), 1),
                                        "frame_latency_ms": round(1e3 / statistics.median(blocks1), 4)},
            "fps_including_d2h": fps_d2h,
            "config": {"workload": workload + (" (labelled stand-in: paris-30k.svg is not in the reference checkout)"
                                                     if workload.startswith("paris") else ""),
                       "canvas": [width, height], "layers": len(comp), "pixel_segments": int(n_segments_full),
                       "frames_in_flight": in_flight, "sharding": sharding_txt, "band_rows": [row0, row1]},
            "stages_us": {k: round(stage.get(k, 0.0), 1) for k in ("prepare_us", "rasterize_us", "exchange_us", "sort_us", "carry_us", "paint_us", "total_us")},
            "roofline": roofline,
            "roofline_painter": painter,
        }
        if primary and rank == 0 and world == 1 and args.animated:
            out["animated"] = animated_leg(local)
        if primary and rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(renderer, width, height, args.cpu_seconds)
        for c in pool:                                              # free the device buffers before the next workload
            try:
                c.close()
            except Exception:
                pass
        return out

    def agreed(ok):
        """every rank reports whether ITS attempt succeeded; the attempt counts only if all did"""
        if dist is None:
            return ok
        t = torch.tensor([1 if ok else 0], device=cdev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    # N > 1: the requested sharded mode first; if it fails on any rank (the RCCL exchange has only ever been exercised on
    # single-GPU boxes and with gloo), fall back to the modes without a data-path collective rather than lose the line
    modes = [args.mode] + [m for m in ("bands", "frames") if m != args.mode] if world > 1 else [None]
    out, errors = None, {}
    for m in modes:
        try:
            out = measure(args.workload, mode_req=m)
            ok = True
        except Exception as e:
            ok, errors[m or "single"] = False, repr(e)
            if world == 1:
                raise
        if agreed(ok):
            break
        out = None
    if out is None:
        raise SystemExit(f"every mode failed: {errors}")
    if errors:
        out["mode_fallback_errors"] = errors
    if world > 1 and not args.svg and args.workload != "triangles-10m-8k" and out["scaling"] == "strong":
        # N > 1: the same sharded mode on BASELINE config 4 (10 M pixel segments at 8192 x 8192), the configuration the multi-GPU
        # target is quoted on; its numbers ride in the same JSON line
        try:
            o2 = measure("triangles-10m-8k", primary=False, mode_req=m)
            ok = True
        except Exception as e:                                      # never lose the primary line to the extra leg
            ok, o2 = False, {"error": repr(e)}
        if agreed(ok):
            out["second_workload"] = {k: o2[k] for k in ("value", "unit", "ms_per_step", "scaling", "mpixel_segments_per_s", "fps_blocks",
                                                           "config", "stages_us", "roofline")}
        else:
            out["second_workload"] = {"error": o2.get("error", "failed on another rank")}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def animated_leg(local, frames=240):
    """BASELINE config 5 on one GPU: the deterministic spaceship (forma_amd/spaceship.py = the reference demo's game logic,
    fixed dt = 1/60 s) at 3840 x 2160, BGR1, clear (1, 1, 1, 0), through the product API exactly as the demo's runner does
    (demo/src/runner.rs:150-165): compose, then render into a caller buffer — with one persistent BufferLayerCache (damage
    tracking) and without a cache.  The game logic runs on the host between frames and is not timed."""
    import torch
    from forma_amd import api
    from forma_amd.spaceship import Spaceship
    W, H = 3840, 2160
    res = {}
    for label, cached in (("no_cache", False), ("with_cache", True)):
        comp, r = api.Composition(), api.Renderer(device=local)
        cache = r.create_buffer_layer_cache() if cached else None
        game = Spaceship(api, W, H)
        buf = np.zeros(W * H * 4, np.uint8)
        lay = api.LinearLayout(W, W * 4, H)
        spent, written, tiles = 0.0, [], ((W + 15) // 16) * ((H + 15) // 16)
        for f in range(frames):
            game.compose(comp)
            b = api.BufferBuilder(buf, lay)
            if cached:
                b = b.layer_cache(cache)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r.render(comp, b.build(), api.BGR1, api.Color(1, 1, 1, 0), None)
            spent += time.perf_counter() - t0
            if cached and f >= 10:
                written.append(int(r._ctx.tiles_written(W, H).sum()))
        res[label] = round(frames / spent, 1)
        if cached:
            res["damaged_tile_fraction"] = round(float(np.mean(written)) / tiles, 4)
            res["actors_at_end"] = len(game.actors)
    return {"workload": "spaceship (deterministic re-implementation of demo/src/demos/spaceship.rs, 3840x2160, BGR1)", "frames": frames,
            "fps_no_cache": res["no_cache"], "fps_with_cache": res["with_cache"], "damaged_tile_fraction": res["damaged_tile_fraction"],
            "actors_at_end": res["actors_at_end"],
            "unit": "frames/s of Renderer::render into a caller buffer (table upload + D2H of the written tiles included)"}


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
