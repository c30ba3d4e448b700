#!/usr/bin/env python3
"""bench.py — frames/s and Mpixel-segments/s of the forma raster hot path on MI355X.

A "step" is one `Renderer::render` frame (prepare lines -> rasterize -> radix sort -> carry pre-pass -> per-tile paint)
of a synthetic scene whose geometry, layer table and styles are already resident in HBM; the image stays device-resident
(the PCIe-inclusive rate is reported separately, never as `value`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME | --svg FILE [--svg-scale S]]
                    [--in-flight F] [--mode multi|exchange|bands|frames] [--no-animated] [--no-cpu-baseline]

N = 1.  ONE renderer context — what a caller of `Renderer::render` has — renders exactly K frames; `value` = K / wall time.
The context keeps `--in-flight` F frame slots (default 3, `forma_hip_set_frames_in_flight`): a device-resident frame is
enqueued on the next slot and verified when the slot comes round, so the kernels of consecutive frames overlap on the GPU
(they are bound by their own dependent round trips, not by a chip-wide resource).  Next to it, always:
  * `fps_render_call`: the same context with ONE frame in flight, every call complete when it returns — SURVEY §8(d)'s
    "1 / wall time of one render call" — with its latency, and `fps_including_d2h`: >= 60 such calls that also copy the
    33 MB image into caller memory (what `Renderer::render` promises; never `value`);
  * `stages_us` and `roofline` (the radix digit pass, HBM-bound): a further region of K frames with one frame in flight and
    HIP events at the stage boundaries on the context's stream — with frames overlapping, a launch duration is not the
    kernel's own;
  * `animated`: BASELINE config 5 (deterministic spaceship, 600 frames at 4K, with and without the buffer-layer cache);
  * `cpu_baseline`: the C++ oracle on the same scene tables (`kind: "port"`).
`roofline.traffic`, `post_sort_traffic_ratio` and the painter's counters are measured in the run at N = 1 (three child runs of this command under
`rocprofv3 --pmc`: the SQ counters, FETCH_SIZE, WRITE_SIZE; tools/pmc_round.py, ~15 s; `--no-pmc` skips them); without rocprofv3 it is read from the committed counter summary and labelled so.

N > 1, launched as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (the contract): default `--mode
multi` — rank 0 holds ONE renderer over all N GPUs (`forma_hip_create_multi`: per-device host threads inside the library,
line-sharded rasterization, HIP bucketing by tile-row owner, ONE RCCL all-to-all of pixel segments over xGMI, band-local
sort + paint, every device copying its rows into the one caller buffer) and the other ranks only take part in the barriers
and the max-over-ranks timing; strong scaling, `value` = frames of the ONE scene per second.  A short preflight in a child
process (killed on timeout) guards the first multi-GPU execution of that path; on failure the run falls back to `--mode
exchange` (one process per GPU, the same layout driven through torch.distributed), then `bands`, then `frames`, and says so.

Rehearsal switches for single-GPU boxes (never set by the driver): FORMA_BENCH_MODE_AT_1=1 runs a sharded mode with one rank
(`multi`: FORMA_BENCH_DEVICES=0,0,0,0 lists the "devices"); FORMA_BENCH_BACKEND=gloo + FORMA_BENCH_ONE_DEVICE=1 run N ranks
on one GPU without RCCL.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import statistics
import subprocess
import sys
import time

os.environ.setdefault("OMP_PROC_BIND", "close")      # the CPU baseline's OpenMP loops: threads stay where their pages are
os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STAGE_KEYS = ("prepare_us", "rasterize_us", "exchange_us", "sort_us", "carry_us", "paint_us")
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2   # wave64 VALU instructions/ns the chip can issue: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles
# ... and what it does issue (tools/ubench_issue.hip, profiles/r06_valu_issue_peak.txt: one workgroup per CU, four waves per SIMD of 16
# independent chains each): ~1 000 G/s of the FULL-rate instructions (mov, add / sub, and / or / xor, lshr, mul / add / fma f32, bitop3) and
# ~570 G/s of everything else (HALF rate: conversions, floor, compares, cndmask, min / max, lshl, bfe / bfi, and_or / or3 / lshl_add, mul_lo,
# mbcnt, DPP moves, every 64-bit and f64 instruction, packed f32); rcp at a quarter.  A kernel's instructions therefore cost
# `slots_per_instruction` full-rate issue slots on average (tools/isa_mix.py over its ISA: 1.63 for k_paint_wave, 1.60 for k_rasterize).
VALU_MEASURED_GINST = 1000.0
PAINT_SLOTS_PER_INST = 1.63
PMC_FILES = [os.path.join("profiles", f"r0{r}_pmc_summary.json") for r in (6, 5, 4, 3)]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--workload", default="paris-like-30k-4k")
    ap.add_argument("--in-flight", type=int, default=3, help="N = 1: frame slots inside the ONE renderer context (frames in flight)")
    ap.add_argument("--mode", default="multi", choices=["multi", "exchange", "bands", "frames"],
                    help="N > 1: 'multi' = ONE renderer over all GPUs (forma_hip_create_multi, RCCL inside the library), driven "
                         "by rank 0; 'exchange' = the same layout with one process per GPU (torch.distributed all-to-all); "
                         "'bands' = replicated scene, band culling, no exchange; 'frames' = whole frames per GPU (weak scaling)")
    ap.add_argument("--svg", default=None, metavar="FILE", help="render this SVG (e.g. the real paris-30k.svg) instead of a synthetic workload")
    ap.add_argument("--svg-scale", type=float, default=1.0)
    ap.add_argument("--no-animated", action="store_true", help="skip BASELINE config 5 (deterministic spaceship, 600 frames at 4K)")
    ap.add_argument("--animated-frames", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic and the painter's counters with rocprofv3 --pmc child runs (three short runs of this "
                                                          "command, ~15 s); the figure of the committed counter summary is reported instead")
    ap.add_argument("--no-d2h", action="store_true", help="skip the PCIe-inclusive legs (profiling runs: their pipelined frames would be "
                    "averaged into the per-kernel durations of the one-frame-in-flight region)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--preflight", default=None, help=argparse.SUPPRESS)     # child process: try a multi-device context, print OK
    return ap.parse_args()


def preflight(devices):
    """child process of `--mode multi`: ONE context over `devices`, a small scene, a few frames, compared across frames.
    Runs where a hang can be killed without losing the bench line."""
    import torch  # noqa: F401  (first: one HIP runtime in the process)
    from forma_amd import api, scenes
    comp = scenes.random_cubics(64, 512, 512)
    r = api.Renderer(devices=devices)
    buf = np.zeros(512 * 512 * 4, np.uint8)
    lay = api.LinearLayout(512, 2048, 512)
    first = None
    # both layouts of the multi-device context (forma_hip_multi_layout): BANDS needs no collective; EXCHANGE is the first
    # execution of the RCCL all-to-all over more than one device that this project has ever seen — each announces itself
    for layout in ("bands", "exchange"):
        r._ctx.set_layout(layout)
        for _ in range(4):
            r.render(comp, api.BufferBuilder(buf, lay).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
            first = buf.copy() if first is None else first
            assert np.array_equal(first, buf), layout
        assert (buf != 255).any()
        print("PREFLIGHT-OK " + layout, flush=True)


def main():
    args = parse()
    if args.preflight is not None:
        preflight([int(v) for v in args.preflight.split(",")])
        return
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # `python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE): ONE process drives all N GPUs — forma_hip_create_multi needs
    # no second process (per-device host threads and the RCCL communicators live inside the library).  The line then says
    # n_gpus = N.  Fewer than N visible devices is an error (rc 2), never a silent one-GPU measurement.
    in_process = "WORLD_SIZE" not in os.environ and args.gpus > 1
    if not in_process and world != args.gpus:
        if rank == 0:
            print(f"WORLD_SIZE={world} does not match --gpus {args.gpus}: measuring {world} GPU(s), n_gpus says so", file=sys.stderr)
        args.gpus = world
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
    sharded = world > 1 or in_process or bool(os.environ.get("FORMA_BENCH_MODE_AT_1"))
    n_gpus = args.gpus if in_process else world
    multi_devices = [int(v) for v in os.environ["FORMA_BENCH_DEVICES"].split(",")] if os.environ.get("FORMA_BENCH_DEVICES") else list(range(n_gpus))
    if in_process:
        visible = torch.cuda.device_count()
        if len(multi_devices) != args.gpus or min(multi_devices) < 0 or max(multi_devices) >= visible:
            print(f"bench.py --gpus {args.gpus}: needs devices {multi_devices}, {visible} visible (FORMA_BENCH_DEVICES=0,0,.. rehearses "
                  f"on fewer) — refusing to measure fewer GPUs than asked for", file=sys.stderr)
            raise SystemExit(2)
        if args.mode != "multi":
            print(f"bench.py --gpus {args.gpus} without a launcher runs --mode multi (the other modes are one process per GPU: "
                  f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} --mode {args.mode})", file=sys.stderr)
            args.mode = "multi"

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        """wall seconds of fn(), bracketed by barrier + synchronize on both sides, maximum over ranks.  The Python collector
        stays out of the timed region: the host loop allocates a few small objects per render call, and a generation-2 pass
        over a 30 000-layer composition is tens of milliseconds — one such pass inside a 23 ms region halved `value` once
        (profiles/r04: 1 362 fps next to blocks of 2 570)."""
        sync_all()
        gc_was = gc.isenabled()
        gc.disable()
        t0 = time.perf_counter()
        fn()
        sync_all()
        dt = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        return sharding.max_over_ranks(dist, dt, device=cdev) if dist is not None else dt

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
        mode = (mode_req or args.mode) if sharded else "single"
        driver = mode != "multi" or rank == 0                    # multi: rank 0 drives every GPU, the others only synchronise
        image = np.zeros((height, width * 4), np.uint8)

        renderer = None
        if driver:
            renderer = api.Renderer(devices=multi_devices) if mode == "multi" else api.Renderer(device=local)
            # first frame through the public API: flattens on the GPU, uploads the tables, plans (multi)
            renderer.render(comp, api.BufferBuilder(image.reshape(-1), api.LinearLayout(width, width * 4, height)).build(), api.RGBA, clear, None, timings=True)
        ctx = renderer._ctx if renderer else None
        n_segments_full = renderer.last_timings["n_segments"] if renderer and mode != "multi" else 0
        if mode == "multi":
            n_full = torch.tensor([len(ctx.segments(1)) if driver else 0], dtype=torch.int64, device=cdev)
            if dist is not None:
                dist.broadcast(n_full, 0)
            n_segments_full = int(n_full.item())
        # frame slots: inside the ONE context (single GPU), and — round 4 — on every device of a multi-device context
        in_flight = max(1, min(4, args.in_flight)) if mode in ("single", "multi") else 1

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

        def frame(timings=False, dst=None):
            if not driver:
                return None
            if xf is not None:
                return xf.frame(channels=channels, clear=clr, timings=timings, device_only=dst is None, dst=dst, stride=width * 4 if dst is not None else None)
            if dst is not None:
                return ctx.render(width, height, channels=channels, clear=clr, crop=crop, dst=dst, stride=width * 4, timings=timings)
            return ctx.render(width, height, channels=channels, clear=clr, crop=crop, device_only=True, timings=timings)

        def frames(n):
            """n frames; with frames in flight the calls return before their frames are done, so the block ends with a sync"""
            for _ in range(n):
                frame()
            if driver and mode in ("single", "frames", "multi"):
                ctx.sync()

        def rate(steps, per_step=1):
            return per_step * steps / timed(lambda: frames(steps))

        frames_per_step = world if mode == "frames" else 1          # frames mode: one whole frame per GPU per step
        gc.collect()
        gc.freeze()                                                 # (the scene's objects leave the collector's generations — before
        #                                                             the GPU warms up: a collection here idles it for ~0.1 s)
        if driver and in_flight > 1:
            ctx.set_frames_in_flight(in_flight)
            # SET-UP, not warm-up: every frame slot runs its first frame synchronously (it learns N, the key masks and J), its
            # second one read-back-free for the first time (buffers grow to their bounds).  Until round 3 the W warm-up frames
            # did this, and with W = 5 over three slots the timed region still held first-time work: `value` sat 17 % under the
            # median of the blocks that followed.
            frames(3 * in_flight + 3)
            # ... and the clocks: the GPU sat idle while Python built the scene; a quarter of a second of frames before the
            # warm-up lets the power management settle (untimed, like everything above)
            if mode in ("single", "multi"):                         # (one process renders: a time-based loop is safe)
                t_ramp = time.perf_counter()
                while time.perf_counter() - t_ramp < 0.25:
                    frames(8)
            else:
                frames(64)                                          # (every rank the same count: frames may hold collectives)
        frames(max(args.warmup, 1))                                 # W warm-up frames
        elapsed = timed(lambda: frames(args.steps))                 # THE timed region: exactly K frames -> `value`
        fps = frames_per_step * args.steps / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        blocks = [rate(args.steps, frames_per_step) for _ in range(5)]
        # the multi-device context's two layouts (forma_hip_multi_layout): `value` is the default's (AUTO), both are listed
        layouts = None
        if mode == "multi" and driver:
            chosen = ctx.info().get("layout")
            layouts = {chosen: round(statistics.median(blocks), 1)}
            other = "exchange" if chosen == "bands" else "bands"
            if other == "exchange" and os.environ.get("FORMA_BENCH_NO_EXCHANGE"):
                layouts[other] = None
            else:
                try:
                    ctx.set_layout(other)
                    frames(3 * in_flight + 3)
                    layouts[other] = round(statistics.median(rate(args.steps, frames_per_step) for _ in range(3)), 1)
                except Exception as e:                                  # noqa: BLE001  (never lose the line to the second layout)
                    layouts[other] = None
                    errors_layout = repr(e)
                    layouts["error"] = errors_layout
                try:
                    ctx.set_layout("auto")
                    frames(3 * in_flight + 3)
                except Exception:                                       # noqa: BLE001
                    pass
        # one frame in flight: every render call is complete when it returns (1 / wall time of one render call)
        if driver and in_flight > 1:
            ctx.set_frames_in_flight(1)
            frames(2)
        blocks1 = [rate(args.steps, frames_per_step) for _ in range(5)] if in_flight > 1 else blocks
        # K more frames, one in flight, HIP events at the stage boundaries on the context's stream(s)
        acc, kacc = {}, {}

        def staged():
            for _ in range(max(args.steps, 60)):                       # (>= 60 frames: a 20-frame mean of 60 us kernels wobbles by 2 %)
                r = frame(timings=True)
                if r is None:
                    continue
                for k, v in r[1].items():
                    acc[k] = acc.get(k, 0.0) + float(v)
                acc["_frames"] = acc.get("_frames", 0) + 1
                if mode == "single":                                   # every kernel's own launch events (forma_hip_kernel_times)
                    for name, _st, _t0, us in ctx.kernel_times():
                        e = kacc.setdefault(name, [0.0, 0])
                        e[0] += us
                        e[1] += 1
        timed(staged)
        nfr = max(acc.get("_frames", 0), 1)
        stage = {k: v / nfr for k, v in acc.items() if k != "_frames"}
        # per kernel: microseconds per FRAME (all its launches of a frame together) and launches per frame
        kernels_us = {k: {"us_per_frame": round(v[0] / nfr, 2), "launches_per_frame": round(v[1] / nfr, 2)} for k, v in kacc.items()}
        if dist is not None and mode == "multi":                      # rank 0 measured; everybody reports the same line
            keys = ["prepare_us", "rasterize_us", "exchange_us", "sort_us", "sort_pass_us", "carry_us", "paint_us", "total_us", "n_segments", "n_sort_passes"]
            tt = torch.tensor([stage.get(k, 0.0) for k in keys], dtype=torch.float64, device=cdev)
            dist.broadcast(tt, 0)
            stage = {k: float(v) for k, v in zip(keys, tt.tolist())}
        n_local = int(round(stage.get("n_segments", 0)))            # segments one device sorts (multi: summed over the devices)
        if mode == "multi":
            n_local = max(1, n_local // max(len(multi_devices), 1))
        passes = int(round(stage.get("n_sort_passes", 0)))
        pass_us = stage.get("sort_pass_us", 0.0)
        algo_bytes_per_pass = 16.0 * n_local                        # 8 B read + 8 B written per key per digit pass
        achieved = (algo_bytes_per_pass / (pass_us * 1e-6) / 1e9) if pass_us > 0 else 0.0
        pmc, pmc_file = None, None
        for f in PMC_FILES:
            try:
                with open(os.path.join(ROOT, f)) as fh:
                    pmc, pmc_file = json.load(fh), f
                break
            except Exception:
                continue
        use_pmc = pmc is not None and workload == "paris-like-30k-4k" and mode == "single"
        roofline = {"bound": "hbm", "kernel": "k_onesweep: one radix digit pass (LSB; 8-bit digits over live key bits, 9-bit where that saves a pass; "
                                              "u64 keys, chained scan)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": next((v.get("hbm_bytes_per_launch") for k, v in pmc["kernels"].items() if k.startswith("k_onesweep")), None) if use_pmc else None,
                    "traffic_source": (pmc_file + " — separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this command on "
                                       "the committed build (gfx950: FETCH_SIZE x 2), NOT measured in this run") if use_pmc else None,
                    "algorithmic_bytes_per_launch": algo_bytes_per_pass, "avg_launch_us": round(pass_us, 2), "passes": passes,
                    "measured": f"HIP events carried by every k_onesweep launch (hipExtLaunchKernelGGL: the dispatch's own start and end) of "
                                f"{acc.get('_frames', 0)} frames with ONE frame in flight, a further timed region of this run; matches `rocprofv3 "
                                "--kernel-trace --stats -- python bench.py --in-flight 1 --no-d2h --no-animated --no-cpu-baseline`, profiles/r06g_kernel_stats_inflight1.csv"}
        # the whole sort against the same roofline: histogram read + p digit passes = 8 N (2 p + 1) bytes — or 16 N p
        # when the histograms come out of the rasterizer's registers (read-back-free frames with <= 3 passes: k_sort_hist and its
        # read of the stream do not run; the counting costs k_rasterize ~13 us, which stays in the rasterize stage)
        sort_us = stage.get("sort_us", 0.0)
        dbg = os.environ.get("FORMA_HIP_DEBUG", "")
        hist_fused = mode in ("single", "bands", "frames") and 0 < passes <= 3 and not any(t in dbg for t in ("no_ras_hist", "no_prezero", "sync"))
        if passes and pass_us > 0:
            # the digit passes alone, no gaps: 16 N p bytes over the sum of the pass kernels' own durations
            roofline["whole_sort_kernels"] = {"algorithmic_bytes": 16.0 * n_local * passes, "us": round(pass_us * passes, 1),
                                              "frac": round(16.0 * n_local * passes / (pass_us * passes * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        if sort_us > 0 and passes:
            sort_bytes = 8.0 * n_local * (2 * passes + (0 if hist_fused else 1))
            whole = sort_bytes / (sort_us * 1e-6) / 1e9
            roofline["whole_sort"] = {"algorithmic_bytes": sort_bytes, "us": round(sort_us, 1), "achieved": round(whole, 1),
                                      "frac": round(whole / HBM_PEAK_GBS, 4), "histograms": "k_rasterize" if hist_fused else "k_sort_hist",
                                      "what": ("every digit pass (the digit histograms are counted by k_rasterize "
                                               "while it makes the keys)" if hist_fused else
                                               "k_sort_hist + every digit pass") + ": the sum of the sort kernels' own launch-event durations of the same frames"}
        # what the stages behind the sort move, against what they have to (8 N in + 4 W H out): counters of the committed build
        if use_pmc:
            # (per kernel name the instantiation that runs on EVERY frame — `dispatches` — and only kernels of this run's frames:
            #  the synchronous first frame of a scene launches k_runs_count and k_runs_wave<0> once each)
            post = []
            for base in ("k_runs_count", "k_runs_wave", "k_carry_rows", "k_paint_wave"):
                if base not in kernels_us:
                    continue
                cands = [v for k, v in pmc["kernels"].items() if k.split("<")[0] == base]
                if cands:
                    post.append(max(cands, key=lambda v: v.get("dispatches", 0)).get("hbm_bytes_per_launch", 0))
            if post and all(post):
                roofline["post_sort_traffic_ratio"] = round(sum(post) / (8.0 * n_local + 4.0 * width * height), 3)
        if in_flight > 1:
            roofline["while_pipelined"] = ("with several frames in flight the kernels of different frames time-share the chip (every kernel's "
                                           "average rises to about 1.5-2x its one-in-flight duration while the frame rate rises): a launch "
                                           "duration is the kernel's own only with one frame in flight, which is where this roofline is measured")
        # the painter is not an HBM kernel: VALU issue and LDS bound it.  Live: its launch time; from the committed counters of
        # the same build: wave-level VALU instructions per launch and the LDS bank-conflict ratio.
        pk_name = next((k for k in ("k_paint_wave", "k_paint_quad") if k in kernels_us), None)
        paint_k_us = kernels_us[pk_name]["us_per_frame"] if pk_name else stage.get("paint_us", 0.0)   # (the kernel's own launch events)
        painter = {"kernel": {"k_paint_quad": "k_paint_quad (one wavefront per four 16x16 tiles: all-solid scenes with many shallow tiles)"}.get(
                       pk_name, "k_paint_wave (one wavefront per 16x16 tile; frames below the chip's wave slots: four strip wavefronts per tile)"),
                   "bound": "valu+lds", "avg_launch_us": round(paint_k_us, 1),
                   "hbm_algorithmic_bytes": 8.0 * n_local + 4.0 * width * height,
                   "hbm_achieved_GBs": round((8.0 * n_local + 4.0 * width * height) / max(paint_k_us, 1e-3) / 1e3, 1)}
        pk = next((v for n_, v in pmc["kernels"].items() if n_.startswith(pk_name or "k_paint_wave")), None) if use_pmc else None
        if pk is not None:                                            # (the counters' key carries template arguments: match by prefix)
            k = pk
            valu = k.get("SQ_INSTS_VALU")
            if valu:
                ach = valu / max(paint_k_us, 1e-3) / 1e3              # G wave-instructions / s
                painter.update({"valu_wave_instructions_per_launch": valu, "achieved": round(ach, 1), "peak": round(VALU_PEAK_GINST, 1),
                                "unit": "G wave64 VALU instructions/s", "frac": round(ach / VALU_PEAK_GINST, 4),
                                "peak_measured": VALU_MEASURED_GINST, "slots_per_instruction": PAINT_SLOTS_PER_INST,
                                "frac_of_measured": round(ach * PAINT_SLOTS_PER_INST / VALU_MEASURED_GINST, 4),
                                "peak_measured_what": "tools/ubench_issue.hip: G full-rate issue slots/s with four waves per SIMD on every CU (mov / add / logic / "
                                                      "f32 mul-add issue at this rate, conversions / compares / selects / min-max / 64-bit / f64 / packed at half); "
                                                      "frac_of_measured = achieved x slots_per_instruction (static mix of the kernel's ISA, tools/isa_mix.py) / peak_measured"})
            if k.get("SQ_LDS_IDX_ACTIVE"):
                painter["lds_bank_conflict_ratio"] = round(k.get("SQ_LDS_BANK_CONFLICT", 0) / k["SQ_LDS_IDX_ACTIVE"], 4)
            painter["counters_source"] = pmc_file + " (separate --pmc passes, NOT this run)"

        # PCIe-inclusive frames: every call copies its image (its band) into caller memory and is complete when it returns
        n_d2h = max(60, args.steps)

        def d2h_frames(n=None):
            for _ in range(n or n_d2h):
                frame(dst=image)
        # (eight untimed frames first: the first frame into caller memory allocates the pinned staging image and a frame into caller memory
        #  learns where to split its painter launch; then the median of three blocks — PCIe rates wobble from block to block, as in the
        #  enqueued leg below: 909 and 972 frames/s came out of two runs of the same build on the same box with one block each)
        fps_d2h, fps_d2h_blocks = None, None
        if not args.no_d2h:
            d2h_frames(8)
            fps_d2h_blocks = [round(frames_per_step * n_d2h / timed(d2h_frames), 2) for _ in range(3)]
            fps_d2h = round(statistics.median(fps_d2h_blocks), 2)
        # frame AND copy enqueued (forma_hip_render_enqueue): three registered caller buffers in turn, two frame slots, ONE context
        # and ONE host thread — the 33 MB copy of frame k crosses PCIe under the kernels of frames k + 1, k + 2; a buffer is
        # complete two enqueues later (what a presenter that rotates window buffers does)
        fps_d2h_enqueue = None
        if mode == "single" and primary and not args.no_d2h:
            bufs = [np.zeros_like(image) for _ in range(3)]
            for b in bufs:
                ctx.register_buffer(b)
            ctx.set_frames_in_flight(2)
            kq = [0]

            def enq(n):
                for _ in range(n):
                    ctx.render_enqueue(width, height, bufs[kq[0] % 3], channels=channels, clear=clr, crop=crop)
                    kq[0] += 1
                ctx.sync()
            enq(8)
            fps_d2h_enqueue = round(statistics.median(n_d2h / timed(lambda: enq(n_d2h)) for _ in range(3)), 2)    # (3 blocks: PCIe rates wobble)
            ctx.set_frames_in_flight(1)
            for b in bufs:
                ctx.unregister_buffer(b)
            del bufs
        # the frame-server case: three INDEPENDENT renderer contexts (three host threads) each delivering complete frames into
        # its own caller buffer — the 33 MB PCIe copy of one context's frame overlaps the kernels of the others
        fps_d2h_server = None
        if mode == "single" and primary and not args.no_d2h:
            import threading
            extra = []
            for _ in range(2):
                r2 = api.Renderer(device=local)
                im2 = np.zeros_like(image)
                r2.render(comp, api.BufferBuilder(im2.reshape(-1), api.LinearLayout(width, width * 4, height)).build(), api.RGBA, clear, None)
                extra.append((r2._ctx, im2))
            pool = [(ctx, image)] + extra

            def server():
                def work(c, im):
                    for _ in range(n_d2h // 3):
                        c.render(width, height, channels=channels, clear=clr, dst=im, stride=width * 4)
                ths = [threading.Thread(target=work, args=p) for p in pool]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            for c, im in extra:
                c.render(width, height, channels=channels, clear=clr, dst=im, stride=width * 4)
            fps_d2h_server = round(3 * (n_d2h // 3) / timed(server), 2)
            for c, _ in extra:
                c.close()

        sharding_txt = {
            "single": "none",
            "multi": f"ONE renderer context over {len(multi_devices)} GPUs (forma_hip_create_multi, driven by rank 0; per-device host threads inside "
                     f"libforma_hip.so), tile-row bands of equal pixel-segment counts, every device writes its rows of the one caller buffer; layout "
                     f"AUTO (forma_hip_multi_layout): BANDS = every device culls the whole scene to its band, no exchange; EXCHANGE = lines / "
                     f"{len(multi_devices)} rasterized per GPU, HIP bucketing by tile-row owner, one RCCL all-to-all of pixel segments",
            "frames": f"frame-parallel x{world}: every GPU renders whole frames of the workload (units = frames), no exchange",
            "bands": f"tile-row bands x{world} of ONE frame, replicated scene, band culling, no data-path collective",
            "exchange": f"ONE frame, one process per GPU: lines / {world} rasterized per GPU, HIP bucketing by tile-row owner, one RCCL all-to-all of "
                        f"pixel segments (torch.distributed, padded equal split on the context's stream), band-local sort + paint"}[mode]
        lat = 1e3 / statistics.median(blocks1)
        out = {
            "metric": "frames/sec (sorted+painted, device-resident) + Mpixel-segments/sec",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if mode in ("single", "frames") else "strong", "vs_baseline": None,
            "dtype": "u64 segments / f64+f32 rasterizer / f32 painter", "data": "synthetic",
            **({"rehearsal_backend": backend} if backend != "nccl" else {}),
            **({"rehearsal_devices": multi_devices, "rehearsal": "ONE physical GPU listed several times: n_gpus counts device CONTEXTS, not GPUs"}
               if mode == "multi" and len(set(multi_devices)) < len(multi_devices) else {}),
            **({"launched": "in-process: one host process drives every GPU (no torchrun)"} if in_process else {}),
            "value_is": (f"ONE renderer context, {in_flight} frames in flight inside it (forma_hip_set_frames_in_flight), one host thread"
                         if mode == "single" else sharding_txt),
            "mpixel_segments_per_s": round(n_segments_full * fps / 1e6, 1),
            "frames_in_flight": in_flight,
            "fps_blocks": {"median": round(statistics.median(blocks), 1), "min": round(min(blocks), 1), "max": round(max(blocks), 1), "blocks": 5},
            "fps_render_call": {"median": round(statistics.median(blocks1), 1), "min": round(min(blocks1), 1), "max": round(max(blocks1), 1),
                                "frame_latency_ms": round(lat, 4),
                                "what": "the same context with ONE frame in flight: every render call is complete when it returns "
                                        "(SURVEY §8d: 1 / wall time of one render call, device-resident output)"},
            "fps_including_d2h": fps_d2h,
            "fps_including_d2h_blocks": fps_d2h_blocks,
            "fps_including_d2h_frames": n_d2h,
            "fps_including_d2h_enqueued_three_buffers": fps_d2h_enqueue,
            "fps_including_d2h_three_contexts": fps_d2h_server,
            "config": {"workload": workload + (" (labelled stand-in: paris-30k.svg is not in the reference checkout)"
                                                     if workload.startswith("paris") else ""),
                       "canvas": [width, height], "layers": len(comp), "pixel_segments": int(n_segments_full),
                       "frames_in_flight": in_flight, "sharding": sharding_txt, "band_rows": [row0, row1]},
            "stages_us": {**{k: round(stage.get(k, 0.0), 1) for k in STAGE_KEYS}, "total_us": round(sum(stage.get(k, 0.0) for k in STAGE_KEYS), 1)},
            "stages_us_what": "a stage = the sum of its kernels' own durations (events carried by each launch: no marker overhead), total_us = "
                              "the sum of the stages = GPU time of a frame's kernels; compare with fps_render_call.frame_latency_ms (wall time of "
                              "an untimed call: kernels + the gaps between ten dependent launches + the host's share)",
            # first kernel start -> last kernel end of a TIMED frame: launches that carry events sit ~3 us further apart than plain ones
            "timed_frame_span_us": round(stage.get("total_us", 0.0), 1),
            "kernels_us": kernels_us,
            "roofline": roofline,
            "roofline_painter": painter,
        }
        out["fps_one_frame_in_flight"] = out["fps_render_call"]       # (the name earlier rounds used)
        if layouts is not None:
            out["multi_layouts_fps"] = layouts
            out["multi_layouts_what"] = ("frames/s of the same multi-device context in both layouts (forma_hip_multi_layout): 'bands' = no exchange, every "
                                         "device culls the scene to its band of tile rows; 'exchange' = line shares + one all-to-all of pixel segments; "
                                         "`value` is the default layout's (AUTO picks per scene)")
        if driver and ctx is not None:
            try:
                out["context"] = ctx.info()                           # devices, frame slots, exchange transport (rccl / copy)
            except Exception:                                         # noqa: BLE001
                pass
        if primary and rank == 0 and world == 1 and not args.no_animated and mode == "single":
            out["animated"] = animated_leg(local, args.animated_frames)
        if primary and rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(renderer, width, height, args.cpu_seconds)
        if ctx is not None:
            try:
                ctx.close()                                         # free the device buffers before the next workload
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

    def multi_preflight_ok():
        """rank 0 tries the multi-device context in a child process that can be killed; everybody learns the verdict"""
        ok, why = True, ""
        if rank == 0:
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--preflight", ",".join(str(d) for d in multi_devices)],
                                   capture_output=True, text=True, timeout=240)
                ok = "PREFLIGHT-OK bands" in p.stdout                 # (BANDS is enough to go on; EXCHANGE is measured only if it passed too)
                if "PREFLIGHT-OK exchange" not in p.stdout:
                    os.environ["FORMA_BENCH_NO_EXCHANGE"] = "1"
                    errors["multi-exchange"] = "preflight: " + (p.stderr or p.stdout)[-400:]
                why = (p.stderr or p.stdout)[-400:] if not ok else ""
            except subprocess.TimeoutExpired as e:
                out_so_far = (e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or ""))
                ok = "PREFLIGHT-OK bands" in out_so_far
                os.environ["FORMA_BENCH_NO_EXCHANGE"] = "1"
                errors["multi-exchange"] = "preflight timed out in the exchange layout (killed)"
                why = "" if ok else "preflight timed out (killed)"
            except Exception as e:                                    # noqa: BLE001
                ok, why = False, repr(e)
        return agreed(ok), why

    # N > 1: the requested sharded mode first; if it fails on any rank, fall back to the next one rather than lose the line
    chain = ["multi", "exchange", "bands", "frames"]
    modes = [args.mode] + [m for m in chain[chain.index(args.mode) + 1:] if m != args.mode] if sharded else [None]
    if in_process:
        modes = ["multi"]                                             # (the other modes need one process per GPU: relaunch() below)
    out, errors, m = None, {}, None
    for m in modes:
        if m == "multi" and (world > 1 or len(set(multi_devices)) > 1):    # (the first execution on distinct devices: in a child that can be killed)
            ok, why = multi_preflight_ok()
            if not ok:
                errors["multi"] = "preflight: " + why
                continue
        try:
            out = measure(args.workload, mode_req=m)
            ok = True
        except Exception as e:                                        # noqa: BLE001
            ok, errors[m or "single"] = False, repr(e)
            if not sharded:
                raise
        if agreed(ok):
            break
        out = None
    if out is None and in_process and len(set(multi_devices)) == len(multi_devices):
        # ONE process could not drive the N GPUs: the same layout with one process per GPU (the launcher the contract names), its
        # line handed through with the reason attached
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup",
               str(args.warmup), "--workload", args.workload, "--mode", "exchange"]
        p = subprocess.run(cmd, capture_output=True, text=True)
        line = next((ln for ln in reversed(p.stdout.splitlines()) if ln.startswith("{")), None)
        if p.returncode != 0 or line is None:
            raise SystemExit(f"every mode failed: {errors}; relaunch under torch.distributed.run: rc {p.returncode}: {(p.stderr or p.stdout)[-400:]}")
        o = json.loads(line)
        o.setdefault("mode_fallback_errors", {}).update(errors)
        print(json.dumps(o), flush=True)
        return
    if out is None:
        raise SystemExit(f"every mode failed: {errors}")
    if errors:
        out["mode_fallback_errors"] = errors
    if rank == 0 and world == 1 and args.gpus == 1 and not args.no_pmc and not args.svg and args.workload == "paris-like-30k-4k" \
            and isinstance(out.get("roofline"), dict):
        live = live_traffic(out)
        if live is not None:
            out["roofline"].update(live["roofline"])
            if isinstance(out.get("roofline_painter"), dict):
                out["roofline_painter"].update(live["roofline_painter"])
    if sharded and not args.svg and args.workload != "triangles-10m-8k" and out["scaling"] == "strong":
        # N > 1: the same sharded mode on BASELINE config 4 (10 M pixel segments at 8192 x 8192), the configuration the multi-GPU
        # target is quoted on; its numbers ride in the same JSON line
        try:
            o2 = measure("triangles-10m-8k", primary=False, mode_req=m)
            ok = True
        except Exception as e:                                      # never lose the primary line to the extra leg
            ok, o2 = False, {"error": repr(e)}
        if agreed(ok):
            out["second_workload"] = {k: o2[k] for k in ("value", "unit", "ms_per_step", "scaling", "mpixel_segments_per_s", "fps_blocks",
                                                           "fps_including_d2h", "config", "stages_us", "roofline")}
        else:
            out["second_workload"] = {"error": o2.get("error", "failed on another rank")}
    if rank == 0:
        try:                                                          # RCCL prints a version banner through C stdio when a communicator is
            import ctypes                                             # created: flush it BEFORE the line, so that the JSON is the last line
            ctypes.CDLL(None).fflush(None)
        except Exception:                                             # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def live_traffic(out):
    """The counter-derived figures of the line measured in THIS run: three child runs of this command under `rocprofv3 --pmc` (the SQ
    counters; FETCH_SIZE; WRITE_SIZE — separate passes, as the MI355X guide's HBM section prescribes; gfx950: FETCH_SIZE counts 128-byte
    requests as 64 -> x 2), three frames each; per kernel the instantiation that runs on every frame, mean over its frame-sized launches
    (tools/pmc_round.py).  Returns {"roofline": {...}, "roofline_painter": {...}} to merge into the line, or None — and the committed
    summary's figures stay — when rocprofv3 is missing, a pass fails or takes more than two minutes, or this run is itself profiled."""
    import shutil
    if not shutil.which("rocprofv3"):
        return None
    if any("ROCPROF" in k or k.startswith("ROCP_") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None                                                   # (this run is being profiled itself: no profiler inside a profiler)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_round
        t0 = time.perf_counter()
        kern, line = pmc_round.collect(pmc_round.PASSES, timeout=120)
        if not line:
            return None
        n = line["config"]["pixel_segments"]
        w, h = line["config"]["canvas"]

        def per_frame(base):                                          # the instantiation of a kernel that runs on every frame
            cands = [v for k, v in kern.items() if k.split("<")[0] == base]
            return max(cands, key=lambda v: v.get("dispatches", 0)) if cands else None
        src = ("measured in this run: child runs of this command under `rocprofv3 --pmc` (separate passes for the SQ counters, FETCH_SIZE and "
               f"WRITE_SIZE; --in-flight 1, three frames; gfx950: 2 x FETCH_SIZE + WRITE_SIZE); {round(time.perf_counter() - t0, 1)} s")
        res = {"roofline": {}, "roofline_painter": {}}
        k = per_frame("k_onesweep")
        if not k or "hbm_bytes_per_launch" not in k:
            return None
        res["roofline"].update({"traffic": int(k["hbm_bytes_per_launch"]), "traffic_source": src + f"; mean over {k.get('dispatches', 0)} k_onesweep launches",
                                "traffic_over_algorithmic": round(k["hbm_bytes_per_launch"] / (16.0 * n), 4) if n else None})
        post = [per_frame(b) for b in ("k_runs_count", "k_runs_wave", "k_carry_rows", "k_paint_wave") if b in out.get("kernels_us", {})]
        if post and all(p_ and p_.get("hbm_bytes_per_launch") for p_ in post):
            res["roofline"]["post_sort_traffic_ratio"] = round(sum(p_["hbm_bytes_per_launch"] for p_ in post) / (8.0 * n + 4.0 * w * h), 3)
        pk_name = next((b for b in ("k_paint_wave", "k_paint_quad") if b in out.get("kernels_us", {})), None)
        pk = per_frame(pk_name) if pk_name else None
        paint_us = out["kernels_us"][pk_name]["us_per_frame"] if pk_name else 0.0
        if pk and pk.get("SQ_INSTS_VALU") and paint_us > 0:
            ach = pk["SQ_INSTS_VALU"] / paint_us / 1e3                 # G wave-instructions / s (the launch time: this run's own events)
            res["roofline_painter"].update({"valu_wave_instructions_per_launch": round(pk["SQ_INSTS_VALU"], 1), "achieved": round(ach, 1),
                                            "frac": round(ach / VALU_PEAK_GINST, 4),
                                            "frac_of_measured": round(ach * PAINT_SLOTS_PER_INST / VALU_MEASURED_GINST, 4), "counters_source": src})
            if pk.get("SQ_LDS_IDX_ACTIVE"):
                res["roofline_painter"]["lds_bank_conflict_ratio"] = round(pk.get("SQ_LDS_BANK_CONFLICT", 0) / pk["SQ_LDS_IDX_ACTIVE"], 4)
        return res
    except Exception as e:                                            # noqa: BLE001 (never lose the line to the counters)
        print(f"bench.py: live counter passes failed ({e!r}); the counter figures are the committed summary's", file=sys.stderr)
        return None


def animated_leg(local, frames=600):
    """BASELINE config 5 on one GPU: the deterministic spaceship (forma_amd/spaceship.py = the reference demo's game logic,
    fixed dt = 1/60 s, 600 frames: `enemy_count(t)` keeps growing that long, spaceship.rs:213-216) at 3840 x 2160, BGR1, clear
    (1, 1, 1, 0), through the product API exactly as the demo's runner does (demo/src/runner.rs:150-165): compose, then render
    into a caller buffer — with one persistent BufferLayerCache (damage tracking) and without a cache.  The game logic runs on
    the host between frames and is not timed."""
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
        r._ctx.close()
    return {"workload": "spaceship (deterministic re-implementation of demo/src/demos/spaceship.rs, 3840x2160, BGR1)", "frames": frames,
            "fps_no_cache": res["no_cache"], "fps_with_cache": res["with_cache"], "damaged_tile_fraction": res["damaged_tile_fraction"],
            "actors_at_end": res["actors_at_end"],
            "unit": "frames/s of Renderer::render into a caller buffer (table upload + D2H of the written tiles included)"}


def cpu_baseline(renderer, width, height, budget_s):
    """The CPU oracle (C++ restatement of forma's CPU backend: OpenMP over lines / pixel segments / tile rows, parallel
    stable radix sort, parallel prefix sum) timed on the host cores on the SAME scene tables the GPU rendered.  The thread
    count is the fastest of a short sweep; threads are pinned (OMP_PROC_BIND=close, OMP_PLACES=cores) and the big per-frame
    arrays are first-touched by the loops that fill them, so pages live next to the threads that use them.  Reported baseline
    only — a restatement, not forma's own Rayon/SIMD build (no Rust toolchain in this image)."""
    from oracle import oracle as orc
    hw = orc.lib().oracle_max_threads()
    t = renderer.host_tables
    o = orc.Oracle(threads=1)
    o.set_geometry(t["x"], t["y"], t["line_slot"]); o.set_geoms(t["geoms"])
    o.set_styles(t["style_offsets"], t["style_words"], None); o.set_images(t["images"], t["texels"])
    cands = sorted({c for c in (8, 16, 32, 48, 64, 96, 128, hw) if c <= hw} or {hw})
    t_start = time.perf_counter()
    sweep, per_stage = {}, {}
    for c in cands:
        o.set_threads(c)
        o.time_frame(width, height, 1)                      # first touch of this thread count's buffers / thread pool
        tm = o.time_frame(width, height, 1)
        sweep[c], per_stage[c] = sum(tm.values()), tm
        if time.perf_counter() - t_start > budget_s * 0.5:
            break
    best = min(sweep, key=sweep.get)
    # every stage at ITS fastest thread count of the sweep: the radix sort stops scaling where memory bandwidth does, the
    # painter (135 tile rows, one task each — forma's own granularity: painter/mod.rs:717-778) goes on for a while
    stage_best = {st: min(per_stage, key=lambda c: per_stage[c][st]) for st in ("prepare", "rasterize", "sort", "paint")}
    o.set_threads(best)
    o.set_stage_threads(**stage_best)
    o.time_frame(width, height, 1)
    per = sum(o.time_frame(width, height, 1).values())
    if per > sweep[best]:                                     # (mixing thread counts lost: stay with the best single count)
        o.set_stage_threads(0, 0, 0, 0)
        stage_best = {st: best for st in stage_best}
        per = sweep[best]
    left = max(1.0, budget_s - (time.perf_counter() - t_start))
    iters = max(3, min(40, int(left / max(per, 1e-3))))
    tm = o.time_frame(width, height, iters)
    per = sum(tm.values())
    cores = max(stage_best.values())
    quota = None                                              # the container's CPU quota (cgroup v2 cpu.max / v1 cfs): threads beyond it only queue
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else round(int(q) / int(period), 2)
    except Exception:                                         # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else round(q / period, 2)
        except Exception:                                     # noqa: BLE001
            pass
    return {"value": round(1.0 / per, 3), "unit": "frames/s", "cores": cores, "kind": "port", "host_threads": hw, "host_cpu_quota": quota,
            "sample": f"{iters} full frames of the same workload (C++ restatement of the CPU backend, OpenMP, pinned; every stage at the "
                      f"fastest thread count of the sweep {sorted(sweep)}: {stage_best}; parallel stable radix sort and prefix sum; the "
                      f"painter works a tile row per task like forma's)" + (f"; the box's cgroup grants {quota:g} CPUs of its {hw} hardware threads, which "
                                                                        f"is where the sweep stops improving" if quota else ""),
            "threads_per_stage": stage_best,
            "thread_sweep_ms": {str(k): round(v * 1e3, 1) for k, v in sweep.items()},
            "thread_sweep_stage_ms": {str(k): {st: round(v * 1e3, 1) for st, v in per_stage[k].items()} for k in per_stage},
            "stages_ms": {k: round(v * 1e3, 2) for k, v in tm.items()}}


if __name__ == "__main__":
    main()
