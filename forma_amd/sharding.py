"""Tile-row sharding of a frame across GPUs: the HOST logic of the process-per-GPU launcher.

`tile_y` is the most significant field of the pixel-segment key and the cover carry never crosses tile rows
(reference forma/src/cpu/painter/mod.rs:518-522, 741-776: rows are the CPU backend's unit of parallelism too), so a
band of tile rows is a complete, independent sort + paint problem.  Three layouts are driven from here:

* `exchange` (the north star's): every rank rasterizes 1/world of the LINES (`line_shares`, `slice_geometry`), HIP kernels
  bucket the pixel segments by tile-row owner, one padded equal-split all-to-all moves them (`ExchangeFrame`: RCCL through
  torch.distributed on the context's own stream; `pair_capacity`, `max_pair_count` size the buckets), the owner sorts and
  paints its band (`band_edges`, `agree_on_bands`, `band_crop`);
* `bands`: replicated scene, band culling, no data-path collective;
* `frames`: whole frames per GPU.

The single-process form of the exchange layout — `forma_hip_create_multi`, RCCL inside libforma_hip.so — needs none of
this module: `forma_hip_render` on a multi-device context plans bands, line shares and capacities itself (csrc/multi.cpp).
This module has no GPU dependency (the collectives are `torch.distributed`: RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def row_histogram(segments: np.ndarray, tiles_h: int) -> np.ndarray:
    """Pixel segments per tile row of a u64 stream (rows outside [0, tiles_h) are never painted)."""
    ty = (segments >> np.uint64(53)).astype(np.int64) - 1
    ty = ty[(ty >= 0) & (ty < tiles_h)]
    return np.bincount(ty, minlength=tiles_h)[:tiles_h].astype(np.int64)


def band_edges(row_hist: Sequence[int], n: int) -> List[int]:
    """n contiguous, non-empty tile-row bands [e[r], e[r+1]) with (nearly) equal pixel-segment counts.
    Needs len(row_hist) >= n."""
    tiles_h = len(row_hist)
    if n < 1 or tiles_h < n:
        raise ValueError(f"cannot cut {tiles_h} tile rows into {n} bands")
    cum = np.cumsum(np.asarray(row_hist, np.float64))
    total = float(cum[-1]) if tiles_h else 0.0
    edges = [0]
    for r in range(1, n):
        if total > 0:
            e = int(np.searchsorted(cum, total * r / n, side="left")) + 1
        else:
            e = (tiles_h * r) // n
        e = max(e, edges[-1] + 1)                 # every band keeps at least one row ...
        e = min(e, tiles_h - (n - r))             # ... and leaves one for each band after it
        edges.append(e)
    edges.append(tiles_h)
    return edges


def agree_on_bands(dist, row_hist, world: int, device=None) -> List[int]:
    """Rank 0's partition, broadcast (every rank rasterizes the same scene, but only one histogram decides)."""
    import torch
    edges = band_edges(row_hist, world)
    t = torch.tensor(edges, dtype=torch.int64, device=device)
    dist.broadcast(t, 0)
    return [int(v) for v in t.cpu().tolist()]


def band_crop(edges: Sequence[int], rank: int, width: int, height: int) -> Tuple[int, int, int, int]:
    """(x0, x1, y0, y1) in pixels of rank's band, clipped to the canvas."""
    return 0, width, edges[rank] * 16, min(edges[rank + 1] * 16, height)


def max_over_ranks(dist, seconds: float, device=None) -> float:
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- exchange mode: the north star's literal layout ------------------------------------------------------------------
# Stage 2 emits segments in LINE order, stages 3-4 own them by TILE ROW.  In `bands` mode every rank rasterizes the whole
# (replicated) scene and keeps its band; in `exchange` mode every rank rasterizes 1/world of the LINES and the pixel
# segments travel to the rank that owns their tile row: one all-to-all of u64 payloads (RCCL over xGMI on GPUs; each pair
# of GPUs has its own link, so the exchange is per-link bound), then each rank sorts and paints its band.

def line_shares(inclusive_sums: Sequence[int], world: int) -> List[int]:
    """Cut lines 0..n into `world` contiguous ranges [c[r], c[r+1]) carrying (nearly) equal pixel-segment counts.
    `inclusive_sums[i]` = pixel segments of lines 0..i — the `lengths` array of forma_hip_prepare_lines (the reference's
    `lengths` after its prefix sum, segment.rs:90-98)."""
    cum = np.asarray(inclusive_sums, np.float64)
    n = len(cum)
    total = float(cum[-1]) if n else 0.0
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(cum, total * r / world, side="left")) + 1 if total > 0 else (n * r) // world
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return cuts


def slice_geometry(x: np.ndarray, y: np.ndarray, line_slot: np.ndarray, l0: int, l1: int):
    """The points and slots of lines [l0, l1): line i joins points i and i + 1, so the slice keeps one point more."""
    if l1 <= l0:
        return x[:0].copy(), y[:0].copy(), line_slot[:0].copy()
    return x[l0:l1 + 1].copy(), y[l0:l1 + 1].copy(), line_slot[l0:l1].copy()


# ---- the exchange frame, device-side (include/forma_hip.h "multi-GPU, exchange layout") -------------------------------------
def pair_capacity(max_pair_count: int) -> int:
    """Slots per (sender, owner) bucket: the largest bucket seen, 6 % slack, a multiple of 2048 (whole rasterizer blocks)."""
    c = int(max_pair_count) + int(max_pair_count) // 16 + 4096
    return (c + 2047) // 2048 * 2048


class ExchangeFrame:
    """One rank's side of `bench.py --mode exchange`: lines [cuts[rank], cuts[rank + 1]) are rasterized here, bucketed by
    tile-row owner with HIP kernels (no torch op on the data path), ONE equal-split all-to-all of the bucket counts and one
    of the padded buckets run on the context's own stream (RCCL through torch.distributed, ordered by events: no host
    synchronisation), then this rank gathers what it received, sorts and paints its band."""

    def __init__(self, ctx, dist, rank: int, world: int, edges: Sequence[int], width: int, height: int, capacity: int):
        import torch
        self.ctx, self.dist, self.rank, self.world = ctx, dist, rank, world
        self.edges = [int(e) for e in edges]
        self.width, self.height = width, height
        self.crop = band_crop(self.edges, rank, width, height)
        ctx.exchange_plan(self.edges, capacity)
        self.send, self.recv, self.words_per_pair = ctx.exchange_views()      # a bucket = capacity segments + its header word
        self.stream = torch.cuda.ExternalStream(ctx.stream_handle(), device=torch.device("cuda", ctx.device))

    def frame(self, channels=(0, 1, 2, 3), clear=(1, 1, 1, 1), dst=None, stride=None, timings=False, device_only=True):
        import torch
        t1 = self.ctx.rasterize_bucket_frame(self.width, self.height, timings=timings)
        if self.world > 1 and self.dist.get_backend() == "nccl":
            with torch.cuda.stream(self.stream):                       # ONE collective, ordered after the bucket kernels, before the gather
                self.dist.all_to_all_single(self.recv, self.send)          # (equal splits of words_per_pair: the headers travel with the data)
        elif self.world > 1:                                           # rehearsal without RCCL (gloo): staged through host memory
            self.stream.synchronize()
            h_in, h_out = self.send.cpu(), torch.empty_like(self.send, device="cpu")
            self.dist.all_to_all_single(h_out, h_in)
            with torch.cuda.stream(self.stream):
                self.recv.copy_(h_out)
            self.stream.synchronize()
        r = self.ctx.gather_sort_paint_frame(self.width, self.height, channels=channels, clear=clear, crop=self.crop, dst=dst,
                                             stride=stride, timings=timings, device_only=device_only)
        if not timings:
            return r
        t2 = r[1]
        for k in ("prepare_us", "rasterize_us"):
            t2[k] = t1[k]
        t2["exchange_us"] = t2.get("exchange_us", 0.0) + t1.get("exchange_us", 0.0)
        t2["total_us"] = t2["total_us"] + t1["total_us"]
        t2["n_lines"] = t1["n_lines"]
        return r[0], t2


def max_pair_count(dist, local_segments: np.ndarray, edges: Sequence[int], world: int, device=None) -> int:
    """The largest (sender, owner) bucket of a frame: this rank's segments per owner band, maximum over all ranks."""
    import torch
    ty = (local_segments >> np.uint64(53)).astype(np.int64) - 1
    keep = (ty >= edges[0]) & (ty < edges[-1])
    owner = np.searchsorted(np.asarray(edges[1:-1], np.int64), ty[keep], side="right")
    mx = int(np.bincount(owner, minlength=world).max()) if keep.any() else 0
    if dist is None or world == 1:
        return mx
    t = torch.tensor([mx], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())
