"""Tile-row sharding of a frame across GPUs.

`tile_y` is the most significant field of the pixel-segment key and the cover carry never crosses tile rows
(reference forma/src/cpu/painter/mod.rs:518-522, 741-776: rows are the CPU backend's unit of parallelism too), so a
frame splits into contiguous bands of tile rows with no data-path exchange: every rank holds the whole (small)
scene, culls lines outside its band and rasterizes / sorts / paints only its own rows.  This module is the host
logic: choose the bands, agree on them across ranks, and time the frame the way bench.py reports it.  It has no
GPU dependency (the collectives are `torch.distributed`: RCCL on GPUs, gloo in the CPU tests).
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
