"""Multi-GPU host logic on CPU: tile-row band partition (pure functions) and the N > 1 protocol of bench.py —
band agreement by broadcast, band-cropped rendering, disjoint row ownership, max-over-ranks timing — run with
world_size 2 on the gloo backend.  The per-band renderer here is the CPU oracle (test infrastructure); on GPUs the
same protocol drives forma_hip_set_band + forma_hip_render."""
import os
import socket

import numpy as np
import pytest

from forma_amd import sharding


def test_band_edges_properties():
    rng = np.random.default_rng(0)
    for tiles_h, n in [(135, 1), (135, 2), (135, 8), (8, 8), (512, 8), (9, 4)]:
        for trial in range(20):
            hist = rng.integers(0, 1000, tiles_h) * (rng.random(tiles_h) < 0.7)
            e = sharding.band_edges(hist, n)
            assert len(e) == n + 1 and e[0] == 0 and e[-1] == tiles_h
            assert all(e[i] < e[i + 1] for i in range(n)), e            # contiguous, non-empty
    # balance: no band carries more than ideal + the heaviest row
    hist = rng.integers(100, 200, 135)
    e = sharding.band_edges(hist, 8)
    loads = [hist[e[i]:e[i + 1]].sum() for i in range(8)]
    assert max(loads) <= hist.sum() / 8 + hist.max()
    with pytest.raises(ValueError):
        sharding.band_edges([1, 2, 3], 4)
    assert sharding.band_edges(np.zeros(16), 4) == [0, 4, 8, 12, 16]    # empty frame: equal heights


def test_row_histogram_ignores_unpainted_rows():
    def seg(ty):
        return np.uint64((ty + 1) << 53) if ty + 1 >= 0 else np.uint64(0)
    v = np.array([seg(-1), seg(0), seg(0), seg(3), seg(7), seg(200)], np.uint64)
    assert sharding.row_histogram(v, 8).tolist() == [2, 0, 0, 1, 0, 0, 0, 1]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import scene as S
    from oracle import oracle as orc
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    W, H = 200, 150
    tiles_h = (H + 15) // 16
    o = orc.Oracle()
    comp = S.random_mixed(n=120, width=W, height=H, seed=5)
    t = comp.tables(o)
    S.load(o, t)
    full = o.render(W, H, clear=(0.1, 0.2, 0.3, 1.0))
    # every rank has the same scene; rank 0's histogram decides the bands
    hist = sharding.row_histogram(o.segments(0), tiles_h)
    if rank != 0:
        hist = np.zeros_like(hist)                       # a wrong local opinion must not matter
    edges = sharding.agree_on_bands(dist, hist, world)
    x0, x1, y0, y1 = sharding.band_crop(edges, rank, W, H)
    band = o.render(W, H, clear=(0.1, 0.2, 0.3, 1.0), crop=(x0, x1, y0, y1), dst=np.full((H, W * 4), 7, np.uint8))
    assert (band[:y0] == 7).all() and (band[y1:] == 7).all()          # a rank writes only its own rows
    assert np.array_equal(band[y0:y1], full[y0:y1])
    elapsed = sharding.max_over_ranks(dist, 1.0 + rank)                # MAX over ranks, as bench.py reports
    assert elapsed == float(world)
    gathered = [None] * world
    dist.all_gather_object(gathered, (edges, y0, y1))
    assert all(g[0] == edges for g in gathered)
    rows = sorted((g[1], g[2]) for g in gathered)
    assert rows[0][0] == 0 and rows[-1][1] == H and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
    np.save(os.path.join(out_dir, f"band{rank}.npy"), band[y0:y1])
    if rank == 0:
        np.save(os.path.join(out_dir, "full.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_band_protocol_gloo(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    stitched = np.concatenate([np.load(tmp_path / f"band{r}.npy") for r in range(world)])
    assert np.array_equal(stitched, full)


def test_line_shares_and_geometry_slices():
    rng = np.random.default_rng(1)
    lengths = rng.integers(0, 50, 1000)
    for world in (1, 2, 3, 8):
        c = sharding.line_shares(np.cumsum(lengths), world)
        assert len(c) == world + 1 and c[0] == 0 and c[-1] == 1000 and all(c[i] <= c[i + 1] for i in range(world))
        loads = [lengths[c[i]:c[i + 1]].sum() for i in range(world)]
        assert max(loads) <= lengths.sum() / world + 2 * lengths.max()
    x = np.arange(11, dtype=np.float32); y = x * 2; ls = np.arange(10, dtype=np.uint32)
    sx, sy, sl = sharding.slice_geometry(x, y, ls, 3, 7)
    assert sx.tolist() == [3, 4, 5, 6, 7] and sl.tolist() == [3, 4, 5, 6] and len(sy) == 5
    assert all(len(a) == 0 for a in sharding.slice_geometry(x, y, ls, 4, 4))


def _padded_worker(rank, world, port, out_dir):
    """The device-side exchange protocol (forma_hip_rasterize_bucket_frame -> ONE equal-split all-to-all of the padded
    buckets, each carrying its header -> forma_hip_gather_sort_paint_frame) with the bucket / gather kernels restated in numpy: what crosses the
    collective, and in which layout, is exactly what sharding.ExchangeFrame moves over RCCL."""
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import scene as S
    from oracle import oracle as orc
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    W, H = 200, 150
    tiles_h = (H + 15) // 16
    clear = (0.1, 0.2, 0.3, 1.0)
    comp = S.Composition(insertion_order=True)                          # layers pushed out of paint order on purpose
    rng = np.random.default_rng(3)
    for order in rng.permutation(40):
        comp.get_mut_or_insert_default(int(order)).insert(S.custom_circle(float(rng.uniform(0, W)), float(rng.uniform(0, H)), float(rng.uniform(8, 60)))) \
            .set_props(S.solid((float(rng.random()), float(rng.random()), float(rng.random()), 0.7)))
    ref = orc.Oracle()
    t = comp.tables(ref)
    S.load(ref, t)
    full = ref.render(W, H, clear=clear)
    full_stream = ref.segments(0)
    sums = ref.prepare_lines(W, H)["lengths"]
    edges = sharding.agree_on_bands(dist, sharding.row_histogram(full_stream, tiles_h), world)
    cuts = sharding.line_shares(sums, world)
    o = orc.Oracle()
    S.load(o, t)
    o.set_geometry(*sharding.slice_geometry(t["x"], t["y"], t["line_slot"], cuts[rank], cuts[rank + 1]))
    o.prepare_lines(W, H)
    mine = o.rasterize()
    cap = sharding.pair_capacity(sharding.max_pair_count(dist, mine, edges, world))
    # k_owner_count / _scan / _scatter: stable partition into padded buckets + (count, overflow) pairs
    ty = (mine >> np.uint64(53)).astype(np.int64) - 1
    owner = np.searchsorted(np.asarray(edges[1:-1], np.int64), ty, side="right")
    owner[(ty < edges[0]) | (ty >= edges[-1])] = world
    w = cap + 1                                                         # a bucket: cap segments + its header word {count | overflow << 32}
    send = np.zeros(world * w, np.int64)
    for g in range(world):
        b = mine[owner == g]
        assert len(b) <= cap
        send[g * w: g * w + len(b)] = b.view(np.int64); send[g * w + cap] = len(b)
    recv = torch.zeros(world * w, dtype=torch.int64)
    dist.all_to_all_single(recv, torch.from_numpy(send))                # ONE equal-split collective: the headers travel with the data
    # k_gather_chunks: rank-major concatenation of the valid prefix of every bucket
    rn = recv.numpy()
    assert all((int(rn[s * w + cap]) >> 32) == 0 for s in range(world))
    got = np.concatenate([rn[s * w: s * w + (int(rn[s * w + cap]) & 0xFFFFFFFF)] for s in range(world)]).view(np.uint64)
    tyf = (full_stream >> np.uint64(53)).astype(np.int64) - 1
    want = full_stream[(tyf >= edges[rank]) & (tyf < edges[rank + 1])]
    assert np.array_equal(got, want)                                   # the band's slice of the single-device stream, same order
    layers = (got >> np.uint64(20)) & np.uint64(0x1FFFFF)
    assert (np.diff(layers.astype(np.int64)) < 0).any()                 # ... and it is NOT layer-sorted: the sort needs the layer digits
    srt = got[np.argsort(got >> np.uint64(20), kind="stable")]
    x0, x1, y0, y1 = sharding.band_crop(edges, rank, W, H)
    band = ref.paint(srt, W, H, clear=clear, crop=(x0, x1, y0, y1), dst=np.full((H, W * 4), 7, np.uint8))
    assert np.array_equal(band[y0:y1], full[y0:y1])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_padded_bucket_exchange_gloo(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_padded_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def test_pair_capacity_and_max_pair_count():
    assert sharding.pair_capacity(0) == 4096 and sharding.pair_capacity(100000) % 2048 == 0 and sharding.pair_capacity(100000) >= 106000
    def seg(ty):
        return np.uint64((ty + 1) << 53)
    v = np.array([seg(0), seg(1), seg(1), seg(5), seg(9), seg(9), seg(9), seg(40)], np.uint64)
    assert sharding.max_pair_count(None, v, [0, 2, 6, 10], 3) == 3
