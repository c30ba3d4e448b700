"""Multi-GPU exchange layout (include/forma_hip.h last section) on the GPU: rasterize a share of the lines,
bucket the pixel segments by tile-row owner with the HIP kernels, move the buckets, gather + sort + paint the band.

* one rank: the whole pipeline against the oracle and against a plain render;
* G ranks on ONE device: G contexts play the ranks, the all-to-all is done by hand with device copies between their
  bucket buffers — every kernel of the exchange path runs exactly as it would on G devices, the bands must stitch to the
  single-device frame and each rank's sorted stream must equal the oracle's stream restricted to its band;
* layers pushed out of paint order (a slice can look layer-sorted while the gathered stream is not);
* a real 2-process RCCL run when the box has two GPUs (skipped on a single-GPU box)."""
import os
import socket

import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def scene_tables(comp):
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(o, t)
    return o, t


def run_ranks(t, W, H, G, clear, frames=2, check=None, geoms_per_frame=None):
    """G contexts on device 0 play G ranks; returns per rank (image rows of its band, its sorted stream) of the LAST frame,
    and the band edges.  Frame 1 is synchronous on every rank, the following ones are read-back-free (sort in place through
    the chunk map).  `check(frame, out, edges)` is called after every frame."""
    import torch
    import forma_amd
    from forma_amd import sharding
    tiles_h = (H + 15) // 16
    ref = forma_amd.Context(0)
    S.load(ref, t)
    lengths = ref.prepare_lines(W, H)["lengths"]
    ref.render(W, H, clear=clear)
    full_stream = ref.segments(0)
    ref.close()
    edges = sharding.band_edges(sharding.row_histogram(full_stream, tiles_h), G)
    cuts = sharding.line_shares(lengths, G)
    ctxs, mx = [], 0
    for r in range(G):
        c = forma_amd.Context(0)
        S.load(c, t)
        c.set_geometry(*sharding.slice_geometry(t["x"], t["y"], t["line_slot"], cuts[r], cuts[r + 1]))
        c.rasterize_frame(W, H)
        mx = max(mx, sharding.max_pair_count(None, c.segments(0), edges, G))
        ctxs.append(c)
    cap = sharding.pair_capacity(mx)
    xs = [sharding.ExchangeFrame(c, None, r, G, edges, W, H, cap) for r, c in enumerate(ctxs)]
    out = None
    for f in range(frames):                                            # frame 1 synchronous, the others read-back-free
        if geoms_per_frame is not None:
            for c in ctxs:
                c.set_geoms(geoms_per_frame[f])
        for x in xs:
            x.ctx.rasterize_bucket_frame(W, H)
        torch.cuda.synchronize()
        for r, x in enumerate(xs):                                     # the all-to-all, by hand: recv[r][s] = send[s][r]
            w = x.words_per_pair                                       # (a bucket: `cap` segments + its header word)
            assert w == cap + 1
            for s, y in enumerate(xs):
                if G > 1:
                    x.recv[s * w:(s + 1) * w].copy_(y.send[r * w:(r + 1) * w])
        torch.cuda.synchronize()
        out = []
        for r, x in enumerate(xs):
            img = np.full((H, W * 4), 7, np.uint8)
            x.ctx.gather_sort_paint_frame(W, H, clear=clear, crop=x.crop, dst=img, device_only=False)
            y0, y1 = x.crop[2], x.crop[3]
            assert (img[:y0] == 7).all() and (img[y1:] == 7).all()    # a rank writes only its own rows
            out.append((img[y0:y1].copy(), x.ctx.segments(1)))
        if check is not None:
            check(f, out, edges)
    for c in ctxs:
        c.close()
    return out, edges


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_exchange_path_with_G_ranks_on_one_device(G):
    W, H = 512, 384
    clear = (0.2, 0.3, 0.4, 1.0)
    o, t = scene_tables(S.random_mixed())
    want = o.render(W, H, clear=clear)
    sorted_full = o.segments(1)
    out, edges = run_ranks(t, W, H, G, clear)
    stitched = np.concatenate([img for img, _ in out])
    assert np.array_equal(stitched, want)
    ty = (sorted_full >> np.uint64(53)).astype(np.int64) - 1
    for r, (_, srt) in enumerate(out):                                 # bit-exact sorted stream of the band
        assert np.array_equal(srt, sorted_full[(ty >= edges[r]) & (ty < edges[r + 1])]), r


def test_exchange_with_layers_pushed_out_of_paint_order():
    """Geometry appended in insertion order [5, 1, 9, 3, 0]: each rank's slice may be layer-sorted by itself while the
    gathered band is not — the gather kernel checks the stream it actually sorts (api.cpp sort_paint_frame)."""
    W, H = 320, 208
    comp = S.Composition(insertion_order=True)
    for order, (x, y) in zip((5, 1, 9, 3, 0), ((60, 60), (120, 80), (180, 100), (240, 120), (150, 150))):
        comp.get_mut_or_insert_default(order).insert(S.custom_circle(x, y, 70)).set_props(S.solid((order / 9, 0.4, 0.6, 0.7)))
    o, t = scene_tables(comp)
    want = o.render(W, H)
    out, _ = run_ranks(t, W, H, 3, (1, 1, 1, 0))
    assert np.array_equal(np.concatenate([img for img, _ in out]), want)


def test_exchange_capacity_overflow_is_reported():
    import forma_amd
    from forma_amd import sharding, FormaError
    W, H = 256, 160
    o, t = scene_tables(S.random_mixed(n=80, width=W, height=H, seed=2))
    c = forma_amd.Context(0)
    S.load(c, t)
    x = sharding.ExchangeFrame(c, None, 0, 1, [0, (H + 15) // 16], W, H, 2048)     # far too small on purpose
    with pytest.raises(FormaError) as e:
        x.frame(device_only=True)
    assert e.value.code == -4
    c.rasterize_frame(W, H)
    cap = sharding.pair_capacity(sharding.max_pair_count(None, c.segments(0), x.edges, 1))
    x = sharding.ExchangeFrame(c, None, 0, 1, x.edges, W, H, cap)                  # re-planned
    img = x.frame(clear=(1, 1, 1, 0), device_only=False, dst=np.zeros((H, W * 4), np.uint8))
    assert np.array_equal(img, o.render(W, H))
    c.close()


def _full_size(workload):
    """scene tables of a BASELINE workload (flattened on the GPU through the product API) + the oracle's frame"""
    from forma_amd import api, scenes
    fn, W, H = scenes.WORKLOADS[workload]
    r = api.Renderer(0)
    r.render(fn(), api.BufferBuilder(np.zeros(W * H * 4, np.uint8), api.LinearLayout(W, W * 4, H)).build(), api.RGBA,
             api.Color(1, 1, 1, 1), None)
    t = dict(r.host_tables)
    r._ctx.close()
    o = orc.Oracle()
    S.load(o, t)
    want = o.render(W, H, clear=(1.0, 1.0, 1.0, 1.0))
    return t, W, H, want, o.segments(1)


@pytest.fixture(scope="module")
def triangles_10m():
    return _full_size("triangles-10m-8k")


@pytest.fixture(scope="module")
def paris_like():
    return _full_size("paris-like-30k-4k")


def _check_full(G, data, frames=3):
    """every rank's sorted stream == the oracle's sorted stream restricted to its band, bands stitch to the oracle's image
    within one code value — on the synchronous first frame AND on the read-back-free frames (in-place sort of the received
    buckets through the chunk map, sort.hip k_sort_hist<true> / k_onesweep<8, true>)"""
    t, W, H, want, sorted_full = data
    ty = (sorted_full >> np.uint64(53)).astype(np.int64) - 1
    seen = []

    def check(f, out, edges):
        stitched = np.concatenate([img for img, _ in out])
        d = np.abs(stitched.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1, (f, int(d.max()))
        for r, (_, srt) in enumerate(out):
            band = sorted_full[(ty >= edges[r]) & (ty < edges[r + 1])]
            assert len(srt) == len(band), (f, r, len(srt), len(band))
            assert np.array_equal(srt, band), (f, r)
        seen.append(f)

    run_ranks(t, W, H, G, (1.0, 1.0, 1.0, 1.0), frames=frames, check=check)
    assert seen == list(range(frames))


@pytest.mark.parametrize("G", [2, 8])
def test_exchange_full_size_triangles_10m_8k(G, triangles_10m):
    """BASELINE config 4 (10 M pixel segments, 8192 x 8192 — the configuration the 8-GPU target is quoted on) through the
    exchange path with G emulated ranks: buckets of > 2048 blocks, multi-tile look-back over the chunk-mapped first pass."""
    _check_full(G, triangles_10m)


@pytest.mark.parametrize("G", [2, 8])
def test_exchange_full_size_paris_like_30k_4k(G, paris_like):
    """BASELINE config 3 stand-in (13.8 M pixel segments, 30 000 layers, gradients + blends) through the exchange path."""
    _check_full(G, paris_like)


def test_exchange_local_count_outgrows_its_bound_between_frames():
    """a read-back-free bucket frame provisions for the previous frame's local segment count + 6 %.  When
    a transform makes the scene grow past that, the excess is never rasterized — the frame must FAIL with FORMA_E_CAPACITY
    (k_owner_scan flags it to every receiver) instead of returning a wrong image; after a re-plan it renders right."""
    import forma_amd
    from forma_amd import sharding, FormaError
    if "sync" in os.environ.get("FORMA_HIP_DEBUG", ""):
        pytest.skip("FORMA_HIP_DEBUG=sync: no read-back-free frames, nothing is provisioned from a previous frame")
    W, H = 512, 384
    o, t = scene_tables(S.random_mixed())
    tiles_h = (H + 15) // 16
    edges = [0, tiles_h]

    def scaled(k, tx=0.0, ty=0.0):
        g = t["geoms"].copy()
        g["flags"] = 1
        g["xf"] = np.array([k, 0.0, 0.0, k, tx, ty], np.float32)      # (overrides the scene's own transforms: same for both backends)
        return g

    c = forma_amd.Context(0)
    S.load(c, t)
    g = scaled(0.5, 100.0, 80.0)                                       # the scene at half size ...
    o.set_geoms(g); c.set_geoms(g)
    want = o.render(W, H)
    n0 = len(o.segments(0))
    x = sharding.ExchangeFrame(c, None, 0, 1, edges, W, H, 8 * sharding.pair_capacity(n0))   # (bucket capacity is NOT what overflows)
    for _ in range(3):                                                 # synchronous, then read-back-free
        img = x.frame(clear=(1, 1, 1, 0), device_only=False, dst=np.zeros((H, W * 4), np.uint8))
        assert np.array_equal(img, want)
    g = scaled(0.9, 20.0, 10.0)                                        # ... grows: far more pixel segments than frame k - 1 + 6 %
    o.set_geoms(g); c.set_geoms(g)
    want2 = o.render(W, H)
    assert len(o.segments(0)) > n0 + n0 // 16 + 4096                 # (more than the read-back-free frame provisions for)
    with pytest.raises(FormaError) as e:
        x.frame(clear=(1, 1, 1, 0), device_only=False, dst=np.zeros((H, W * 4), np.uint8))
    assert e.value.code == -4
    x = sharding.ExchangeFrame(c, None, 0, 1, edges, W, H, 8 * sharding.pair_capacity(len(o.segments(0))))   # re-planned
    for _ in range(3):
        img = x.frame(clear=(1, 1, 1, 0), device_only=False, dst=np.zeros((H, W * 4), np.uint8))
        assert np.array_equal(img, want2)
    g = scaled(0.9, 23.0, 12.0)                                        # a drift the 6 % slack covers: stays read-back-free and right
    o.set_geoms(g); c.set_geoms(g)
    want3 = o.render(W, H)
    for _ in range(2):
        img = x.frame(clear=(1, 1, 1, 0), device_only=False, dst=np.zeros((H, W * 4), np.uint8))
        assert np.array_equal(img, want3)
    c.close()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rccl_worker(rank, world, port, out_dir):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import scene as S2
    import forma_amd
    from forma_amd import sharding
    from oracle import oracle as orc2
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    W, H = 512, 384
    tiles_h = (H + 15) // 16
    clear = (0.2, 0.3, 0.4, 1.0)
    o = orc2.Oracle()
    t = S2.random_mixed().tables(o)
    S2.load(o, t)
    want = o.render(W, H, clear=clear)
    c = forma_amd.Context(rank)
    S2.load(c, t)
    lengths = c.prepare_lines(W, H)["lengths"]
    c.render(W, H, clear=clear)
    edges = sharding.agree_on_bands(dist, sharding.row_histogram(c.segments(0), tiles_h), world, device="cuda")
    cuts = sharding.line_shares(lengths, world)
    c.set_geometry(*sharding.slice_geometry(t["x"], t["y"], t["line_slot"], cuts[rank], cuts[rank + 1]))
    c.rasterize_frame(W, H)
    cap = sharding.pair_capacity(sharding.max_pair_count(dist, c.segments(0), edges, world, device="cuda"))
    x = sharding.ExchangeFrame(c, dist, rank, world, edges, W, H, cap)
    for _ in range(3):
        img = x.frame(clear=clear, device_only=False, dst=np.full((H, W * 4), 7, np.uint8))
    y0, y1 = x.crop[2], x.crop[3]
    assert np.array_equal(img[y0:y1], want[y0:y1])
    assert (img[:y0] == 7).all() and (img[y1:] == 7).all()
    dist.barrier()
    dist.destroy_process_group()


def _rccl_world1_worker(rank, world, port, out_dir):
    """Everything `bench.py --mode exchange` asks of RCCL, on ONE GPU with a world of one: the process group on the `nccl`
    backend, the band agreement / maxima / time reduce on CUDA tensors, and the two equal-split all-to-alls on the context's
    own stream with the LIBRARY's buffers (memory torch did not allocate) as send and receive tensors."""
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import scene as S2
    import forma_amd
    from forma_amd import sharding
    from oracle import oracle as orc2
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    W, H = 512, 384
    tiles_h = (H + 15) // 16
    o = orc2.Oracle()
    t = S2.random_mixed().tables(o)
    S2.load(o, t)
    want = o.render(W, H)
    c = forma_amd.Context(0)
    S2.load(c, t)
    c.render(W, H)
    edges = sharding.agree_on_bands(dist, sharding.row_histogram(c.segments(0), tiles_h), 1, device="cuda")
    assert edges == [0, tiles_h]
    c.rasterize_frame(W, H)
    stream0 = c.segments(0)
    mx = sharding.max_pair_count(dist, stream0, edges, 1, device="cuda")
    assert sharding.max_over_ranks(dist, 1.5, device="cuda") == 1.5
    x = sharding.ExchangeFrame(c, dist, 0, 1, edges, W, H, sharding.pair_capacity(mx))
    c.rasterize_bucket_frame(W, H)
    x.recv.zero_()
    with torch.cuda.stream(x.stream):                                  # what ExchangeFrame.frame does when world > 1
        dist.all_to_all_single(x.recv, x.send)
    x.stream.synchronize()
    hdr = int(x.recv[x.words_per_pair - 1].item())                     # the bucket's header: count | overflow << 32
    n = hdr & 0xFFFFFFFF
    ty = (stream0 >> np.uint64(53)).astype(np.int64) - 1
    kept = stream0[(ty >= 0) & (ty < tiles_h)]
    assert n == len(kept) and (hdr >> 32) == 0
    assert np.array_equal(x.recv[:n].cpu().numpy().view(np.uint64), kept)     # the bucket arrived, in rasterizer order
    img = x.frame(clear=(1, 1, 1, 0), device_only=False, dst=np.zeros((H, W * 4), np.uint8))
    assert np.array_equal(img, want)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_rccl_calls_on_library_buffers_with_a_world_of_one():
    import torch.multiprocessing as mp
    mp.spawn(_rccl_world1_worker, args=(1, _free_port(), ""), nprocs=1, join=True)


@pytest.mark.timeout(300)                      # (first executed on a multi-GPU node by the driver: fail, never hang)
def test_two_ranks_over_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(2, _free_port(), ""), nprocs=2, join=True)
