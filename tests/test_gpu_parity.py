"""GPU parity: every stage of libforma_hip.so against the CPU oracle on identical inputs, through the
C ABI.  Bars (BASELINE.json north_star): unsorted and sorted u64 streams bit-exact; RGBA8 within 1
code value (observed: identical)."""
import os

import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

LINE_KEYS = ("orders", "x0", "y0", "dx", "dy", "a", "b", "c", "d", "lengths")
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_cpu_64x64.npz"))


@pytest.fixture(scope="module")
def ctx():
    import forma_amd
    c = forma_amd.Context(0)
    yield c
    c.close()


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def both(ctx, comp):
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(o, t); S.load(ctx, t)
    return o, t


SMALL = dict(S.e2e_scenes())


@pytest.fixture(scope="module")
def mixed():
    return S.random_mixed()


@pytest.mark.parametrize("name", sorted(SMALL))
def test_e2e_all_stages(ctx, name):
    o, _ = both(ctx, SMALL[name])
    lo = o.prepare_lines(64, 64); lg = ctx.prepare_lines(64, 64)
    for k in LINE_KEYS:
        assert bits_equal(lo[k], lg[k]), (name, k)
    so = o.rasterize(); sg = ctx.rasterize_lines(lo)
    assert np.array_equal(so, sg), name
    assert np.array_equal(o.sort(), ctx.sort_array(so)), name
    img_o = o.render(64, 64)
    img_g = ctx.render(64, 64)
    assert np.array_equal(ctx.segments(0), o.segments(0)) and np.array_equal(ctx.segments(1), o.segments(1))
    d = np.abs(img_o.astype(int) - img_g.astype(int))
    assert d.max() <= 1, (name, d.max(), int((d > 0).sum()))
    assert d.max() == 0, (name, "not bit-identical", int((d > 0).sum()))
    # and against the reference's own PNG golden (tolerance of e2e-tests/tests/test_env.rs:278)
    dg = np.abs(img_g.reshape(64, 64, 4).astype(int) - GOLD[name].astype(int))
    assert dg.max() <= 8


@pytest.mark.parametrize("size", [(512, 384), (500, 301)])
def test_mixed_scene(ctx, mixed, size):
    w, h = size
    o, _ = both(ctx, mixed)
    lo = o.prepare_lines(w, h); lg = ctx.prepare_lines(w, h)
    for k in LINE_KEYS:
        assert bits_equal(lo[k], lg[k]), k
    img_o = o.render(w, h, clear=(0.2, 0.3, 0.4, 1.0))
    img_g = ctx.render(w, h, clear=(0.2, 0.3, 0.4, 1.0))
    assert np.array_equal(ctx.segments(0), o.segments(0))
    assert np.array_equal(ctx.segments(1), o.segments(1))
    d = np.abs(img_o.astype(int) - img_g.astype(int))
    assert d.max() <= 1, (d.max(), int((d > 0).sum()))


@pytest.mark.parametrize("channels,clear", [(S.BGRA, (1, 1, 1, 0)), (S.RGB1, (0, 0, 0, 0.5)), (S.BGR0, (0.5, 0.25, 0.125, 1.0))])
def test_channels_and_clear(ctx, mixed, channels, clear):
    o, _ = both(ctx, mixed)
    a = o.render(256, 256, channels=channels, clear=clear)
    b = ctx.render(256, 256, channels=channels, clear=clear)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1


def test_crop(ctx, mixed):
    o, _ = both(ctx, mixed)
    crop = (40, 300, 33, 200)
    base = np.full((384, 512 * 4), 7, np.uint8)
    a = o.render(512, 384, crop=crop, dst=base.copy())
    b = ctx.render(512, 384, crop=crop)     # device image is written only inside the crop
    x0, x1, y0, y1 = (40 // 16) * 16, ((300 + 15) // 16) * 16, (33 // 16) * 16, ((200 + 15) // 16) * 16
    assert np.array_equal(a[y0:y1, x0 * 4:x1 * 4], b[y0:y1, x0 * 4:x1 * 4])
    assert (a[:y0] == 7).all() and (a[y1:] == 7).all()


def test_random_cubics_1080p(ctx):
    """BASELINE.json configs[1]: 1000 random cubic Beziers, solid fill, 1920x1080."""
    comp = S.random_cubics(1000, 1920, 1080)
    o, _ = both(ctx, comp)
    img_o = o.render(1920, 1080, clear=(1, 1, 1, 1))
    img_g, t = ctx.render(1920, 1080, clear=(1, 1, 1, 1), timings=True)
    u_o, u_g = o.segments(0), ctx.segments(0)
    assert len(u_o) == len(u_g) == t["n_segments"]
    assert np.array_equal(u_o, u_g)
    assert np.array_equal(o.segments(1), ctx.segments(1))
    d = np.abs(img_o.astype(int) - img_g.astype(int))
    assert d.max() <= 1, (d.max(), int((d > 0).sum()))


def test_sort_properties_large(ctx):
    """Size-independent properties at 10 M keys: non-decreasing keys, stable, permutation."""
    rng = np.random.default_rng(1)
    n = 10_000_000
    v = rng.integers(0, 2 ** 62, n, dtype=np.uint64)
    v &= np.uint64(~((0x7FF - 0x1FF) << 53) & 0xFFFFFFFFFFFFFFFF)   # keep a few digits constant -> digit skipping
    s = ctx.sort_array(v)
    k = s >> np.uint64(20)
    assert (k[1:] >= k[:-1]).all()
    ref = v[np.argsort(v >> np.uint64(20), kind="stable")]
    assert np.array_equal(s, ref)


def test_empty_and_degenerate(ctx):
    o = orc.Oracle()
    comp = S.Composition()
    t = comp.tables(o)
    S.load(o, t); S.load(ctx, t)
    a = o.render(33, 17, clear=(1, 0, 0, 1)); b = ctx.render(33, 17, clear=(1, 0, 0, 1))
    assert np.array_equal(a, b)
    # a single horizontal line (culled) and a shape completely outside the canvas
    comp.get_mut_or_insert_default(3).insert(S.P().move_to(0, 5).line_to(10, 5).build())
    comp.get_mut_or_insert_default(5).insert(S.custom_square(-50, -50, -20, -20)).set_props(S.solid((0, 1, 0, 1)))
    comp.get_mut_or_insert_default(6).insert(S.custom_square(-50, 3, 1000, 9)).set_props(S.solid((0, 0, 1, 0.5)))
    t = comp.tables(o)
    S.load(o, t); S.load(ctx, t)
    a = o.render(33, 17, clear=(1, 0, 0, 1)); b = ctx.render(33, 17, clear=(1, 0, 0, 1))
    assert np.array_equal(ctx.segments(1), o.segments(1))
    assert np.array_equal(a, b)


def test_deep_tiles_overflow_painter(ctx):
    """> 512 translucent layers over the same tiles: the tile's layer list does not fit the common-case painter and is
    painted by the deep variant (second launch)."""
    rng = np.random.default_rng(3)
    comp = S.Composition()
    for i in range(700):
        x0, y0 = rng.uniform(-10, 30, 2)
        comp.get_mut_or_insert_default(i).insert(S.custom_square(float(x0), float(y0), float(x0 + rng.uniform(20, 60)), float(y0 + rng.uniform(20, 60)))) \
            .set_props(S.solid((float(rng.random()), float(rng.random()), float(rng.random()), 0.05)))
    o, _ = both(ctx, comp)
    a = o.render(96, 80, clear=(1, 1, 1, 1)); b = ctx.render(96, 80, clear=(1, 1, 1, 1))
    assert np.array_equal(ctx.segments(1), o.segments(1))
    d = np.abs(a.astype(int) - b.astype(int))
    assert d.max() <= 1, (d.max(), int((d > 0).sum()))


def test_layers_inserted_out_of_order(ctx):
    """Geometry inserted in an order that is NOT the paint order: the rasterizer stream is not layer-sorted, so the radix
    sort must run its layer digits too (the layer-presorted shortcut must not fire)."""
    rng = np.random.default_rng(11)
    comp = S.Composition(insertion_order=True)
    orders = rng.permutation(200)
    for o_ in orders:
        x0, y0 = rng.uniform(0, 200, 2)
        comp.get_mut_or_insert_default(int(o_)).insert(S.custom_circle(float(x0), float(y0), float(rng.uniform(5, 40)))) \
            .set_props(S.solid((float(rng.random()), float(rng.random()), float(rng.random()), float(rng.choice([1.0, 0.6])))))
    o, _ = both(ctx, comp)
    a = o.render(256, 256, clear=(0, 0, 0, 1)); b, t = ctx.render(256, 256, clear=(0, 0, 0, 1), timings=True)
    assert np.array_equal(ctx.segments(0), o.segments(0))
    assert np.array_equal(ctx.segments(1), o.segments(1))
    assert t["n_sort_passes"] >= 3
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1


def test_repeated_frames_take_the_asynchronous_path(ctx, mixed):
    """From the second frame of a scene on, forma_hip_render enqueues the whole frame without reading N, J or the key
    masks back (they are predicted from the previous frame and verified afterwards).  Same results, every frame."""
    o, _ = both(ctx, mixed)
    want = o.render(512, 384, clear=(0.2, 0.3, 0.4, 1.0))
    for _ in range(4):
        got, t = ctx.render(512, 384, clear=(0.2, 0.3, 0.4, 1.0), timings=True)
        assert np.array_equal(ctx.segments(1), o.segments(1))
        assert t["n_segments"] == len(o.segments(0))
        assert np.abs(want.astype(int) - got.astype(int)).max() <= 1


def test_prediction_failure_falls_back(ctx):
    """Frame k+1 has far more pixel segments (and different live key bits) than frame k predicted: the asynchronous
    attempt must detect it (bounds / plan guard) and the frame must still come out right."""
    rng = np.random.default_rng(5)
    comp = S.Composition()
    for i in range(150):
        x0, y0 = rng.uniform(0, 300, 2)
        comp.get_mut_or_insert_default(i).insert(S.custom_circle(float(x0), float(y0), float(rng.uniform(5, 30)))) \
            .set_props(S.solid((float(rng.random()), float(rng.random()), float(rng.random()), 0.7)))
    o = orc.Oracle()
    for shift in (-5000.0, -5000.0, 0.0, 0.0, -120.0, 40.0):          # off-canvas twice, then on, then partly
        for order, layer in comp.layers.items():
            layer.set_transform([1.0, 0.0, 0.0, 1.0, shift, 0.0])
        t = comp.tables(o)
        S.load(o, t)
        ctx.set_geoms(t["geoms"])                                       # geometry and styles stay resident
        if shift == -5000.0:
            S.load(ctx, t)
            ctx.render(320, 320, clear=(1, 1, 1, 1))                    # first frame of the scene: synchronous
        want = o.render(320, 320, clear=(1, 1, 1, 1))
        got = ctx.render(320, 320, clear=(1, 1, 1, 1))
        assert np.array_equal(ctx.segments(1), o.segments(1)), shift
        assert np.abs(want.astype(int) - got.astype(int)).max() <= 1, shift


def test_tile_row_bands_stitch_to_the_full_frame(ctx, mixed):
    """Multi-GPU sharding, exercised on one device: forma_hip_set_band restricts a context to a band of tile rows (lines
    outside are culled, foreign segments neutralised); painting every band with its crop and stitching the rows must
    reproduce the un-sharded frame exactly.  Bands come from the same host logic bench.py uses."""
    from forma_amd import sharding
    W, H = 500, 301
    o, _ = both(ctx, mixed)
    full_o = o.render(W, H, clear=(0.2, 0.3, 0.4, 1.0))
    ctx.set_band(0, 0)
    full = ctx.render(W, H, clear=(0.2, 0.3, 0.4, 1.0))
    tiles_h = (H + 15) // 16
    hist = sharding.row_histogram(ctx.segments(0), tiles_h)
    n_full = len(ctx.segments(0))
    for world in (2, 3, 8):
        edges = sharding.band_edges(hist, world)
        out = np.zeros_like(full)
        n_band_total = 0
        for rank in range(world):
            ctx.set_band(edges[rank], edges[rank + 1])
            x0, x1, y0, y1 = sharding.band_crop(edges, rank, W, H)
            for _ in range(2):                                           # second frame: the asynchronous path
                img = ctx.render(W, H, clear=(0.2, 0.3, 0.4, 1.0), crop=(x0, x1, y0, y1))
            out[y0:y1] = img[y0:y1]
            s = ctx.segments(1)
            ty = (s >> np.uint64(53)).astype(np.int64) - 1
            own = (ty >= edges[rank]) & (ty < edges[rank + 1])
            assert (ty[~own] == -1).all()                                # foreign segments are parked in tile row -1
            n_band_total += int(own.sum())
        assert n_band_total == int(hist.sum())                           # every paintable segment has exactly one owner
        assert np.array_equal(out, full)
    ctx.set_band(0, 0)
    assert np.abs(full.astype(int) - full_o.astype(int)).max() <= 1
    assert n_full == len(o.segments(0))


@pytest.mark.parametrize("size", [(1, 1), (16, 16), (17, 15), (3840, 16), (16, 2160), (33, 1000)])
def test_odd_canvas_shapes(ctx, mixed, size):
    """Single-tile, single-row, single-column and ragged canvases (partial edge tiles, cpu/buffer/layout/mod.rs:264-295)."""
    w, h = size
    o, _ = both(ctx, mixed)
    a = o.render(w, h, clear=(0.5, 0.5, 0.5, 1.0)); b = ctx.render(w, h, clear=(0.5, 0.5, 0.5, 1.0))
    assert np.array_equal(ctx.segments(0), o.segments(0))
    assert np.array_equal(ctx.segments(1), o.segments(1))
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1


def test_layer_limit_and_far_away_geometry(ctx):
    """Order = LAYER_LIMIT (2^21 - 1, consts.rs:107-109) uses every layer bit of the key; geometry far left of the canvas
    collapses into the tile_x = -1 bucket (pixel_segment.rs:47-52) and still feeds the carry; geometry far above /
    below / right is culled or parked."""
    LIMIT = (1 << 21) - 1
    comp = S.Composition()
    comp.get_mut_or_insert_default(0).insert(S.custom_square(-3000.0, 10.0, 60.0, 50.0)).set_props(S.solid((1, 0, 0, 0.5)))        # 3000 px left
    comp.get_mut_or_insert_default(7).insert(S.custom_square(20.0, -5000.0, 90.0, 5000.0)).set_props(S.solid((0, 1, 0, 0.5)))      # tall
    comp.get_mut_or_insert_default(9).insert(S.custom_square(500.0, 500.0, 900.0, 900.0)).set_props(S.solid((0, 0, 1, 1)))         # off canvas
    comp.get_mut_or_insert_default(LIMIT).insert(S.custom_circle(64.0, 64.0, 50.0)).set_props(S.solid((0, 0, 1, 0.5)))
    o, _ = both(ctx, comp)
    a = o.render(128, 128, clear=(1, 1, 1, 1)); b = ctx.render(128, 128, clear=(1, 1, 1, 1))
    u = ctx.segments(0)
    assert np.array_equal(u, o.segments(0)) and np.array_equal(ctx.segments(1), o.segments(1))
    assert ((u >> np.uint64(20)) & np.uint64(0x1FFFFF)).max() == LIMIT
    assert (((u >> np.uint64(41)) & np.uint64(0xFFF)) == 0).any()          # the left-of-canvas bucket is populated
    assert np.array_equal(a, b)


def test_argument_validation(ctx):
    """The C ABI returns FORMA_E_ARG where the reference asserts / panics (consts.rs:25-26, layout/mod.rs:188-193)."""
    from forma_amd._lib import FormaError
    with pytest.raises(FormaError):
        ctx.render(65537, 16)
    with pytest.raises(FormaError):
        ctx.render(16, 32769)
    with pytest.raises(FormaError):
        ctx.render(64, 64, dst=np.zeros((64, 64 * 4), np.uint8), stride=63 * 4)
    with pytest.raises(FormaError):
        ctx.render(64, 64, channels=(0, 1, 2, 9))
    g = np.zeros(1, orc.GEOM_DTYPE); g[0]["order"] = 1 << 21
    with pytest.raises(FormaError):
        ctx.set_geoms(g)


def test_triangles_10m_8k(ctx):
    """BASELINE.json configs[3] at full size: ~10 M pixel segments on 8192 x 8192 — sorted stream bit-exact, image within
    1 code value, plus the size-independent properties (permutation, non-decreasing keys)."""
    comp = S.Composition()
    rng = np.random.default_rng(4)
    for i in range(19400):
        ox, oy = rng.uniform(0, 8192 - 256), rng.uniform(0, 8192 - 256)
        p = (rng.random((3, 2)) * 256 + np.array([ox, oy])).astype(np.float32)
        comp.get_mut_or_insert_default(i).insert(S.P().move_to(float(p[0, 0]), float(p[0, 1])).line_to(float(p[1, 0]), float(p[1, 1]))
                                                 .line_to(float(p[2, 0]), float(p[2, 1])).build()) \
            .set_props(S.solid((float(rng.random()), float(rng.random()), float(rng.random()), 1.0)))
    o, _ = both(ctx, comp)
    a = o.render(8192, 8192, clear=(1, 1, 1, 1))
    b, t = ctx.render(8192, 8192, clear=(1, 1, 1, 1), timings=True)
    s = ctx.segments(1)
    assert 9_000_000 < len(s) < 11_000_000 and t["n_segments"] == len(s)
    assert np.array_equal(np.sort(ctx.segments(0)), np.sort(s))            # permutation of the rasterizer's stream
    k = s >> np.uint64(20)
    assert (k[1:] >= k[:-1]).all()
    assert np.array_equal(s, o.segments(1))
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    assert d.max() <= 1


def _many_small_squares(n, width, row_y=4.0, seed=21, first_order=0, size=6.0):
    rng = np.random.default_rng(seed)
    comp = S.Composition()
    for i in range(n):
        x0 = float(rng.uniform(0, width - size))
        comp.get_mut_or_insert_default(first_order + i).insert(S.custom_square(x0, row_y, x0 + size, row_y + size)) \
            .set_props(S.solid((float(rng.random()), float(rng.random()), float(rng.random()), 0.5)))
    return comp


def test_tile_row_with_more_runs_than_the_in_lds_sort_holds(ctx):
    """The carry pre-pass orders a tile row's runs by (layer, tile_x) in LDS when the row has <= 16384 runs (the common
    case) and by a global radix sort otherwise.  18 000 small layers in ONE tile row take the second path — on the
    synchronous first frame and on the read-back-free frames after it."""
    comp = _many_small_squares(18000, 1024)
    o, _ = both(ctx, comp)
    want = o.render(1024, 32, clear=(1, 1, 1, 1))
    for _ in range(3):
        got = ctx.render(1024, 32, clear=(1, 1, 1, 1))
        assert np.array_equal(ctx.segments(1), o.segments(1))
        assert np.abs(want.astype(int) - got.astype(int)).max() <= 1


def test_row_run_count_jumps_between_frames(ctx):
    """Frame k fits the in-LDS run sort, frame k+1 (same geometry, layers moved into one row) does not: the asynchronous
    frame guessed 'fits', the device guard voids it and the synchronous re-run takes the global sort.  And back."""
    rng = np.random.default_rng(8)
    comp = S.Composition()
    n = 17500
    for i in range(n):
        x0, y0 = float(rng.uniform(0, 1000)), float(rng.uniform(0, 1000))
        comp.get_mut_or_insert_default(i).insert(S.custom_square(x0, y0, x0 + 5.0, y0 + 5.0)) \
            .set_props(S.solid((float(rng.random()), float(rng.random()), float(rng.random()), 0.6)))
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(o, t); S.load(ctx, t)
    base_geoms = t["geoms"].copy()
    for squash in (False, False, True, True, False, False):
        g = base_geoms.copy()
        if squash:                                   # y -> y / 128: every layer lands in the first tile rows
            g["flags"] = 1
            g["xf"] = np.array([1.0, 0.0, 0.0, 1.0 / 128.0, 0.0, 2.0], np.float32)   # [ux, vx, uy, vy, tx, ty]
        o.set_geoms(g); ctx.set_geoms(g)
        want = o.render(1024, 1024, clear=(0, 0, 0, 1))
        got = ctx.render(1024, 1024, clear=(0, 0, 0, 1))
        assert np.array_equal(ctx.segments(1), o.segments(1)), squash
        assert np.abs(want.astype(int) - got.astype(int)).max() <= 1, squash


def test_orders_above_16_bits_use_the_global_run_sort(ctx):
    """The in-LDS run sort packs `layer << 16 | index`: scenes whose orders do not fit 16 bits keep the global sort."""
    comp = _many_small_squares(300, 256, first_order=70000, size=20.0)
    o, _ = both(ctx, comp)
    want = o.render(256, 64, clear=(1, 1, 1, 1))
    for _ in range(2):
        got = ctx.render(256, 64, clear=(1, 1, 1, 1))
        assert np.array_equal(ctx.segments(1), o.segments(1))
        assert np.abs(want.astype(int) - got.astype(int)).max() <= 1


def test_received_stream_entry_points_on_one_gpu(ctx, mixed):
    """forma_hip_rasterize_frame / _segments_device / _reserve_segments / _sort_paint_frame — the entry points a host-driven
    exchange uses — with world = 1: rasterize only, take the unsorted stream as a torch view of the context's buffer, keep
    the painted rows, hand them back through reserve_segments and sort + paint them: same image as a plain render."""
    import torch
    o, _ = both(ctx, mixed)
    W, H = 512, 384
    tiles_h = (H + 15) // 16
    want = ctx.render(W, H, clear=(0.2, 0.3, 0.4, 1.0))
    full = ctx.segments(0)
    ctx.rasterize_frame(W, H)
    seg = ctx.unsorted_view()
    assert seg.is_cuda and seg.numel() == len(full)
    assert np.array_equal(seg.cpu().numpy().view(np.uint64), full)
    ty = ((seg >> 53) & 0x7FF) - 1
    kept = seg[(ty >= 0) & (ty < tiles_h)]
    recv = ctx.reserve_view(int(kept.numel()))
    recv.copy_(kept)
    torch.cuda.synchronize()
    tyh = (full >> np.uint64(53)).astype(np.int64) - 1
    assert np.array_equal(recv.cpu().numpy().view(np.uint64), full[(tyh >= 0) & (tyh < tiles_h)])
    got = ctx.sort_paint_frame(int(recv.numel()), W, H, clear=(0.2, 0.3, 0.4, 1.0), device_only=False)
    assert np.array_equal(got, want)
