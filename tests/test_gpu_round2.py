"""GPU parity, part 2: entry points and configurations that round 1 left untested (VERDICT r1 "What's weak" 2):
the 4-bit radix instantiation (the north star's literal digit width), `forma_hip_paint`, the remaining channel orders,
BASELINE config 1 at 256 x 256, config 5 (deterministic spaceship) at 3840 x 2160 with the buffer-layer cache, the
Flusher and a non-linear `Layout` through the product API.  Everything goes through the C ABI / product API and is
compared with the oracle on identical inputs."""
import os

import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import forma_amd
    c = forma_amd.Context(0)
    yield c
    c.close()


def both(ctx, comp):
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(o, t); S.load(ctx, t)
    return o, t


# ---- a8: the 4-bit digit pass ---------------------------------------------------------------------------------------------
def test_sort_4bit_digits_10m_keys(ctx):
    """forma_hip_sort(..., digit_bits = 4): k_onesweep<4>, 11 passes over the 44 key bits.  Same properties as the 8-bit
    test: keys non-decreasing, stable, a permutation — i.e. bit-identical to a stable sort by `v >> 20`."""
    rng = np.random.default_rng(11)
    n = 10_000_000
    v = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    s = ctx.sort_array(v, digit_bits=4)
    ref = v[np.argsort(v >> np.uint64(20), kind="stable")]
    assert np.array_equal(s, ref)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 16383, 16384, 16385, 100_001])
def test_sort_ragged_sizes_both_digit_widths(ctx, n):
    rng = np.random.default_rng(n + 5)
    v = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    v &= np.uint64(0xFFFFF00FFFFFFFFF)                      # a dead nibble inside the key: digit skipping with 4-bit digits
    ref = v[np.argsort(v >> np.uint64(20), kind="stable")]
    for bits in (4, 8, 9):
        assert np.array_equal(ctx.sort_array(v, digit_bits=bits), ref), (n, bits)


def test_sort_9bit_digits_10m_keys(ctx):
    """forma_hip_sort(..., digit_bits = 9): k_onesweep<9> — 512 bins, per-wave counters in 16-bit halves — which a frame picks by
    itself when nine-bit digits save a whole pass: stable, a permutation, keys non-decreasing."""
    rng = np.random.default_rng(19)
    n = 10_000_000
    v = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    ref = v[np.argsort(v >> np.uint64(20), kind="stable")]
    assert np.array_equal(ctx.sort_array(v, digit_bits=9), ref)


def test_value_range_digits_on_a_power_of_two_canvas(monkeypatch):
    """A canvas of 2^k tiles uses the values 1 .. 2^k of a key field that stores tile + 1: k + 1 live bits for ONE value, and at
    4096 x 4096 (256 tiles: 9 + 9 live bits) that costs a third digit pass.  Read-back-free frames sort by (field - its minimum
    on the previous frame) in two passes; k_sort_hist checks the keys against that span and voids a frame that leaves it (a
    shape that moves left of the canvas stores tile_x + 1 = 0): the re-run and every later frame use plain digits.  Stream and
    image equal the oracle's throughout.  (Eight-bit digits forced: with nine-bit ones 9 + 9 bits are two passes anyway; the
    8192 x 8192 canvas of BASELINE config 4 — 10 + 10 bits — is where the default plan gains, tests/test_gpu_parity.py.)"""
    W = H = 4096
    rng = np.random.default_rng(41)
    comp = S.Composition()
    order = 0
    for k in range(160):
        x, y = float(rng.uniform(0, W - 220)), float(rng.uniform(0, H - 220))
        comp.get_mut_or_insert_default(order).insert(S.custom_circle(x + 100, y + 100, float(rng.uniform(20, 100)))).set_props(
            S.solid(tuple(float(v) for v in rng.random(3)) + (1.0,)))
        order += 1
    # shapes in the last tile column and row (tile 255: field value 256 = bit 8) and in the first ones
    for (x0, y0, x1, y1) in ((4060, 10, 4095.5, 300), (10, 4060, 400, 4095.5), (4000, 4000, 4095.9, 4095.9), (0.5, 0.5, 40, 40)):
        comp.get_mut_or_insert_default(order).insert(S.custom_square(x0, y0, x1, y1)).set_props(S.solid((0.2, 0.4, 0.6, 1.0)))
        order += 1
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", "digit_bits=8")              # (with nine-bit digits this canvas needs no help: 9 + 9 bits; 8192 x 8192 does)
    ctx = forma_amd.Context(0)
    o, t = both(ctx, comp)
    want = o.render(W, H)
    passes = []
    for frame in range(5):
        img, tm = ctx.render(W, H, timings=True)
        passes.append(int(tm["n_sort_passes"]))
        assert np.array_equal(ctx.segments(1), o.segments(1)), frame
        assert np.abs(want.astype(int) - img.astype(int)).max() <= 1, frame
    assert passes[0] == 3 and passes[-1] == 2, passes                  # plain digits on the synchronous frame, then the span is known
    # the last layer moves left of the canvas: its segments store tile_x + 1 = 0, below the span the digits were planned for
    g = t["geoms"].copy()
    last = g["order"] == order - 1
    g["flags"][last] = 1
    g["xf"][last] = np.array([1.0, 0.0, 0.0, 1.0, -30.0, 0.0], np.float32)
    o.set_geoms(g); ctx.set_geoms(g)
    want = o.render(W, H)
    for frame in range(4):
        img, tm = ctx.render(W, H, timings=True)
        assert np.array_equal(ctx.segments(1), o.segments(1)), ("moved", frame)
        assert np.abs(want.astype(int) - img.astype(int)).max() <= 1, ("moved", frame)
        assert int(tm["n_sort_passes"]) == 3, (frame, tm["n_sort_passes"])
    ctx.close()


@pytest.mark.parametrize("switch", ["", "no_ras_hist", "no_prezero", "digit_bits=4"])
@pytest.mark.parametrize("paint_order", [True, False])
def test_sort_histograms_taken_by_the_rasterizer(monkeypatch, switch, paint_order):
    """On read-back-free frames k_rasterize counts the digits of the speculated sort plan while it makes the keys (up to three
    passes: the sort then runs without k_sort_hist's read of the whole stream), leaves the tile-field spans in its mask records
    and voids the frame if a key leaves the span a biased digit was planned for.  Frames with it, without it (no_ras_hist; no_prezero:
    nobody cleared the histograms ahead of the rasterizer; four-bit digits and layers inserted out of paint order: more passes
    than the rasterizer counts) — sorted stream and image equal the oracle's on every frame, also after the geometry moves."""
    W, H = 1500, 1100
    rng = np.random.default_rng(77)
    comp = S.Composition()
    orders = list(range(90))
    if not paint_order:
        rng.shuffle(orders)                                             # the stream is not non-decreasing in layer: layer digits too
    for order in orders:
        x, y = float(rng.uniform(-40, W - 100)), float(rng.uniform(-40, H - 100))
        comp.get_mut_or_insert_default(int(order)).insert(S.custom_circle(x + 60, y + 60, float(rng.uniform(10, 140)))).set_props(
            S.solid(tuple(float(v) for v in rng.random(3)) + (float(rng.uniform(0.3, 1.0)),)))
    import forma_amd
    if switch:
        monkeypatch.setenv("FORMA_HIP_DEBUG", switch)
    ctx = forma_amd.Context(0)
    o, t = both(ctx, comp)
    want = o.render(W, H)
    for frame in range(4):
        img = ctx.render(W, H)
        assert np.array_equal(ctx.segments(1), o.segments(1)), frame
        assert np.abs(want.astype(int) - img.astype(int)).max() <= 1, frame
    g = t["geoms"].copy()
    g["flags"][:] = 1
    g["xf"][:] = np.array([0.9, 0.0, 0.0, 1.1, 35.0, -20.0], np.float32)      # every shape moves: other digits, other spans
    o.set_geoms(g); ctx.set_geoms(g)
    want = o.render(W, H)
    for frame in range(3):
        img = ctx.render(W, H)
        assert np.array_equal(ctx.segments(1), o.segments(1)), ("moved", frame)
        assert np.abs(want.astype(int) - img.astype(int)).max() <= 1, ("moved", frame)
    ctx.close()


@pytest.mark.parametrize("bits", [0, 4, 8, 9])
def test_sort_coherent_streams(ctx, bits):
    """what a rasterizer emits: long runs of equal tile digits, the same digit coming back within one wave row of 64 keys
    (A A B B A A: an outline that leaves a tile and returns), rows of one digit, digits that alternate key by key — the
    ranking takes its run-structured path where a row allows and the general one where it does not; both are stable"""
    rng = np.random.default_rng(23 + bits)
    n = 3_000_000
    tiles = rng.integers(0, 1 << 23, 4000, dtype=np.uint64)             # (tile_y, tile_x) values in play
    runs = rng.integers(1, 48, n // 8)
    pick = rng.integers(0, 6, runs.size)                                # a few tiles alternate locally: repeats inside a row
    base = np.repeat(np.arange(runs.size) // 50, 1)
    tile_of_run = tiles[(base * 6 + pick) % tiles.size]
    t = np.repeat(tile_of_run, runs)[:n]
    t[1000:1200:2] = tiles[0]; t[1001:1200:2] = tiles[1]               # strict alternation
    t[5000:9000] = tiles[2]                                             # whole tiles of the sort with one digit
    low = rng.integers(0, 1 << 41, t.size, dtype=np.uint64)
    v = (t << np.uint64(41)) | low
    ref = v[np.argsort(v >> np.uint64(20), kind="stable")]
    assert np.array_equal(ctx.sort_array(v, digit_bits=bits), ref)
    # layer-sorted form: only the tile bits vary between neighbours (what a frame sorts: two or three digit passes)
    v2 = (t << np.uint64(41)) | (np.arange(t.size, dtype=np.uint64) & np.uint64(0xFFFFF))
    ref2 = v2[np.argsort(v2 >> np.uint64(20), kind="stable")]
    assert np.array_equal(ctx.sort_array(v2, digit_bits=bits), ref2)


def test_whole_frames_with_4bit_digits():
    """One context created under FORMA_HIP_DEBUG=digit_bits=4 renders whole frames (synchronous first frame, read-back-free
    second frame, out-of-order layers = full-key sort): streams bit-identical, image identical."""
    import forma_amd
    os.environ["FORMA_HIP_DEBUG"] = "digit_bits=4"
    try:
        c4 = forma_amd.Context(0)
    finally:
        del os.environ["FORMA_HIP_DEBUG"]
    try:
        for comp, (w, h) in ((S.random_mixed(), (512, 384)), (S.random_cubics(300, 1920, 1080), (1920, 1080))):
            o, _ = both(c4, comp)
            want = o.render(w, h, clear=(0.1, 0.2, 0.3, 1.0))
            for frame in range(2):
                got, tm = c4.render(w, h, clear=(0.1, 0.2, 0.3, 1.0), timings=True)
                assert np.array_equal(c4.segments(0), o.segments(0)) and np.array_equal(c4.segments(1), o.segments(1))
                assert np.abs(want.astype(int) - got.astype(int)).max() <= 1
            assert tm["n_sort_passes"] >= 3                 # 4-bit digits: more passes than the 8-bit plan's two
        comp = S.Composition(insertion_order=True)          # geometry pushed out of paint order: the layer digits are sorted too
        for order in (5, 1, 9, 3, 70000):
            comp.get_mut_or_insert_default(order).insert(S.custom_circle(100 + order % 7 * 20, 90, 60)).set_props(S.solid((order % 3 / 2, 0.5, 0.2, 0.8)))
        o, _ = both(c4, comp)
        want = o.render(320, 200)
        got = c4.render(320, 200)
        assert np.array_equal(c4.segments(1), o.segments(1)) and np.abs(want.astype(int) - got.astype(int)).max() <= 1
    finally:
        c4.close()


# ---- forma_hip_paint --------------------------------------------------------------------------------------------------------
def seg(layer, tile_x, tile_y=0, lx=0, ly=0, dam=0, cover=0):
    return orc.pixel_segment(layer, tile_x, tile_y, lx, ly, dam, cover)


def styles(props_by_layer):
    n = max(props_by_layer) + 1
    offsets = np.full(n, S.NONE, np.uint32); words = []
    for lid, p in sorted(props_by_layer.items()):
        offsets[lid] = len(words); words += S.encode_props(p, [])
    return offsets, np.asarray(words, np.uint32)


def test_paint_entry_point_on_reference_vectors(ctx):
    """forma_hip_paint on the hand-built sorted streams of the reference's painter tests `skip_opaque_tiles`
    (painter/mod.rs:1606-1715: left-of-canvas carry, opaque cover hides lower layers) and `crop` (:1717-1781)."""
    T = 16
    segments = [seg(2, -1, 0, T - 1, y, 0, 16) for y in range(T)] + [seg(0, -1, 0, T - 1, 0, 0, 16), seg(1, 0, 0, 0, 1, 0, 16)]
    segments += [seg(2, 1, 0, T - 1, y, 0, -16) for y in range(T)]
    segments.sort()
    off, words = styles({0: S.solid((0, 0, 1, 1)), 1: S.solid((0, 1, 0, 1)), 2: S.solid((1, 0, 0, 1))})
    o = orc.Oracle(); o.set_styles(off, words); ctx.set_styles(off, words)
    want = o.paint(segments, 3 * T, T, clear=(0, 0, 0, 1))
    got = ctx.paint(segments, 3 * T, T, clear=(0, 0, 0, 1))
    assert np.array_equal(want, got)
    px = got.reshape(T, 3 * T, 4)
    assert (px[:, :2 * T] == [255, 0, 0, 255]).all() and (px[0, 2 * T:] == [0, 0, 255, 255]).all() and (px[1, 2 * T:] == [0, 255, 0, 255]).all()
    assert (px[2:, 2 * T:] == [0, 0, 0, 255]).all()
    # crop: only the tiles of the (tile-rounded) rectangle are written into the caller's buffer
    segments = sorted(seg(0, 0, j, T - 1, y, 0, 16) for j in range(3) for y in range(T))
    off, words = styles({0: S.solid((0, 0, 1, 1))})
    o.set_styles(off, words); ctx.set_styles(off, words)
    crop = (T, 2 * T + T // 2, T, 2 * T)
    want = o.paint(segments, 3 * T, 3 * T, clear=(1, 0, 0, 1), crop=crop)
    got = ctx.paint(segments, 3 * T, 3 * T, clear=(1, 0, 0, 1), crop=crop, dst=np.zeros((3 * T, 3 * T * 4), np.uint8))
    assert np.array_equal(want, got)
    px = got.reshape(3 * T, 3 * T, 4)
    assert not px[:T].any() and not px[2 * T:].any() and not px[T:2 * T, :T].any() and (px[T:2 * T, T:] == [0, 0, 255, 255]).all()


def test_paint_entry_point_on_rendered_streams(ctx):
    """forma_hip_paint(sorted stream of the oracle) == oracle.paint(same stream), with odd strides, all fills / blends /
    clips (random_mixed) and an empty stream."""
    comp = S.random_mixed()
    o, _ = both(ctx, comp)
    w, h = 500, 301
    o.prepare_lines(w, h); o.rasterize(); srt = o.sort()
    stride = w * 4 + 12
    want = o.paint(srt, w, h, clear=(0.3, 0.3, 0.3, 1.0), stride=stride, dst=np.full((h, stride), 9, np.uint8))
    got = ctx.paint(srt, w, h, clear=(0.3, 0.3, 0.3, 1.0), stride=stride, dst=np.full((h, stride), 9, np.uint8))
    assert np.abs(want.astype(int) - got.astype(int)).max() <= 1
    assert (got[:, w * 4:] == 9).all()                      # bytes between the row and the stride are never touched
    want = o.paint(np.zeros(0, np.uint64), 40, 24, clear=(0.5, 0.25, 1.0, 0.5))
    got = ctx.paint(np.zeros(0, np.uint64), 40, 24, clear=(0.5, 0.25, 1.0, 0.5))
    assert np.array_equal(want, got)


# ---- a16 / f4: every channel order of cpu/channel.rs:57-62 ---------------------------------------------------------------------
@pytest.mark.parametrize("channels", [S.RGBA, S.BGRA, S.RGB0, S.BGR0, S.RGB1, S.BGR1], ids=["RGBA", "BGRA", "RGB0", "BGR0", "RGB1", "BGR1"])
@pytest.mark.parametrize("clear", [(1.0, 1.0, 1.0, 0.0), (0.2, 0.4, 0.6, 1.0)], ids=["clear_alpha0", "clear_opaque"])
def test_all_channel_orders(ctx, channels, clear):
    o, _ = both(ctx, S.random_mixed(n=120, width=256, height=160, seed=3))
    a = o.render(256, 160, channels=channels, clear=clear)
    b = ctx.render(256, 160, channels=channels, clear=clear)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
    assert np.array_equal(a, b)


# ---- BASELINE configs[0]: the e2e shapes at 256 x 256 -------------------------------------------------------------------------
def test_config1_e2e_scenes_at_256(ctx, monkeypatch):
    """the e2e scene builders with WIDTH = HEIGHT = 256, PADDING = 32 (16 x 16 tiles), every stage."""
    monkeypatch.setattr(S, "WIDTH", 256.0); monkeypatch.setattr(S, "HEIGHT", 256.0); monkeypatch.setattr(S, "PADDING", 32.0)
    scenes = S.e2e_scenes()
    for name in ("linear_gradient", "radial_gradient", "solid_color__blue", "blend_modes__Hue", "blend_modes__SoftLight",
                 "fill_rules__EvenOdd", "clipping", "clipping2", "texture"):
        o, _ = both(ctx, scenes[name])
        want = o.render(256, 256)
        got = ctx.render(256, 256)
        assert np.array_equal(ctx.segments(0), o.segments(0)), name
        assert np.array_equal(ctx.segments(1), o.segments(1)), name
        assert np.array_equal(want, got), name


# ---- BASELINE configs[4]: deterministic spaceship at 4K with the per-tile damage optimizer --------------------------------------
def test_config5_spaceship_4k_with_cache_matches_oracle():
    """`forma_amd.spaceship.Spaceship` (the reference demo's game logic) at 3840 x 2160, BGR1, clear
    (1, 1, 1, 0), one persistent BufferLayerCache (demo/src/runner.rs:150-165).  The product and the oracle-backed mirror
    of the API run the same game for 4 s of game time (240 frames); after EVERY frame the two carried buffers must hold
    the same bytes, and on sampled frames the same set of tiles must have been rewritten."""
    import ref_api
    from forma_amd import api
    from forma_amd.spaceship import Spaceship
    W, H = 3840, 2160
    sides = []
    for a in (ref_api, api):
        comp, r = a.Composition(), a.Renderer()
        sides.append(dict(api=a, comp=comp, r=r, cache=r.create_buffer_layer_cache(), game=Spaceship(a, W, H),
                          buf=np.zeros(W * H * 4, np.uint8), lay=a.LinearLayout(W, W * 4, H)))
    partial = 0
    for f in range(240):
        sample = f < 8 or f % 8 == 0
        touched = []
        for s in sides:
            s["game"].compose(s["comp"])
            before = s["buf"].copy() if sample else None
            s["r"].render(s["comp"], s["api"].BufferBuilder(s["buf"], s["lay"]).layer_cache(s["cache"]).build(), s["api"].BGR1,
                          s["api"].Color(1, 1, 1, 0), None)
            if sample:
                touched.append((s["buf"] != before).reshape(H // 16, 16, W // 16, 16, 4).any(axis=(1, 3, 4)))
        assert len(sides[0]["game"].actors) == len(sides[1]["game"].actors)
        assert np.array_equal(sides[0]["buf"], sides[1]["buf"]), f"frame {f}"
        if sample:
            assert np.array_equal(touched[0], touched[1]), f"frame {f}: different tiles rewritten"
            partial += int(touched[1].any() and not touched[1].all())
    assert partial >= 10                                    # partial damage (a few tiles per frame) was actually exercised
    assert len(sides[1]["game"].actors) >= 5


# ---- a17 / f4: Flusher and a generic Layout through the product API -------------------------------------------------------------
def test_generic_layout_and_flusher():
    from forma_amd import api

    class Tiled(api.Layout):                                # cpu/buffer/layout/mod.rs:51-163: a user-defined Layout
        def __init__(self, w, h):
            self._w, self._h = w, h

        def width(self): return self._w
        def height(self): return self._h
        def slices_per_tile(self): return 16

        def slices(self, buffer):
            tiles = buffer.reshape(self.height_in_tiles() * self.width_in_tiles(), 16, 64)
            return [tiles[t, y] for t in range(len(tiles)) for y in range(16)]

        @staticmethod
        def write(slices, flusher, fill):
            kind, payload = fill
            for y, row in enumerate(slices):
                px = row.reshape(16, 4)
                px[:] = payload if kind == "solid" else np.asarray(payload).reshape(16, 16, 4)[:, y]
            if flusher is not None:
                for row in slices:
                    flusher.flush(row)

    class Counting:
        def __init__(self): self.n = 0
        def flush(self, s): self.n += 1; assert len(s) == 64

    W, H = 80, 48                                           # 5 x 3 tiles
    comp = api.Composition()
    P = api.Point
    tri = api.PathBuilder().move_to(P(4, 4)).line_to(P(70, 10)).line_to(P(30, 44)).build()
    gb = api.GradientBuilder(P(0, 0), P(80, 48)); gb.color(api.Color(1, 0, 0, 1)).color(api.Color(0, 0, 1, 1))
    comp.get_mut_or_insert_default(api.Order(0)).insert(tri).set_props(api.Props(func=api.Func.Draw(api.Style(fill=api.Fill.Gradient(gb.build())))))
    r = api.Renderer(0)
    linear = np.zeros(W * H * 4, np.uint8)
    r.render(comp, api.BufferBuilder(linear, api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    tiled = np.zeros(W * H * 4, np.uint8)
    fl = Counting()
    r.render(comp, api.BufferBuilder(tiled, Tiled(W, H)).flusher(fl).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    want = linear.reshape(3, 16, 5, 16, 4).transpose(0, 2, 1, 3, 4).reshape(-1)
    assert np.array_equal(tiled, want)
    assert fl.n == 15 * 16
    # with a cache, a second identical frame writes nothing through the layout
    cache = r.create_buffer_layer_cache()
    r.render(comp, api.BufferBuilder(tiled, Tiled(W, H)).layer_cache(cache).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    tiled[:] = 7; fl = Counting()
    r.render(comp, api.BufferBuilder(tiled, Tiled(W, H)).layer_cache(cache).flusher(fl).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    assert (tiled == 7).all() and fl.n == 0


def test_tiles_written_reports_crop_and_damage(ctx):
    o, t = both(ctx, S.random_mixed(n=60, width=160, height=96, seed=8))
    ctx.render(160, 96, crop=(20, 100, 10, 40))
    f = ctx.tiles_written(160, 96).reshape(6, 10)
    want = np.zeros((6, 10), np.uint8); want[0:3, 1:7] = 1
    assert np.array_equal(f, want)
    ctx.cache_clear(7)
    buf = np.zeros((96, 640), np.uint8)
    ctx.render(160, 96, cache_id=7, dst=buf)
    assert ctx.tiles_written(160, 96).all()
    un = np.ones_like(t["unchanged"])
    ctx.set_styles(t["style_offsets"], t["style_words"], un)
    ctx.render(160, 96, cache_id=7, dst=buf)
    assert not ctx.tiles_written(160, 96).any()
    ctx.render(160, 96, cache_id=7, device_only=True)        # device-resident frame: flags fetched on demand
    assert not ctx.tiles_written(160, 96).any()


def test_bad_texture_index_and_style_offsets_are_rejected(ctx):
    """a style offset near 2^32 must not wrap; a texture style naming an image that was never uploaded is an
    argument error at render time (set_styles / set_images are separate calls)."""
    from forma_amd import FormaError
    with pytest.raises(FormaError):
        ctx.set_styles(np.asarray([0xFFFFFFFE], np.uint32), np.zeros(8, np.uint32))
    img = S.Image.from_srgba([[255, 0, 0, 255]] * 4, 2, 2)
    comp = S.Composition()
    comp.get_mut_or_insert_default(0).insert(S.square()).set_props(S.Props(fill=S.Texture((1, 0, 0, 1, 0, 0), img)))
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(ctx, t)
    ctx.set_images(np.zeros(0, orc.IMAGE_DTYPE), np.zeros((0, 4), np.uint16))
    with pytest.raises(FormaError):
        ctx.render(64, 64)
    S.load(ctx, t)
    S.load(o, t)
    assert np.array_equal(ctx.render(64, 64), o.render(64, 64))


# ---- frames into caller memory ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("crop", [None, (37, 500, 130, 1333)])
def test_frames_into_caller_memory_strided_and_registered(crop):
    """`dst` is complete when render returns (cpu/buffer/mod.rs:43-49), pixels outside the crop are untouched; a strided
    destination and a registered (page-locked) one receive the same bytes"""
    import forma_amd
    W, H = 560, 1400                                                   # 88 tile rows, the last one partial
    o = orc.Oracle()
    t = S.random_mixed(n=500, width=W, height=H, seed=61).tables(o)
    S.load(o, t)
    c = forma_amd.Context(0)
    S.load(c, t)
    for frame in range(4):                                             # synchronous first, then read-back-free frames
        clear = (0.1 * frame, 0.3, 0.5, 1.0)
        want = o.render(W, H, clear=clear, crop=crop, dst=np.full((H, W * 4), 77, np.uint8))
        got = c.render(W, H, clear=clear, crop=crop, dst=np.full((H, W * 4), 77, np.uint8))
        assert np.abs(want.astype(int) - got.astype(int)).max() <= 1, frame
    assert np.array_equal(c.segments(1), o.segments(1))
    wide = np.full((H, W * 4 + 64), 5, np.uint8)
    c.register_buffer(wide)
    c.render(W, H, dst=wide, stride=W * 4 + 64)
    assert np.abs(wide[:, :W * 4].astype(int) - o.render(W, H).astype(int)).max() <= 1 and (wide[:, W * 4:] == 5).all()
    c.unregister_buffer(wide)
    c.close()


def test_render_enqueue_and_tiles_deeper_than_the_painters_lists():
    """a deferred frame's image leaves behind its kernels, before the frame is verified; tiles that only k_paint_huge can paint
    (lists in global memory, sized by the host at settle time) are painted after that: the crop is copied again and the caller
    sees the finished image"""
    import forma_amd
    W, H = 64, 256
    comp = S.Composition()
    for i in range(4300):                                              # beyond k_paint_deep's 4096-entry lists
        comp.get_mut_or_insert_default(i).insert(S.custom_square(8, 100, 40, 130)).set_props(S.solid((0.5, 0.4, 0.3, 0.01)))
    for i in range(4300, 4310):
        comp.get_mut_or_insert_default(i).insert(S.custom_circle(32, 20 + 22 * (i - 4300), 18)).set_props(S.solid((0.1, 0.6, 0.9, 0.7)))
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(o, t)
    want = o.render(W, H)
    c = forma_amd.Context(0, frames_in_flight=2)
    S.load(c, t)
    bufs = [np.zeros((H, W * 4), np.uint8) for _ in range(3)]
    for k in range(7):
        c.render_enqueue(W, H, bufs[k % 3])
    c.sync()
    for b in bufs:
        assert np.abs(want.astype(int) - b.astype(int)).max() <= 1
    c.close()


def test_render_enqueue_into_registered_buffers():
    """forma_hip_render_enqueue: frame AND copy are enqueued on the next frame slot; a buffer is complete once `frames in flight`
    further frames have been enqueued, or after sync — three buffers in turn, scene changes in between"""
    import forma_amd
    W, H = 640, 1100
    o = orc.Oracle()
    t = S.random_mixed(n=300, width=W, height=H, seed=62).tables(o)
    S.load(o, t)
    c = forma_amd.Context(0, frames_in_flight=2)
    S.load(c, t)
    bufs = [np.zeros((H, W * 4), np.uint8) for _ in range(3)]
    for b in bufs:
        c.register_buffer(b)
    clears = [(1, 1, 1, 1), (0.2, 0.3, 0.4, 1.0), (0, 0, 0, 0), (0.5, 0.1, 0.9, 0.5)]
    wants = [o.render(W, H, clear=cl) for cl in clears]
    shown = 0
    for k in range(14):
        c.render_enqueue(W, H, bufs[k % 3], clear=clears[k % 4])
        if k >= 2:                                                     # two frames were enqueued behind frame k - 2: its buffer is done
            j = k - 2
            assert np.abs(bufs[j % 3].astype(int) - wants[j % 4].astype(int)).max() <= 1, j
            shown += 1
    c.sync()
    for j in (12, 13):
        assert np.abs(bufs[j % 3].astype(int) - wants[j % 4].astype(int)).max() <= 1, j
    assert shown == 12
    # the synchronous call still is: complete on return, and it sees everything enqueued before it
    got = c.render(W, H, clear=clears[1], dst=np.zeros((H, W * 4), np.uint8))
    assert np.abs(got.astype(int) - wants[1].astype(int)).max() <= 1
    for b in bufs:
        c.unregister_buffer(b)
    c.close()



def test_kernel_times_of_a_timed_frame(ctx):
    """forma_hip_kernel_times: every kernel of a timed frame, with its own launch-event duration.  The stage times of
    forma_timings_t are sums over the list; the frame's total (first start -> last end) is at least the sum."""
    both(ctx, S.random_mixed(n=200, width=1024, height=512, seed=5))
    for _ in range(3):                                       # third frame: read-back-free path
        _, tm = ctx.render(1024, 512, clear=(1, 1, 1, 1), device_only=True, timings=True)
    ks = ctx.kernel_times()
    names = [k[0] for k in ks]
    assert any(n.startswith("k_rasterize") for n in names) and any(n.startswith("k_paint_wave") for n in names), names
    assert all(us > 0.0 for _, _, _, us in ks)
    for stage, key in ((0, "prepare_us"), (1, "rasterize_us"), (2, "sort_us"), (3, "carry_us"), (4, "paint_us")):
        assert abs(sum(us for _, st, _, us in ks if st == stage) - tm[key]) < 0.05 + 1e-3 * tm[key], key
    assert tm["total_us"] + 1.0 >= sum(us for _, st, _, us in ks if st != 5)
    starts = [t0 for _, _, t0, _ in ks]
    assert starts == sorted(starts)


@pytest.mark.parametrize("switch", ["", "force_cull", "no_cull", "force_cull,strip_tiles=100000000", "no_cull,strip_tiles=0"])
def test_occlusion_culling_and_strip_painters_change_no_pixel(monkeypatch, switch):
    """The painters drop the entries below a tile's topmost occluder while they build its list (PaintParams::cull), and small
    frames are painted by four strip wavefronts per tile (k_paint_wave<.., NPX = 1>): neither may change a pixel.  Opaque cubics
    over the whole canvas (hundreds of hidden layers per tile: the case culling exists for), a mixed scene with gradients, blend
    modes and clips (culling must switch itself off), each against the oracle under every switch."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", switch)
    c = forma_amd.Context(0)
    try:
        for comp, (w, h) in ((S.random_cubics(n=400, width=640, height=360, seed=3), (640, 360)),
                             (S.random_cubics(n=300, width=512, height=256, seed=4, alpha=0.6), (512, 256)),
                             (S.random_mixed(n=300, width=512, height=384, seed=9), (512, 384))):
            o, _ = both(c, comp)
            for _ in range(3):                               # synchronous, then read-back-free frames
                img = c.render(w, h, clear=(1.0, 1.0, 1.0, 1.0))
            ref = o.render(w, h, clear=(1.0, 1.0, 1.0, 1.0))
            assert np.abs(img.astype(np.int16) - ref.astype(np.int16)).max() <= 1
    finally:
        c.close()


def test_painter_order_survives_crops_canvas_changes_and_voided_frames(monkeypatch):
    """The painters start with the previous frame's heavy tiles (PaintParams::order_*): a schedule only.  Many frames of one
    scene with a low threshold (every tile "heavy": the heavy section and the flags are exercised to their capacity), the
    crop and the canvas changing in between, a frame voided by new geometry — every image equals the oracle's."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", "strip_tiles=0,paint_quad=0,order_thr=1")   # (one wavefront per tile, every tile "heavy")
    c = forma_amd.Context(0)
    try:
        comp = S.random_mixed(n=400, width=1024, height=768, seed=21)
        o, _ = both(c, comp)
        for (w, h), crop in (((1024, 768), None), ((1024, 768), (100, 900, 64, 700)), ((1024, 768), (100, 900, 64, 700)), ((800, 600), None),
                             ((1024, 768), None)):
            ref = o.render(w, h, clear=(0.9, 0.9, 0.9, 1.0), crop=crop)
            for k in range(7):                           # (the threshold is steered frame by frame: the lists change under the same scene)
                img = c.render(w, h, clear=(0.9, 0.9, 0.9, 1.0), crop=crop)
                y0, y1, x0, x1 = (crop[2], crop[3], crop[0], crop[1]) if crop else (0, h, 0, w)
                a = img.reshape(h, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16); b = ref.reshape(h, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16)
                assert np.abs(a - b).max() <= 1, (w, h, crop, k)
        o2, _ = both(c, S.random_cubics(n=300, width=1024, height=768, seed=22, alpha=0.7))     # new geometry: predictions void
        ref = o2.render(1024, 768, clear=(1.0, 1.0, 1.0, 1.0))
        for k in range(5):
            img = c.render(1024, 768, clear=(1.0, 1.0, 1.0, 1.0))
            assert np.abs(img.astype(np.int16) - ref.astype(np.int16)).max() <= 1, k
    finally:
        c.close()


@pytest.mark.parametrize("size", [(1000, 500), (1936, 1088), (72, 40)])
def test_quad_painter_on_canvases_that_are_no_multiple_of_four_tiles(monkeypatch, size):
    """k_paint_quad (four tiles per wavefront, all-solid scenes) forced on canvases whose tile rows end inside a quad, with a
    crop that starts and ends inside quads, translucent and opaque layers (culling on: opaque cubics hide most entries)."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", "paint_quad=2")
    w, h = size
    c = forma_amd.Context(0)
    try:
        for comp in (S.random_cubics(n=250, width=w, height=h, seed=31), S.random_cubics(n=120, width=w, height=h, seed=32, alpha=0.5)):
            o, _ = both(c, comp)
            for crop in (None, (16 * 3 + 5, max(w - 40, 16 * 3 + 6), 17, max(h - 9, 18))):
                ref = o.render(w, h, clear=(1.0, 1.0, 1.0, 0.0), crop=crop)
                for _ in range(3):
                    img = c.render(w, h, clear=(1.0, 1.0, 1.0, 0.0), crop=crop)
                y0, y1, x0, x1 = (crop[2], crop[3], crop[0], crop[1]) if crop else (0, h, 0, w)
                a = img.reshape(h, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16); b = ref.reshape(h, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16)
                assert np.abs(a - b).max() <= 1, (size, crop)
    finally:
        c.close()


@pytest.mark.parametrize("switch", ["runs_chain=1", "runs_chain=0", "runs_chain=1,no_prezero", "runs_chain=1,carry_slices=3"])
def test_runs_numbered_per_tile_row_without_a_counting_pass(monkeypatch, switch):
    """launch_runs' chain (k_runs_wave<true>): read-back-free frames number their runs per tile row from the index of the row's
    first segment, every workgroup taking its predecessors' head counts from their status words — no k_runs_count.  Same
    images as the oracle's on canvases whose 2 048-segment tiles hold many rows (a 1080 x 40 strip), one row only (4096 x 16,
    a row of more than 64 tiles: the look-back probes twice), the usual mix with a crop; with every per-frame buffer poisoned
    (what lies between two rows' records is never a run), and after the scene SHRANK (last frame's records of the same tile
    sit right behind this frame's)."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", switch + ",poison_frame=255")
    c = forma_amd.Context(0)
    try:
        cases = (((1024, 768), S.random_mixed(n=400, width=1024, height=768, seed=41), S.random_mixed(n=150, width=1024, height=768, seed=42)),
                 ((1080, 40), S.random_cubics(n=300, width=1080, height=40, seed=43, alpha=0.6), S.random_cubics(n=40, width=1080, height=40, seed=44)),
                 ((4096, 16), S.random_cubics(n=1500, width=4096, height=16, seed=45, alpha=0.5), S.random_cubics(n=900, width=4096, height=16, seed=46)))
        for (w, h), big, small in cases:
            for comp in (big, small):                    # (the second scene of a canvas has fewer runs in every row)
                o, _ = both(c, comp)
                for crop in (None, (16 * 2 + 3, w - 21, 0, h) if h <= 48 else (16 * 2 + 3, w - 21, 33, h - 50)):
                    ref = o.render(w, h, clear=(1.0, 1.0, 1.0, 1.0), crop=crop)
                    for k in range(4):                   # (the first frame of a geometry is synchronous: the chain starts with the second)
                        img = c.render(w, h, clear=(1.0, 1.0, 1.0, 1.0), crop=crop)
                        y0, y1, x0, x1 = (crop[2], crop[3], crop[0], crop[1]) if crop else (0, h, 0, w)
                        a = img.reshape(h, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16); b = ref.reshape(h, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16)
                        assert np.abs(a - b).max() <= 1, (switch, (w, h), crop, k)
    finally:
        c.close()


@pytest.mark.parametrize("switch", ["runs_blk=1,runs_chain=0", "runs_blk=1,runs_chain=0,carry_slices=1", "runs_blk=1,runs_chain=0,blk_round=4",
                                    "runs_blk=1,runs_chain=0,carry_slices=1,blk_round=1", "runs_blk=1,runs_chain=0,no_prezero,blk_round=2",
                                    "runs_blk=0,runs_chain=0"])
def test_runs_numbered_per_tile_of_the_run_kernel(monkeypatch, switch):
    """launch_runs' BLOCKS numbering (k_runs_wave<2>): read-back-free frames whose tile rows take one COVL carry workgroup each
    number their runs per 2 048-segment tile of the run kernel — no counting pass, no look-back — into sparse arrays, and
    k_carry_rows finds a row's runs through the tiles' head counts (256 tiles per round; `blk_round` forces several rounds on small
    rows), writes the dense records' second halves, completes the runs that cross their chunk and fills the first-run table.
    Same images as the oracle's: the usual mix with a crop, a strip whose tiles of the run kernel hold many rows (1080 x 40), one
    row of more than 64 such tiles (4096 x 16), light rows (the 512-lane variant) and heavy ones (`carry_slices=1`: the 1 024-lane
    one); every per-frame buffer poisoned (what lies between two tiles' records is never a run); after the scene SHRANK."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", switch + ",poison_frame=255")
    c = forma_amd.Context(0)
    blk_frames = 0
    try:
        cases = (((1024, 768), S.random_mixed(n=400, width=1024, height=768, seed=41), S.random_mixed(n=150, width=1024, height=768, seed=42)),
                 ((1080, 40), S.random_cubics(n=300, width=1080, height=40, seed=43, alpha=0.6), S.random_cubics(n=40, width=1080, height=40, seed=44)),
                 ((4096, 16), S.random_cubics(n=1500, width=4096, height=16, seed=45, alpha=0.5), S.random_cubics(n=900, width=4096, height=16, seed=46)),
                 ((2048, 256), S.random_cubics(n=700, width=2048, height=256, seed=47, alpha=0.7), S.random_cubics(n=90, width=2048, height=256, seed=48)))
        for (w, h), big, small in cases:
            for comp in (big, small):                    # (the second scene of a canvas has fewer runs in every row)
                o, _ = both(c, comp)
                for crop in (None, (16 * 2 + 3, w - 21, 0, h) if h <= 48 else (16 * 2 + 3, w - 21, 33, h - 50)):
                    ref = o.render(w, h, clear=(1.0, 1.0, 1.0, 1.0), crop=crop)
                    for k in range(4):                   # (the first frame of a geometry is synchronous)
                        img, _tm = c.render(w, h, clear=(1.0, 1.0, 1.0, 1.0), crop=crop, timings=True)
                        names = [nm for nm, _st, _t0, _us in c.kernel_times()]
                        # (with the chain switched off, a frame that finds its runs without the counting pass is a BLOCKS frame)
                        if any("k_runs_wave" in nm for nm in names) and not any("k_runs_count" in nm for nm in names):
                            blk_frames += 1
                        y0, y1, x0, x1 = (crop[2], crop[3], crop[0], crop[1]) if crop else (0, h, 0, w)
                        a = img.reshape(h, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16); b = ref.reshape(h, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16)
                        assert np.abs(a - b).max() <= 1, (switch, (w, h), crop, k)
        # the switch is a request, the carry variant decides: the numbering must have been used by most read-back-free frames
        if switch.startswith("runs_blk=1"):
            assert blk_frames >= 24, (switch, blk_frames)
        else:
            assert blk_frames == 0, (switch, blk_frames)
    finally:
        c.close()


@pytest.mark.parametrize("switch", ["carry_half=2", "carry_half=4", ""])
def test_sliced_tile_rows_on_half_workgroups(monkeypatch, switch):
    """k_carry_rows<true, 2048, 4, 512> with SEVERAL workgroups per tile row (each a range of layers): the policy takes it for
    frames of many sliced rows (1080p), `carry_half=N` forces N slices.  Heavy rows on a canvas of 48 tile rows, translucent
    layers (every carry is visible), a crop, then fewer shapes (the slices' predictions void): images equal the oracle's."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", (switch + "," if switch else "") + "poison_frame=255")
    c = forma_amd.Context(0)
    try:
        for comp in (S.random_cubics(n=900, width=1920, height=768, seed=51, alpha=0.5), S.random_mixed(n=500, width=1920, height=768, seed=52),
                     S.random_cubics(n=120, width=1920, height=768, seed=53, alpha=0.8)):
            o, _ = both(c, comp)
            for crop in (None, (40, 1900, 35, 700)):
                ref = o.render(1920, 768, clear=(0.2, 0.3, 0.4, 1.0), crop=crop)
                for k in range(4):
                    img = c.render(1920, 768, clear=(0.2, 0.3, 0.4, 1.0), crop=crop)
                    y0, y1, x0, x1 = (crop[2], crop[3], crop[0], crop[1]) if crop else (0, 768, 0, 1920)
                    a = img.reshape(768, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16); b = ref.reshape(768, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16)
                    assert np.abs(a - b).max() <= 1, (switch, crop, k)
    finally:
        c.close()
