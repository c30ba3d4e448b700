"""Randomised whole-frame parity after the round-2 kernel changes: many seeds of the all-features scene generator at
canvas shapes chosen to hit the paths a fixed scene list misses — tile rows with more runs than one piece of the carry
walk (4096), rows beyond the in-LDS sort's capacity (16384: global run-key sort), one-tile-high and one-tile-wide
canvases, canvases that are not multiples of the tile size.  Every case: first frame (synchronous) and second frame
(read-back-free) both bit-equal to the oracle in the sorted stream and in the image."""
import os

import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

CASES = [
    # seed, shapes, width, height
    (11, 300, 512, 384), (12, 300, 333, 211), (13, 120, 16, 400), (14, 120, 700, 16), (15, 500, 1000, 1000),
    (16, 800, 1920, 1080), (17, 60, 50, 50), (18, 1500, 2048, 96),
    (19, 6000, 4096, 64),          # rows with > 4096 runs: several pieces of the carry walk
    (20, 16000, 16384, 48),        # rows with > 16384 runs: the in-LDS sort does not fit, global run-key sort
    (21, 2000, 3840, 2160), (22, 40, 4000, 30),
]


@pytest.mark.parametrize("seed,n,W,H", CASES)
def test_random_scene_matches_oracle(seed, n, W, H):
    import forma_amd
    comp = S.random_mixed(n=n, width=W, height=H, seed=seed)
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(o, t)
    clear = (0.1, 0.2, 0.3, 1.0) if seed % 2 else (1.0, 1.0, 1.0, 0.0)
    want = o.render(W, H, clear=clear)
    want_sorted = o.segments(1)
    c = forma_amd.Context(0)
    try:
        S.load(c, t)
        for frame in range(3):                                           # synchronous, then read-back-free twice
            got = c.render(W, H, clear=clear)
            assert np.array_equal(c.segments(1), want_sorted), (seed, frame)
            assert np.array_equal(got, want), (seed, frame, int(np.abs(got.astype(int) - want.astype(int)).max()))
    finally:
        c.close()


def test_rows_beyond_the_local_sort_take_the_global_sort_and_very_deep_tiles_render():
    """Capacity edges of the carry pre-pass / painter: (a) case 20 above really has a tile row with more than 16384 runs (so
    k_carry_rows<false> + the global run-key sort ran, not the in-LDS sort); (b) a tile with more layers than the painter's
    4096-entry LDS lists RENDERS — the reference has no such limit (LayerWorkbench::populate_layers,
    layer_workbench/mod.rs:250-278): k_paint_deep records the tile, k_paint_huge paints it with lists in global memory —
    on the synchronous frame and on the read-back-free ones, with and without a buffer-layer cache."""
    import forma_amd
    W, H = 16384, 48
    o = orc.Oracle()
    t = S.random_mixed(n=16000, width=W, height=H, seed=20).tables(o)
    c = forma_amd.Context(0)
    try:
        S.load(c, t)
        c.render(W, H)
        srt = c.segments(1)
        key = srt >> np.uint64(20)                                       # (tile_y, tile_x, layer)
        heads = np.concatenate(([True], key[1:] != key[:-1]))
        ty = (srt[heads] >> np.uint64(53)).astype(np.int64) - 1
        per_row = np.bincount(ty[(ty >= 0) & (ty < 3)], minlength=3)
        assert per_row.max() > 16384, per_row
        rng = np.random.default_rng(3)
        comp = S.Composition()
        for order in range(9000):                                        # 9000 translucent layers over the same tiles
            x0, y0 = float(rng.uniform(0, 20)), float(rng.uniform(0, 20))
            shape = S.custom_square(x0, y0, x0 + float(rng.uniform(4, 30)), y0 + float(rng.uniform(4, 30)))
            comp.get_mut_or_insert_default(order).insert(shape).set_props(
                S.solid((float(rng.random()), float(rng.random()), float(rng.random()), 0.02 if order % 7 else 1.0)))
        t2 = comp.tables(o)
        S.load(o, t2); S.load(c, t2)
        want = o.render(48, 48)
        for frame in range(3):
            got = c.render(48, 48)
            assert np.abs(want.astype(int) - got.astype(int)).max() <= 1, frame
        assert np.array_equal(c.segments(1), o.segments(1))
        bo, bc = np.zeros((48, 48 * 4), np.uint8), np.zeros((48, 48 * 4), np.uint8)
        for frame in range(2):                                           # the same through a buffer-layer cache
            o.render(48, 48, cache_id=1, dst=bo); c.render(48, 48, cache_id=1, dst=bc)
            assert np.abs(bo.astype(int) - bc.astype(int)).max() <= 1, frame
    finally:
        c.close()


@pytest.mark.parametrize("layers", [100, 129, 600, 1024, 1025, 2500, 4096, 4097])
def test_every_tier_of_the_deep_painters(layers):
    """A tile's layer list decides who paints it: <= 128 entries the wave painters, <= 1 024 `k_paint_deep<1024>` (four workgroups
    per CU, round 6), <= 4 096 `k_paint_deep<4096>`, beyond that `k_paint_huge` (lists in global memory).  Translucent squares
    (nothing is culled, every layer blends) stacked over the same few tiles, counts on both sides of every tier's limit, with a
    few tiles of every depth below in the same frame; synchronous, read-back-free and three-slot frames."""
    import forma_amd
    rng = np.random.default_rng(1000 + layers)
    comp = S.Composition()
    for order in range(layers):
        # the first tile (0..16 x 0..16) is covered by EVERY layer; tiles to the right by fewer and fewer
        x1 = 16.0 + float(rng.uniform(0, 112)) * (order % 3 != 0)
        comp.get_mut_or_insert_default(order).insert(S.custom_square(1.0, 1.0 + float(rng.uniform(0, 6)), x1, 15.0)).set_props(
            S.solid((float(rng.random()), float(rng.random()), float(rng.random()), 0.03)))
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(o, t)
    want = o.render(160, 32)
    c = forma_amd.Context(0)
    try:
        S.load(c, t)
        for frame in range(3):
            got = c.render(160, 32)
            assert np.abs(want.astype(int) - got.astype(int)).max() <= 1, (layers, frame)
        assert np.array_equal(c.segments(1), o.segments(1))
        c.set_frames_in_flight(3)
        for frame in range(7):
            c.render(160, 32, device_only=True)
        assert np.abs(want.astype(int) - c.read_image(160, 32).astype(int)).max() <= 1, (layers, "slots")
    finally:
        c.close()


def _runs_per_row(sorted_stream, tiles_w, tiles_h):
    key = sorted_stream >> np.uint64(20)
    heads = sorted_stream[np.concatenate(([True], key[1:] != key[:-1]))]
    ty = (heads >> np.uint64(53)).astype(np.int64) - 1
    tx = ((heads >> np.uint64(41)) & np.uint64(0xFFF)).astype(np.int64)
    ok = (ty >= 0) & (ty < tiles_h) & (tx <= tiles_w)
    return np.bincount(ty[ok], minlength=tiles_h)


def test_carry_rows_with_the_rows_covers_in_lds_and_rows_that_outgrow_it():
    """k_carry_rows<.., COVL> (round 6): a tile row in ONE slice whose runs fit 5 632 stages its cover sums and style summaries in
    LDS before the walk.  Rows just below the cap (the variant runs), a transform that grows them beyond it between two
    read-back-free frames (the frame that guessed COVL is void, the re-run takes the large variant, the guess is not made again
    for this geometry), and back; the 512-lane form on light rows.  Images and streams against the oracle every time."""
    import forma_amd
    W, H = 4096, 48
    o = orc.Oracle()
    c = forma_amd.Context(0)
    try:
        seen_below = seen_above = False
        for n in (70, 110):
            t = S.random_cubics(n=n, width=W, height=H, seed=70 + n, alpha=0.6).tables(o)
            S.load(o, t); S.load(c, t)
            for scale in (1.0, 1.0, 1.6, 1.6, 1.0, 0.5):       # (n = 70: rows of 5 291 -> 5 676 -> 5 291 runs)
                g = t["geoms"].copy()
                g["flags"] = 1
                g["xf"] = np.array([scale, 0.0, 0.0, 1.0, 0.0, 0.0], np.float32)
                o.set_geoms(g); c.set_geoms(g)
                want = o.render(W, H)
                for frame in range(3):
                    got = c.render(W, H)
                    assert np.abs(want.astype(int) - got.astype(int)).max() <= 1, (n, scale, frame)
                assert np.array_equal(c.segments(1), o.segments(1)), (n, scale)
                mx = int(_runs_per_row(o.segments(1), W // 16, H // 16).max())
                seen_below |= 2048 < mx <= 5632
                seen_above |= mx > 5632
        assert seen_below and seen_above, "the scenes must bracket the COVL variant's capacity"
    finally:
        c.close()


def test_layers_cut_by_the_bottom_edge_in_a_partial_last_tile_row():
    """Canvas height not a multiple of 16: lines entirely below the canvas are culled (segment.rs:41-52), so a layer that crosses
    the bottom edge keeps a non-zero cover on the invisible pixel rows of the last tile row, which the reference carries through
    every tile to the right (and paints with zero visible coverage).  16 000 shapes on 8192 x 40 put 4 529 such layers into
    one tile.  Without a buffer-layer cache nothing can observe them and the carry pre-pass drops carries that are empty on the
    VISIBLE rows (same pixels, shallow tiles); with a cache they are state (a tile's layer count, passes/tile_unchanged.rs)
    and are carried like the reference's — the tile is then deeper than the painter's LDS lists and goes to k_paint_huge."""
    import forma_amd
    W, H = 8192, 40
    o = orc.Oracle()
    t = S.random_mixed(n=16000, width=W, height=H, seed=20).tables(o)
    S.load(o, t)
    want = o.render(W, H)
    c = forma_amd.Context(0)
    try:
        S.load(c, t)
        for _ in range(2):
            assert np.array_equal(c.render(W, H), want)
        bo, bc = np.full((H, W * 4), 201, np.uint8), np.full((H, W * 4), 201, np.uint8)
        for frame in range(2):
            o.render(W, H, cache_id=2, dst=bo); c.render(W, H, cache_id=2, dst=bc)
            assert np.array_equal(bo, bc), frame
    finally:
        c.close()


@pytest.mark.parametrize("slices,global_sort", [(2, False), (3, False), (8, False), (2, True), (5, True)])
def test_carry_pre_pass_with_several_workgroups_per_tile_row(slices, global_sort, monkeypatch):
    """k_carry_rows shares a tile row between `slices` workgroups, each a range of layers (single GPU: by itself only when
    the rows are few; a multi-GPU band: always).  FORMA_HIP_DEBUG=carry_slices=N forces the count: every count, both run orders
    (in-LDS sort of a slice / cut of the globally sorted keys), the same bits as the oracle — frames 1 (synchronous) to 3."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", "carry_slices=%d%s" % (slices, ",global_runsort" if global_sort else ""))
    for seed, n, W, H in ((31, 400, 640, 480), (32, 2500, 2048, 64), (33, 150, 100, 700)):
        comp = S.random_mixed(n=n, width=W, height=H, seed=seed)
        o = orc.Oracle()
        t = comp.tables(o)
        S.load(o, t)
        want = o.render(W, H, clear=(0.3, 0.2, 0.1, 1.0))
        c = forma_amd.Context(0)
        try:
            S.load(c, t)
            for frame in range(3):
                got = c.render(W, H, clear=(0.3, 0.2, 0.1, 1.0))
                assert np.array_equal(got, want), (seed, slices, frame)
            bo, bc = np.zeros((H, W * 4), np.uint8), np.zeros((H, W * 4), np.uint8)
            for frame in range(2):
                o.render(W, H, cache_id=0, dst=bo); c.render(W, H, cache_id=0, dst=bc)
                assert np.array_equal(bo, bc), (seed, slices, frame)
        finally:
            c.close()


@pytest.mark.parametrize("slices,global_sort", [(1, False), (3, False), (8, False), (1, True), (4, True)])
def test_span_group_lists(slices, global_sort, monkeypatch):
    """k_carry_rows leaves the spans of a tile row a second time by group of 16 tile columns and the wave painter scans its
    group's list instead of the row's (by itself only on frames whose rows hold > 256 spans; FORMA_HIP_DEBUG=span_groups forces them
    on every frame and row).  Canvases of 1, 2, 7 and 128 groups, every slice count, both run orders, with and without a
    cache: the same bits as the oracle.  The third scene is wide and shallow: its spans cross many groups, the static pool
    (two entries per run) does not hold them and those rows fall back to the row lists on the device."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", "span_groups,carry_slices=%d%s" % (slices, ",global_runsort" if global_sort else ""))
    scenes = [(S.random_mixed(n=400, width=640, height=480, seed=41), 640, 480),
              (S.random_mixed(n=2500, width=2048, height=64, seed=42), 2048, 64),
              (S.random_mixed(n=150, width=100, height=700, seed=43), 100, 700),
              (S.random_mixed(n=60, width=256, height=48, seed=44), 256, 48)]
    wide = S.Composition()                                      # 300 canvas-wide bars: every span crosses all 8 groups
    for i in range(300):
        wide.get_mut_or_insert_default(i).insert(S.custom_square(3 + (i % 5), (i * 7) % 90, 2040 - (i % 11), (i * 7) % 90 + 9)).set_props(
            S.Props(fill=(0.1 + (i % 7) * 0.1, 0.5, 0.9 - (i % 5) * 0.1, 0.6)))
    scenes.append((wide, 2048, 96))
    for k, (comp, W, H) in enumerate(scenes):
        o = orc.Oracle()
        t = comp.tables(o)
        S.load(o, t)
        want = o.render(W, H, clear=(0.3, 0.2, 0.1, 1.0))
        c = forma_amd.Context(0)
        try:
            S.load(c, t)
            for frame in range(3):
                got = c.render(W, H, clear=(0.3, 0.2, 0.1, 1.0))
                assert np.array_equal(got, want), (k, slices, frame)
            bo, bc = np.zeros((H, W * 4), np.uint8), np.zeros((H, W * 4), np.uint8)
            for frame in range(2):
                o.render(W, H, cache_id=0, dst=bo); c.render(W, H, cache_id=0, dst=bc)
                assert np.array_equal(bo, bc), (k, slices, frame)
        finally:
            c.close()


_POISON_SCRIPT = r"""
import os, sys
import os

import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import scene as S
from oracle import oracle as orc
import forma_amd
for seed, n, W, H in ((51, 300, 640, 480), (52, 40, 100, 52)):
    comp = S.random_mixed(n=n, width=W, height=H, seed=seed)
    o = orc.Oracle(); t = comp.tables(o); S.load(o, t)
    want = o.render(W, H, clear=(0.2, 0.4, 0.6, 1.0))
    c = forma_amd.Context(0); S.load(c, t)
    for frame in range(4):                       # synchronous frame, then read-back-free ones
        assert np.array_equal(c.render(W, H, clear=(0.2, 0.4, 0.6, 1.0)), want), (seed, frame)
    assert np.array_equal(c.segments(1), o.segments(1))
    bo, bc = np.zeros((H, W * 4), np.uint8), np.zeros((H, W * 4), np.uint8)
    for frame in range(3):
        o.render(W, H, cache_id=1, dst=bo); c.render(W, H, cache_id=1, dst=bc)
        assert np.array_equal(bo, bc), (seed, frame)
    c.close()
print("poison ok")
"""


@pytest.mark.parametrize("byte", ["0xFF", "0xA5", "0x00"])
def test_nothing_reads_what_it_did_not_write(byte):
    """FORMA_HIP_DEBUG=poison=BYTE fills every fresh device allocation with one byte: frames (synchronous and read-back-free, with and
    without a cache, span group lists forced on) must not depend on what a buffer held before this frame wrote it — hipMalloc
    does not zero, and a fresh box hands out whatever the last tenant left."""
    import subprocess, sys
    env = dict(os.environ, FORMA_HIP_DEBUG="poison=%s,span_groups" % byte)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _POISON_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "poison ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])


@pytest.mark.parametrize("block", list(range(int(os.environ.get("FORMA_TEST_FUZZ_BLOCKS", "3")))))
def test_random_frames_with_crops_and_channel_orders(block):
    """Twenty random all-features scenes per block on canvases of any shape, a quarter of them cropped, under four channel orders
    (one puts alpha first and red last: there a tile folded to a solid colour and a painted tile are DIFFERENT bytes —
    to_srgb_bytes of the selected channels vs. select after encoding, painter/mod.rs:156-162, 466-483 — so every fold decision
    has to be the reference's), translucent clear colours; synchronous, read-back-free and in-flight frames: sorted stream
    bit-equal, image within one code value of the oracle.  (FORMA_TEST_FUZZ_BLOCKS=30 for a long run.)"""
    import forma_amd
    c = forma_amd.Context(0, frames_in_flight=3 if block % 2 else 1)
    o = orc.Oracle()
    try:
        for seed in range(block * 20, block * 20 + 20):
            rng = np.random.default_rng(7000 + seed)
            w, h = int(rng.integers(17, 900)), int(rng.integers(17, 500))
            comp = S.random_mixed(n=int(rng.integers(1, 260)), width=w, height=h, seed=9000 + seed)
            if seed % 5 == 0:                             # every fifth scene all solid / Over / unclipped: the painter's own kernel for those
                for layer in comp.layers.values():
                    layer.set_props(S.Props(fill_rule=layer.props.fill_rule,
                                            fill=tuple(float(v) for v in rng.random(3)) + ((1.0,) if rng.random() < 0.4 else (float(rng.random()),))))
            t = comp.tables(o)
            S.load(o, t); S.load(c, t)
            crop = None
            if seed % 4 == 1:
                x0, y0 = int(rng.integers(0, w - 1)), int(rng.integers(0, h - 1))
                crop = (x0, int(rng.integers(x0 + 1, w + 1)), y0, int(rng.integers(y0 + 1, h + 1)))
            ch = [(0, 1, 2, 3), (2, 1, 0, 3), (0, 1, 2, 5), (3, 2, 1, 0)][seed % 4]
            clear = tuple(float(v) for v in rng.random(4))
            want = o.render(w, h, clear=clear, crop=crop, channels=ch).reshape(h, w, 4).astype(int)
            m = np.ones((h, w), bool)
            if crop is not None:                          # (outside the crop's tiles neither backend writes)
                m[:] = False
                m[crop[2] // 16 * 16: min(h, (crop[3] + 15) // 16 * 16), crop[0] // 16 * 16: min(w, (crop[1] + 15) // 16 * 16)] = True
            for frame in range(4):
                if frame < 2:
                    got = c.render(w, h, clear=clear, crop=crop, channels=ch)
                else:
                    c.render(w, h, clear=clear, crop=crop, channels=ch, device_only=True)
                    got = c.read_image(w, h)
                d = np.abs(want - got.reshape(h, w, 4).astype(int))[m]
                assert d.max(initial=0) <= 1, (seed, frame, w, h, crop, ch)
                assert np.array_equal(c.segments(1), o.segments(1)), (seed, frame)
    finally:
        c.close()


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_absurd_geometry_is_an_error_not_a_fault(devices):
    """Coordinates no canvas can hold — a point at x = -1e20, 3e38, +-inf, NaN, 2^31 — ask for more pixel segments than a device
    holds (or for none).  The reference wraps its 32-bit sums in release builds and panics in debug ones (segment.rs:86-98);
    here the line kernels sum in 64 bits, saturate, and the frame ends as FORMA_E_CAPACITY (or renders, when the lines are
    dropped) — never as prefix sums that wrapped around and a device memory fault.  The context stays usable."""
    import forma_amd
    from forma_amd._lib import FormaError
    specials = [np.nan, np.inf, -np.inf, 3e38, -3e38, 1e20, -1e20, 2147483648.0, -2147483904.0, 16777216.0, -1e9]
    o = orc.Oracle()
    c = forma_amd.Context(0, devices=devices) if devices else forma_amd.Context(0, frames_in_flight=2)   # (a device that fails must not
    try:                                                                                                 #  leave the others at a barrier)
        good = S.random_mixed(n=40, width=320, height=200, seed=3)
        tg = good.tables(o)
        S.load(o, tg)
        want = o.render(320, 200)
        outcomes = set()
        for seed in range(8 if devices else 24):                        # (a failed multi-device frame re-plans twice before it gives up)
            rng = np.random.default_rng(66000 + seed)
            t = dict(S.random_mixed(n=int(rng.integers(1, 30)), width=320, height=200, seed=67000 + seed).tables(o))
            x, y = t["x"].copy(), t["y"].copy()
            for _ in range(int(rng.integers(1, 6))):
                i = int(rng.integers(0, len(x)))
                (x if rng.random() < 0.5 else y)[i] = np.float32(specials[int(rng.integers(0, len(specials)))])
            t["x"], t["y"] = x, y
            S.load(c, t)
            S.load(o, t)
            try:                                                        # (the checker refuses the same frames: oracle_render -> -4)
                want_bad = o.render(320, 200)
            except AssertionError:
                want_bad = None
            try:
                for _ in range(3):
                    c.render(320, 200, device_only=True)
                if not devices:
                    c.sync()
                outcomes.add("rendered")
                assert want_bad is not None, seed
                # (pixel coordinates beyond i32 — NaN, inf, 1e20 — go through `to_int_unchecked` in the reference, rasterizer.rs:78-80:
                #  undefined there, INT_MIN on the checker's x86, saturating on the GPU.  Bits are compared where they are defined.)
                if float(np.abs(np.nan_to_num(np.concatenate([x, y]).astype(np.float64), nan=np.inf)).max()) * 16.0 < 2.0 ** 31:
                    assert np.abs(want_bad.astype(int) - c.read_image(320, 200).astype(int)).max() <= 1, seed
                    if not devices:
                        assert np.array_equal(c.segments(1), o.segments(1)), seed
            except FormaError as e:
                assert e.code == -4, e                                  # FORMA_E_CAPACITY
                assert want_bad is None, seed
                outcomes.add("capacity")
            if seed % 4 == 3:                                           # the context is still good for a sane scene
                S.load(c, tg)
                assert np.array_equal(c.render(320, 200), want), seed
        assert outcomes == {"rendered", "capacity"}
    finally:
        c.close()
