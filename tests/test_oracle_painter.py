"""Pins the oracle's Painter / for_each_row / CachedTile / Layout::write restatement to the reference's own
unit tests of `forma/src/cpu/painter/mod.rs:1012-1825` (every test of that module; `double_area_*`,
`f32_to_u8_scaled` and `srgb` already live in test_oracle_vectors.py).  Each test is the reference test of the
same name replayed through the oracle: `paint_tile` (mod.rs:976-1000) = oracle.Workbench.drive_tile_painting +
Painter::colors, `for_each_row` (mod.rs:719-778) = Oracle.paint on a hand-built sorted stream."""
import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

TILE_WIDTH = TILE_HEIGHT = 16
PIXEL_WIDTH = 16
RED = (1.0, 0.0, 0.0, 1.0)
RED_GREEN_50 = (1.0, 0.5, 0.0, 1.0)
RED_50 = (0.5, 0.0, 0.0, 1.0)
RED_50_GREEN_50 = (0.5, 0.5, 0.0, 1.0)
GREEN = (0.0, 1.0, 0.0, 1.0)
GREEN_50 = (0.0, 0.5, 0.0, 1.0)
BLUE = (0.0, 0.0, 1.0, 1.0)
WHITE = (1.0, 1.0, 1.0, 1.0)
BLACK = (0.0, 0.0, 0.0, 1.0)
BLACK_ALPHA_50 = (0.0, 0.0, 0.0, 0.5)
BLACK_ALPHA_0 = (0.0, 0.0, 0.0, 0.0)
BLACK_RGBA, RED_RGBA, GREEN_RGBA, BLUE_RGBA = [0, 0, 0, 255], [255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255]
HUGE = float(np.float32(2.0 ** 64))     # usize::MAX as f32 (fill_cpu_view(usize::MAX, usize::MAX, ..), mod.rs:964)


def style_tables(props_by_layer):
    n = max(props_by_layer) + 1
    offsets = np.full(n, S.NONE, np.uint32)
    words = []
    for lid, p in sorted(props_by_layer.items()):
        offsets[lid] = len(words)
        words += S.encode_props(p, [])
    return offsets, np.asarray(words, np.uint32)


def line_segments(points, same_layer=False):
    """mod.rs:943-974: one line per entry (its own layer unless same_layer), rasterized and sorted."""
    o = orc.Oracle()
    xs, ys, slots = [], [], []
    for i, (p0, p1) in enumerate(points):
        xs += [p0[0], p1[0]]; ys += [p0[1], p1[1]]
        slots += [0 if same_layer else i, S.NONE]
    n_geoms = 1 if same_layer else len(points)
    geoms = np.zeros(n_geoms, orc.GEOM_DTYPE)
    geoms["order"] = np.arange(n_geoms)
    o.set_geometry(np.asarray(xs, np.float32), np.asarray(ys, np.float32), np.asarray(slots[:-1], np.uint32))
    o.set_geoms(geoms)
    o.prepare_lines(HUGE, HUGE)
    o.rasterize()
    return o.sort()


def paint_tile(carries, segments, props_by_layer, clear_color, tile_x=0, wb=None):
    """mod.rs:976-1000; returns Painter::colors(): [256][4] column-major (index = x * TILE_HEIGHT + y)."""
    if wb is None:
        o = orc.Oracle()
        o.set_styles(*style_tables(props_by_layer))
        wb = orc.Workbench(o)
        wb._keep = o
        wb.cached_tile(use=False)
        wb.init(carries)
    wb.context(segments, tile_x=tile_x, clear_color=clear_color)
    wb.drive_tile_painting()
    return wb.colors(), wb


def solid(c, **kw):
    return S.Props(fill=tuple(c), **kw)


def rows(colors, column, n=4):
    return [tuple(float(v) for v in colors[column * TILE_HEIGHT + y]) for y in range(n)]


def test_carry_cover():                 # mod.rs:1042-1077
    cover = [0] * 16; cover[1] = 16
    segments = line_segments([((0.0, 0.0), (0.0, float(TILE_HEIGHT)))])
    colors, _ = paint_tile([(1, cover)], segments, {0: solid(GREEN), 1: solid(RED)}, BLACK)
    assert rows(colors, 0, 2) == [GREEN, RED]


def test_overlapping_triangles():       # mod.rs:1079-1134
    segments = line_segments([((0.0, 0.0), (4.0, 4.0)), ((4.0, 0.0), (0.0, 4.0))])
    colors, _ = paint_tile([], segments, {0: solid(GREEN), 1: solid(RED)}, BLACK)
    assert rows(colors, 0) == [GREEN_50, BLACK, BLACK, RED_50]
    assert rows(colors, 1) == [GREEN, GREEN_50, RED_50, RED]
    assert rows(colors, 2) == [GREEN, RED_50_GREEN_50, RED, RED]
    assert rows(colors, 3) == [RED_50_GREEN_50, RED, RED, RED]


TWO_EDGES = [((0.0, 0.0), (0.0, float(TILE_HEIGHT)))] * 2


def test_transparent_overlay():         # mod.rs:1136-1167
    colors, _ = paint_tile([], line_segments(TWO_EDGES), {0: solid(RED), 1: solid(BLACK_ALPHA_50)}, BLACK)
    assert rows(colors, 0, 1) == [RED_50]


def test_linear_blend_over():           # mod.rs:1169-1200
    colors, _ = paint_tile([], line_segments(TWO_EDGES), {0: solid(RED), 1: solid((0.0, 1.0, 0.0, 0.5))}, BLACK)
    assert rows(colors, 0, 1) == [RED_50_GREEN_50]


def test_linear_blend_difference():     # mod.rs:1202-1234
    colors, _ = paint_tile([], line_segments(TWO_EDGES),
                           {0: solid(RED), 1: solid((0.0, 1.0, 0.0, 0.5), blend_mode="Difference")}, BLACK)
    assert rows(colors, 0, 1) == [RED_GREEN_50]


def test_linear_blend_hue_white_opaque_background():        # mod.rs:1236-1258
    colors, _ = paint_tile([], line_segments(TWO_EDGES[:1]), {0: solid((0.0, 1.0, 0.0, 0.5), blend_mode="Hue")}, WHITE)
    assert rows(colors, 0, 1) == [WHITE]


def test_linear_blend_hue_white_transparent_background():   # mod.rs:1260-1282
    colors, _ = paint_tile([], line_segments(TWO_EDGES[:1]), {0: solid((0.0, 1.0, 0.0, 0.5), blend_mode="Hue")},
                           (1.0, 1.0, 1.0, 0.0))
    assert rows(colors, 0, 1) == [(0.5, 1.0, 0.5, 0.5)]


def test_cover_carry_is_empty():        # mod.rs:1284-1343
    for v, want in [(0, True), (1, False), (-1, False), (16, False), (-16, False)]:
        assert orc.cover_is_empty([v] * 16, even_odd=False) is want
    for v, want in [(0, True), (1, False), (-1, False), (16, False), (-16, False), (32, True), (-32, True), (48, False), (-48, False)]:
        assert orc.cover_is_empty([v] * 16, even_odd=True) is want


def test_cover_carry_is_full():         # mod.rs:1345-1404
    for v, want in [(0, False), (1, False), (-1, False), (16, True), (-16, True)]:
        assert orc.cover_is_full([v] * 16, even_odd=False) is want
    for v, want in [(0, False), (1, False), (-1, False), (16, True), (-16, True), (32, False), (-32, False), (48, True), (-48, True)]:
        assert orc.cover_is_full([v] * 16, even_odd=True) is want


def test_clip():                        # mod.rs:1406-1500
    segments = line_segments([((0.0, 0.0), (4.0, 4.0)), ((0.0, 0.0), (0.0, 4.0)), ((0.0, 0.0), (0.0, 4.0))])
    props = {0: S.Props(clip=2), 1: solid(GREEN, is_clipped=True), 2: solid(RED, is_clipped=True), 3: solid(GREEN)}
    colors, wb = paint_tile([], segments, props, BLACK)
    col = [BLACK] * 4
    for i in range(4):
        col[i] = (0.5, 0.25, 0.0, 1.0)
        if i >= 1:
            col[i - 1] = RED
        assert rows(colors, i) == col
    # second tile of the same row with the same workbench: the carried covers alone
    segments = line_segments([((4.0, 0.0), (4.0, 4.0))])
    colors, _ = paint_tile(None, segments, None, BLACK, tile_x=1, wb=wb)
    for i in range(4):
        assert rows(colors, i) == [RED] * 4


def seg(layer, tile_x, tile_y=0, lx=0, ly=0, dam=0, cover=0):
    return orc.pixel_segment(layer, tile_x, tile_y, lx, ly, dam, cover)


def white_flusher(slice_):
    slice_[:] = 255


def default_style_oracle(n_layers=2):
    o = orc.Oracle()
    o.set_styles(*style_tables({i: S.Props() for i in range(n_layers)}))
    return o


def test_flusher():                     # mod.rs:1533-1572
    width = TILE_WIDTH + TILE_WIDTH // 2
    segments = sorted([seg(0, 0), seg(0, 1), seg(1, 0), seg(1, 1)])
    o = default_style_oracle()
    buf = o.paint(segments, width, TILE_HEIGHT, clear=BLACK_ALPHA_0, flusher=white_flusher)
    assert (buf == 255).all()


def test_flush_background():            # mod.rs:1574-1604
    o = default_style_oracle()
    buf = o.paint([], TILE_WIDTH, TILE_HEIGHT, clear=BLACK_ALPHA_0, flusher=white_flusher)
    assert (buf == 255).all()


def test_flusher_sees_each_written_row_slice_once():
    """Layout::write (buffer/layout/mod.rs:264-295): one flush per row slice of every written tile, slice =
    row[..TILE_WIDTH * 4] or the shorter row of an edge tile; a tile the optimizer skips is not flushed."""
    width, height = TILE_WIDTH + 5, TILE_HEIGHT + 3
    seen = []
    o = default_style_oracle()
    o.paint([], width, height, clear=BLACK, flusher=lambda s: seen.append(len(s)))
    assert sorted(seen) == sorted([64] * 16 + [20] * 16 + [64] * 3 + [20] * 3)
    # with a cache, the second identical frame writes (and flushes) nothing
    seen.clear()
    o.paint([], width, height, clear=BLACK, cache_id=0, flusher=lambda s: seen.append(len(s)))
    assert len(seen) == 38
    seen.clear()
    o.paint([], width, height, clear=BLACK, cache_id=0, flusher=lambda s: seen.append(len(s)))
    assert seen == []


def test_skip_opaque_tiles():           # mod.rs:1606-1715
    segments = [seg(2, -1, 0, TILE_WIDTH - 1, y, 0, PIXEL_WIDTH) for y in range(TILE_HEIGHT)]
    segments.append(seg(0, -1, 0, TILE_WIDTH - 1, 0, 0, PIXEL_WIDTH))
    segments.append(seg(1, 0, 0, 0, 1, 0, PIXEL_WIDTH))
    segments += [seg(2, 1, 0, TILE_WIDTH - 1, y, 0, -PIXEL_WIDTH) for y in range(TILE_HEIGHT)]
    segments.sort()
    o = orc.Oracle()
    o.set_styles(*style_tables({0: solid(BLUE), 1: solid(GREEN), 2: solid(RED)}))
    buf = o.paint(segments, TILE_WIDTH * 3, TILE_HEIGHT, clear=BLACK).reshape(TILE_HEIGHT, TILE_WIDTH * 3, 4)
    # First two tiles need to be completely red.
    assert (buf[:, : 2 * TILE_WIDTH] == RED_RGBA).all()
    # The last tile contains one blue and one green line, followed by black lines (clear color).
    assert (buf[0, 2 * TILE_WIDTH:] == BLUE_RGBA).all()
    assert (buf[1, 2 * TILE_WIDTH:] == GREEN_RGBA).all()
    assert (buf[2:, 2 * TILE_WIDTH:] == BLACK_RGBA).all()


def test_crop():                        # mod.rs:1717-1781
    segments = sorted(seg(0, 0, j, TILE_WIDTH - 1, y, 0, PIXEL_WIDTH) for j in range(3) for y in range(TILE_HEIGHT))
    o = orc.Oracle()
    o.set_styles(*style_tables({0: solid(BLUE)}))
    buf = o.paint(segments, TILE_WIDTH * 3, TILE_HEIGHT * 3, clear=RED,
                  crop=(TILE_WIDTH, TILE_WIDTH * 2 + TILE_WIDTH // 2, TILE_HEIGHT, TILE_HEIGHT * 2))
    buf = buf.reshape(TILE_HEIGHT * 3, TILE_WIDTH * 3, 4)
    # First and third rows of tiles stay untouched (zero), second row begins with an untouched tile ...
    assert not buf[:TILE_HEIGHT].any() and not buf[2 * TILE_HEIGHT:].any()
    assert not buf[TILE_HEIGHT: 2 * TILE_HEIGHT, :TILE_WIDTH].any()
    # ... followed by two blue tiles (the crop is rounded out to whole tiles, Rect::new renderer.rs:43-52)
    assert (buf[TILE_HEIGHT: 2 * TILE_HEIGHT, TILE_WIDTH:] == BLUE_RGBA).all()


def test_tiles_len():                   # mod.rs:1783-1795: LinearLayout(4 tiles wide, stride 5 tiles, 8 tiles high) -> 32 tiles
    width, stride, height = TILE_WIDTH * 4, TILE_WIDTH * 5 * 4, TILE_HEIGHT * 8
    o = default_style_oracle()
    seen = []
    buf = o.paint([], width, height, clear=BLACK, stride=stride, flusher=lambda s: seen.append(len(s)))
    assert len(seen) == 32 * TILE_HEIGHT and set(seen) == {TILE_WIDTH * 4}
    assert not buf[:, width * 4:].any()                       # bytes between width and the stride are never touched


def test_cached_tiles():                # mod.rs:1797-1814: CachedTile::{solid_color, layer_count, update_*}
    o = default_style_oracle()
    wb = orc.Workbench(o)
    assert wb.cached_tile_state() == (None, None)
    wb.cached_tile(solid_color=[255, 0, 0, 255])
    assert wb.cached_tile_state() == (None, [255, 0, 0, 255])
    wb.cached_tile(layer_count=2, solid_color=[255, 0, 0, 255])
    assert wb.cached_tile_state() == (2, [255, 0, 0, 255])
    # update_layer_count through the pass that owns it; update_solid_color(Some / None) through convert_optimizer_op
    wb.init([(0, [16] * 16)])
    wb.context([], clear_color=BLACK)
    assert wb.drive_tile_painting() == (orc.Workbench.OP_SOLID, BLACK_RGBA)      # default style: opaque black cover
    assert wb.cached_tile_state() == (1, BLACK_RGBA)
    wb.context([seg(0, 0, 0, 3, 3, 8, 4)], clear_color=BLACK)
    assert wb.drive_tile_painting()[0] == orc.Workbench.OP_COLOR_BUFFER
    assert wb.cached_tile_state() == (1, None)                                   # update_solid_color(None), :708-712
