"""Pins the CPU oracle against the reference's own unit-test vectors.

Every expected value below is the literal from the cited reference test; nothing is re-derived.
"""
import numpy as np
import pytest

from oracle import oracle as orc

BIG = float(np.float32(2.0 ** 64))  # `usize::MAX as f32` (reference cpu/rasterizer.rs:188)


def line_segments(lines, same_layer=True, sort=False):
    """reference cpu/rasterizer.rs:176-195 `segments()` / painter/mod.rs:936-963 `line_segments()`."""
    o = orc.Oracle()
    xs, ys, slots = [], [], []
    for i, ((ax, ay), (bx, by)) in enumerate(lines):
        if xs:
            slots.append(orc.NONE)
        xs += [ax, bx]; ys += [ay, by]
        slots.append(0 if same_layer else i)
    n_geoms = 1 if same_layer else len(lines)
    geoms = np.zeros(n_geoms, orc.GEOM_DTYPE)
    geoms["order"] = np.arange(n_geoms)
    o.set_geometry(xs, ys, slots)
    o.set_geoms(geoms)
    o.prepare_lines(BIG, BIG)
    segs = o.rasterize()
    return o.sort() if sort else segs


def areas_and_covers(p0, p1):
    f = orc.seg_fields(line_segments([(p0, p1)]))
    return list(zip(f["double_area"].tolist(), f["cover"].tolist()))


def tiles(p0, p1):
    f = orc.seg_fields(line_segments([(p0, p1)]))
    return list(zip(f["tile_x"].tolist(), f["tile_y"].tolist(), f["local_x"].tolist(), f["local_y"].tolist()))


# ---- cpu/rasterizer.rs:204-244 -------------------------------------------------------------
def test_find_first_7():
    L = orc.lib()
    got = [L.oracle_find(i - 1, 2.0, 3.0, 0.2, 0.1) for i in range(7)]
    exp = [np.float32(v) for v in [0.1, 0.2, 2.2, 3.1, 4.2, 6.1, 6.2]]
    assert [np.float32(g) for g in got] == exp


def test_find_ab_large_ratio():
    L = orc.lib()
    got = [np.float32(L.oracle_find(i - 1, 16_777_216.0, 0.0001, 10.0, 0.00001)) for i in range(2, 4)]
    assert got == [np.float32(0.00021), np.float32(0.00031)]


# ---- cpu/rasterizer.rs:246-401 -------------------------------------------------------------
OCTANTS = [
    ((0, 0), (3, 2), [(11 * 16, 11), (5 * 8 + 2 * (5 * 8), 5), (5 * 8, 5), (11 * 16, 11)]),
    ((0, 0), (2, 3), [(16 * 11 + 2 * (16 * 5), 16), (8 * 5, 8), (8 * 5 + 2 * (8 * 11), 8), (16 * 11, 16)]),
    ((0, 0), (-2, 3), [(16 * 11, 16), (8 * 5 + 2 * (8 * 11), 8), (8 * 5, 8), (16 * 11 + 2 * (16 * 5), 16)]),
    ((0, 0), (-3, 2), [(11 * 16, 11), (5 * 8, 5), (5 * 8 + 2 * (5 * 8), 5), (11 * 16, 11)]),
    ((3, 2), (0, 0), [(-(11 * 16), -11), (-(5 * 8), -5), (-(5 * 8 + 2 * (5 * 8)), -5), (-(11 * 16), -11)]),
    ((2, 3), (0, 0), [(-(16 * 11), -16), (-(8 * 5 + 2 * (8 * 11)), -8), (-(8 * 5), -8), (-(16 * 11 + 2 * (16 * 5)), -16)]),
    ((0, 3), (2, 0), [(-(16 * 11 + 2 * (16 * 5)), -16), (-(8 * 5), -8), (-(8 * 5 + 2 * (8 * 11)), -8), (-(16 * 11), -16)]),
    ((0, 2), (3, 0), [(-(11 * 16), -11), (-(5 * 8 + 2 * (5 * 8)), -5), (-(5 * 8), -5), (-(11 * 16), -11)]),
]


@pytest.mark.parametrize("p0,p1,exp", OCTANTS)
def test_area_cover_octants(p0, p1, exp):
    assert areas_and_covers(p0, p1) == exp


AXES = [
    ((0, 0), (1, 0), []),
    ((0, 0), (1, 1), [(16 * 16, 16)]),
    ((0, 0), (0, 1), [(2 * 16 * 16, 16)]),
    ((0, 0), (-1, 1), [(16 * 16, 16)]),
    ((0, 0), (-1, 0), []),
    ((1, 1), (0, 0), [(-(16 * 16), -16)]),
    ((0, 1), (0, 0), [(2 * -(16 * 16), -16)]),
    ((0, 1), (1, 0), [(-(16 * 16), -16)]),
]


@pytest.mark.parametrize("p0,p1,exp", AXES)
def test_area_cover_axes(p0, p1, exp):
    assert areas_and_covers(p0, p1) == exp


# ---- cpu/rasterizer.rs:403-543 -------------------------------------------------------------
T = 16
TILE_OCTANTS = [
    ((T, T), (T + 3, T + 2), [(1, 1, 0, 0), (1, 1, 1, 0), (1, 1, 1, 1), (1, 1, 2, 1)]),
    ((T, T), (T + 2, T + 3), [(1, 1, 0, 0), (1, 1, 0, 1), (1, 1, 1, 1), (1, 1, 1, 2)]),
    ((-T, T), (-T - 2, T + 3), [(-1, 1, T - 1, 0), (-1, 1, T - 1, 1), (-1, 1, T - 2, 1), (-1, 1, T - 2, 2)]),
    ((-T, T), (-T - 3, T + 2), [(-1, 1, T - 1, 0), (-1, 1, T - 2, 0), (-1, 1, T - 2, 1), (-1, 1, T - 3, 1)]),
    ((-T, T), (-T - 3, T - 2), [(-1, 0, T - 1, T - 1), (-1, 0, T - 2, T - 1), (-1, 0, T - 2, T - 2), (-1, 0, T - 3, T - 2)]),
    ((-T, T), (-T - 2, T - 3), [(-1, 0, T - 1, T - 1), (-1, 0, T - 1, T - 2), (-1, 0, T - 2, T - 2), (-1, 0, T - 2, T - 3)]),
    ((T, T), (T + 2, T - 3), [(1, 0, 0, T - 1), (1, 0, 0, T - 2), (1, 0, 1, T - 2), (1, 0, 1, T - 3)]),
    ((T, T), (T + 3, T - 2), [(1, 0, 0, T - 1), (1, 0, 1, T - 1), (1, 0, 1, T - 2), (1, 0, 2, T - 2)]),
]


@pytest.mark.parametrize("p0,p1,exp", TILE_OCTANTS)
def test_tile_octants(p0, p1, exp):
    assert tiles(p0, p1) == exp


def test_start_and_end_not_on_pixel_border():  # cpu/rasterizer.rs:545-556
    assert areas_and_covers((0.5, 0.25), (4.0, 2.0))[0] == (4 * 8, 4)
    assert areas_and_covers((0.0, 0.0), (3.5, 1.75))[4] == (4 * 8 + 2 * (4 * 8), 4)


# ---- cpu/pixel_segment.rs:220-369 ----------------------------------------------------------
def _ps(layer, tx, ty, lx, ly, dam, cover):
    v = np.array([orc.lib().oracle_pixel_segment_new(layer, tx, ty, lx, ly, dam, cover)], np.uint64)
    f = orc.seg_fields(v)
    return {k: int(a[0]) for k, a in f.items()}


def test_pixel_segment():  # :229-250
    f = _ps(3, 4, 5, 6, 7, 8, 9)
    assert (f["layer"], f["tile_x"], f["tile_y"], f["local_x"], f["local_y"], f["double_area"], f["cover"]) == (3, 4, 5, 6, 7, 8 * 9, 9)


def test_pixel_segment_max_min():  # :252-318
    layer_max = (1 << 21) - 1
    f = _ps(layer_max, (1 << 12) - 2, (1 << 11) - 2, 15, 15, 32, 16)
    assert (f["layer"], f["tile_x"], f["tile_y"], f["local_x"], f["local_y"], f["double_area"], f["cover"]) == (
        layer_max, (1 << 12) - 2, (1 << 11) - 2, 15, 15, 32 * 16, 16)
    f = _ps(0, -1, -1, 0, 0, 0, -16)
    assert (f["layer"], f["tile_x"], f["tile_y"], f["double_area"], f["cover"]) == (0, -1, -1, 0, -16)


def test_pixel_segment_clipping():  # :320-355: every tile < -1 collapses to -1
    f = _ps(0, -2, -2, 0, 0, 0, 0)
    assert (f["tile_x"], f["tile_y"]) == (-1, -1)
    f = _ps(0, -32768, -32768, 0, 0, 0, 0)
    assert (f["tile_x"], f["tile_y"]) == (-1, -1)


def test_sort_is_key_order_and_stable():  # Ord = v >> 20, pixel_segment.rs:161-171
    rng = np.random.default_rng(0)
    v = rng.integers(0, 2 ** 63, 5000, dtype=np.uint64) & np.uint64(0x001F_FFFF_FFFF_FFFF)
    v |= (rng.integers(0, 8, 5000, dtype=np.uint64) << np.uint64(53))
    w = v.copy()
    orc.lib().oracle_sort_array(w.ctypes.data, len(w))
    order = np.argsort(v >> np.uint64(20), kind="stable")
    assert np.array_equal(w, v[order])


# ---- cpu/painter/mod.rs:1012-1040, 1502-1531 -----------------------------------------------
def test_double_area_to_coverage():
    cov = orc.lib().oracle_coverage
    area = 512
    nz = [(-area * 2, 1.0), (-area * 3 // 2, 1.0), (-area, 1.0), (-area // 2, 0.5), (0, 0.0), (area // 2, 0.5),
          (area, 1.0), (area * 3 // 2, 1.0), (area * 2, 1.0)]
    for a, e in nz:
        assert cov(a, 0) == e
    eo = [(-area * 3 // 2, 0.5), (-area, 1.0), (-area // 2, 0.5), (0, 0.0), (area // 2, 0.5), (area, 1.0), (area * 3 // 2, 0.5)]
    for a, e in eo:
        assert cov(a, 1) == e


def test_f32_to_u8_scaled_and_srgb():
    L = orc.lib()
    # f32_to_u8_scaled (:1502-1515)
    assert L.oracle_to_u8(-0.001) == 0 and L.oracle_to_u8(1.001) == 255
    for i in range(255):
        assert L.oracle_to_u8(float(np.float32(i) * (np.float32(1.0) / np.float32(255.0)))) == i
    # srgb (:1517-1531)
    col = np.array([0.001 * 0.5, 0.2 * 0.5, 0.5 * 0.5, 0.5], np.float32)
    out = np.zeros(4, np.uint8)
    L.oracle_srgb_bytes(col.ctypes.data, out.ctypes.data)
    assert out.tolist() == [2, 89, 137, 128]


# ---- cpu/painter/styling.rs test_blend_mode_* (:700-731): SIMD form vs scalar form, eps 1e-3 --
def test_blend_simd_matches_scalar():
    L = orc.lib()
    colors = [(0.125, 0.25, 0.625, 0.5), (0.25, 0.125, 0.75, 0.5), (0.625, 0.5, 0.125, 0.5), (0.375, 1.0, 0.875, 0.5),
              (0.5, 0.5, 0.5, 0.5), (0.875, 0.125, 0.0, 0.5)]
    for mode in range(16):
        for d in colors:
            for s in colors:
                dd = np.array(d, np.float32); ss = np.array(s, np.float32); out = np.zeros(3, np.float32)
                L.oracle_blend_simd(mode, dd.ctypes.data, ss.ctypes.data, out.ctypes.data)
                for c in range(3):
                    ref = L.oracle_blend_fn(mode, c, dd.ctypes.data, ss.ctypes.data)
                    assert abs(out[c] - ref) <= 1e-3, (mode, d, s, c, out[c], ref)


# ---- styling.rs f16 (:450-500) ----------------------------------------------------------------
def test_f16_roundtrip_alpha_distinct():
    L = orc.lib()
    hs = {L.oracle_f32_to_f16(float(np.float32(a) / np.float32(255.0))) for a in range(256)}
    assert len(hs) == 256
    mse = np.mean([(a / 255.0 - L.oracle_f16_to_f32(L.oracle_f32_to_f16(float(np.float32(a) / np.float32(255.0))))) ** 2 for a in range(256)])
    assert mse < 5e-8
