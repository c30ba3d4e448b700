"""Pins the oracle's fills, the reference's flat-index iterator and `Point::angle` to the reference's own unit tests —
the vectors VERDICT r2 listed as covered only indirectly (through the e2e PNGs):

  cpu/painter/styling.rs:733-767   `linear_gradient`
  cpu/painter/styling.rs:769-822   `radial_gradient`
  cpu/painter/styling.rs:1014-1131 `texture_color_at_with_{identity, scale_x2, translation, axis_inverted}`
  utils/prefix_scan.rs:217-379     `empty_iter`, `local_iter`, `local_iter_rev`, `both_ends`, `empty_groups`,
                                   `empty_groups_rev`, `par_iter`, `par_iter2`
  math/point.rs:171-180            `angle`

Each test replays the reference test of the same name: same inputs, same expected values.  The GPU side of the fills is
covered by `test_gpu_fill_columns_match_the_oracle` below (`-m gpu`): the same columns painted by the HIP painter."""
import ctypes as C
import math

import numpy as np
import pytest

import scene as S
from oracle import oracle as orc


def gradient_column(grad, x, y):
    """colors(&gradient.color_at(x, y)): 8 rows (j = 0..7) of [r, g, b, a]"""
    words = np.asarray(S.encode_props(S.Props(fill=grad), []), np.uint32)
    out = np.zeros(32, np.float32)
    orc.lib().oracle_gradient_column(words.ctypes.data, x, y, out.ctypes.data)
    return out.reshape(4, 8).T


def eq4(c):                                   # color_eq!: all four channels equal, returns the value
    assert c[0] == c[1] == c[2] == c[3]
    return float(c[0])


def test_linear_gradient():                   # styling.rs:733-767
    g = S.gradient((0.0, 7.0), (7.0, 0.0), [(v,) * 4 for v in (0.25, 0.75, 0.25, 0.75, 0.25)])
    col = gradient_column(g, 0.0, 0.0)
    assert list(col[0]) == [0.25] * 4
    assert eq4(col[1]) < eq4(col[2]) < eq4(col[3])
    assert eq4(col[4]) > eq4(col[5]) > eq4(col[6])
    assert list(col[7]) == [0.25] * 4
    col = gradient_column(g, 3.0, 0.0)
    assert eq4(col[0]) < 0.75
    assert eq4(col[1]) > eq4(col[2]) > eq4(col[3])
    assert list(col[3]) == [0.25] * 4
    assert eq4(col[3]) < eq4(col[4]) < eq4(col[5]) < eq4(col[6])
    assert eq4(col[7]) < 0.75
    col = gradient_column(g, 7.0, 0.0)
    assert list(col[0]) == [0.25] * 4
    assert eq4(col[1]) < eq4(col[2]) < eq4(col[3])
    assert eq4(col[4]) > eq4(col[5]) > eq4(col[6])
    assert list(col[7]) == [0.25] * 4


def test_radial_gradient():                   # styling.rs:769-822
    e = float(np.float32(7.0) * (np.float32(1.0) / np.sqrt(np.float32(2.0))))
    g = S.gradient((0.0, 0.0), (e, e), [(0.25,) * 4, (0.75,) * 4], radial=True)
    col = gradient_column(g, 0.0, 0.0)
    assert list(col[0]) == [0.25] * 4
    assert eq4(col[1]) < eq4(col[2]) < eq4(col[3]) < eq4(col[4]) < eq4(col[5]) < eq4(col[6])
    assert list(col[7]) == [0.75] * 4
    col = gradient_column(g, 3.0, 0.0)
    assert eq4(col[0]) < eq4(col[1]) < eq4(col[2]) < eq4(col[3]) < eq4(col[4]) < eq4(col[5]) < eq4(col[6])
    assert list(col[7]) == [0.75] * 4
    col = gradient_column(g, 4.0, 0.0)
    assert eq4(col[0]) < eq4(col[1]) < eq4(col[2]) < eq4(col[3]) < eq4(col[4]) < eq4(col[5])
    assert list(col[6]) == [0.75] * 4 and list(col[7]) == [0.75] * 4
    col = gradient_column(g, 7.0, 0.0)
    for j in range(8):
        assert list(col[j]) == [0.75] * 4


# ---- Texture::color_at (styling.rs:1014-1131) ---------------------------------------------------------------------------
C00 = (0.0000, 0.03125, 0.0625, 0.09375)
C01 = (0.1250, 0.15625, 0.1875, 0.21875)
C10 = (0.2500, 0.28125, 0.3125, 0.34375)
C11 = (0.3750, 0.40625, 0.4375, 0.46875)
C20 = (0.5000, 0.53125, 0.5625, 0.59375)
C21 = (0.6250, 0.65625, 0.6875, 0.71875)


def linear_image(colors, width, height):      # Image::from_linear_rgba (styling.rs:324-331): f16::from per channel
    L = orc.lib()
    tex = np.array([[L.oracle_f32_to_f16(v) for v in c] for c in colors], np.uint16)
    return S.Image(tex, width, height)


def texture_words_and_tables(transform):
    img = linear_image([C00, C01, C10, C11, C20, C21], 2, 3)
    images = []
    words = np.asarray(S.encode_props(S.Props(fill=S.Texture(tuple(transform), img)), images), np.uint32)
    tab = np.zeros(1, orc.IMAGE_DTYPE)
    tab[0] = (0, 2, 3)
    return words, tab, np.ascontiguousarray(img.texels, np.uint16)


def apply_texture_color_at(transform):        # styling.rs:1052-1064: texture.color_at(-2.0, -2.0)
    words, tab, texels = texture_words_and_tables(transform)
    out = np.zeros(32, np.float32)
    orc.lib().oracle_texture_column(words.ctypes.data, tab.ctypes.data, 1, texels.ctypes.data, -2.0, -2.0, out.ctypes.data)
    return out.reshape(4, 8).T                # [row j][channel]


TEXTURE_CASES = {                             # AffineTransform {ux, uy, vx, vy, tx, ty} -> the eight texels of the column
    "identity": ((1.0, 0.0, 0.0, 1.0, 0.0, 0.0), [C00, C00, C00, C10, C20, C20, C20, C20]),
    "scale_x2": ((0.5, 0.0, 0.0, 0.5, 0.0, 0.0), [C00, C00, C00, C00, C10, C10, C20, C20]),
    "translation": ((1.0, 0.0, 0.0, 1.0, 1.0, 1.0), [C00, C00, C10, C20, C20, C20, C20, C20]),
    "axis_inverted": ((0.0, 1.0, 1.0, 0.0, 0.0, 0.0), [C00, C00, C00, C01, C01, C01, C01, C01]),
}


@pytest.mark.parametrize("name", list(TEXTURE_CASES))
def test_texture_color_at(name):              # styling.rs:1079-1131 (the test colours are exact in the bias-shifted half format)
    transform, want = TEXTURE_CASES[name]
    got = apply_texture_color_at(transform)
    assert np.array_equal(got, np.asarray(want, np.float32)), name


# ---- PrefixScanIter (utils/prefix_scan.rs) -----------------------------------------------------------------------------------
class PrefixScanIter:
    def __init__(self, sums=None, handle=None):
        self.L = orc.lib()
        if handle is None:
            a = np.asarray(sums, np.uint32)
            handle = self.L.oracle_psi_new(a.ctypes.data if len(a) else None, len(a))
        self.h = handle

    def __del__(self):
        self.L.oracle_psi_free(self.h)

    def _pair(self, fn):
        g, l = C.c_uint32(0), C.c_uint32(0)
        return (g.value, l.value) if fn(self.h, C.byref(g), C.byref(l)) else None

    def next(self):
        return self._pair(self.L.oracle_psi_next)

    def next_back(self):
        return self._pair(self.L.oracle_psi_next_back)

    def len(self):
        return self.L.oracle_psi_len(self.h)

    def collect(self, rev=False):
        out = []
        while True:
            v = self.next_back() if rev else self.next()
            if v is None:
                return out
            out.append(v)

    def par_collect(self):
        """rayon's `bridge` over the Producer: split in the middle down to single items (split_at :119-146), concatenate in order"""
        n = self.len()
        if n <= 1:
            return self.collect()
        right = PrefixScanIter(handle=self.L.oracle_psi_split_at(self.h, n // 2))
        return self.par_collect() + right.par_collect()


LOCAL = [(0, 0), (0, 1), (1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2), (2, 3), (3, 0), (3, 1), (3, 2), (3, 3), (3, 4), (3, 5)]


def test_prefix_scan_empty_iter():            # :222-226
    assert PrefixScanIter([]).collect() == []


def test_prefix_scan_local_iter():            # :228-254
    assert PrefixScanIter([2, 5, 9, 15]).collect() == LOCAL


def test_prefix_scan_local_iter_rev():        # :256-282
    assert PrefixScanIter([2, 5, 9, 15]).collect(rev=True) == LOCAL[::-1]


def test_prefix_scan_both_ends():             # :284-312
    it = PrefixScanIter([2, 5, 9, 15])
    assert it.len() == 15
    assert it.next() == (0, 0) and it.next_back() == (3, 5)
    assert it.next() == (0, 1) and it.next_back() == (3, 4)
    assert it.next() == (1, 0) and it.next_back() == (3, 3)
    assert it.next() == (1, 1)
    assert it.len() == 8
    assert it.next_back() == (3, 2) and it.next() == (1, 2)
    assert it.next_back() == (3, 1) and it.next() == (2, 0)
    assert it.next_back() == (3, 0) and it.next() == (2, 1)
    assert it.next_back() == (2, 3) and it.next() == (2, 2)
    assert it.next() is None and it.next_back() is None
    assert it.len() == 0


def test_prefix_scan_empty_groups():          # :314-323
    assert PrefixScanIter([2, 2, 5, 5]).collect() == [(0, 0), (0, 1), (2, 0), (2, 1), (2, 2)]


def test_prefix_scan_empty_groups_rev():      # :325-334
    assert PrefixScanIter([2, 2, 5, 5]).collect(rev=True) == [(2, 2), (2, 1), (2, 0), (0, 1), (0, 0)]


def test_prefix_scan_par_iter():              # :336-362
    assert PrefixScanIter([2, 5, 9, 15]).par_collect() == LOCAL


def test_prefix_scan_par_iter2():             # :364-378
    assert PrefixScanIter([3, 6, 10, 11]).par_collect() == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2),
                                                            (2, 3), (3, 0)]


def test_prefix_scan_is_the_order_of_the_oracles_rasterizer():
    """the oracle's rasterize() walks lines directly instead of iterating PrefixScanIter: same (line, i) sequence"""
    rng = np.random.default_rng(1)
    lens = rng.integers(0, 6, 200)
    sums = np.cumsum(lens).astype(np.uint32)
    want = [(int(li), int(i)) for li, n in enumerate(lens) for i in range(int(n))]
    assert PrefixScanIter(sums).collect() == want
    assert PrefixScanIter(sums).par_collect() == want


# ---- Point::angle (math/point.rs:171-180) -------------------------------------------------------------------------------------------
def angle(x, y):
    out = C.c_float(0)
    return out.value if orc.lib().oracle_point_angle(x, y, C.byref(out)) else None


def test_point_angle():
    f = np.float32
    assert angle(1.0, 0.0) == 0.0
    assert angle(1e10, 0.0) == 0.0
    assert angle(-1.0, 0.0) == float(f(math.pi))
    assert angle(0.0, 1.0) == float(f(math.pi / 2))
    assert angle(0.0, -1.0) == -float(f(math.pi / 2))
    assert abs(math.pi / 4 - angle(1.0, 1.0)) < 1e-3
    assert abs(-math.pi / 4 - angle(1.0, -1.0)) < 1e-3
    assert angle(0.0, 0.0) is None            # len < f32::EPSILON (point.rs:84-86)


# ---- the same fills through the HIP painter --------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_fill_columns_match_the_oracle():
    """the gradients and textures of the reference tests above as full-cover layers on a 16 x 16 canvas: the HIP painter's
    pixels equal the oracle's (which the tests above pin to the reference's expected values)"""
    import forma_amd
    e = float(np.float32(7.0) * (np.float32(1.0) / np.sqrt(np.float32(2.0))))
    fills = [S.gradient((0.0, 7.0), (7.0, 0.0), [(v, v, v, 1.0) for v in (0.25, 0.75, 0.25, 0.75, 0.25)]),
             S.gradient((0.0, 0.0), (e, e), [(0.25, 0.25, 0.25, 1.0), (0.75, 0.75, 0.75, 1.0)], radial=True)]
    img = linear_image([C00, C01, C10, C11, C20, C21], 2, 3)
    for transform, _ in TEXTURE_CASES.values():
        t = list(transform)
        t[4] -= 2.0 * (t[0] + t[2]); t[5] -= 2.0 * (t[1] + t[3])       # the tests sample from (-2, -2): pixel (0, 0) samples there
        fills.append(S.Texture(tuple(t), img))
    c = forma_amd.Context(0)
    for k, fill in enumerate(fills):
        comp = S.Composition()
        comp.get_mut_or_insert_default(0).insert(S.custom_square(0, 0, 16, 16)).set_props(S.Props(fill=fill))
        o = orc.Oracle()
        t = comp.tables(o)
        S.load(o, t); S.load(c, t)
        want = o.render(16, 16, clear=(0, 0, 0, 1))
        got = c.render(16, 16, clear=(0, 0, 0, 1))
        assert np.array_equal(got, want), k
        assert len(np.unique(want.reshape(-1, 4), axis=0)) > 1, k       # the fill really varies over the tile
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nstops", [2, 3, 11, 12, 13, 22, 23, 40])
def test_gpu_gradients_of_any_stop_count(nstops):
    """the wave painter keeps a gradient's first 11 stops in LDS and walks longer ones window by window: linear and radial
    gradients of 2..40 stops under partial coverage (circles), stacked with other blend modes, a texture-free clip and a solid
    layer on top — pixels equal the oracle's (styling.rs:58-193; the reference has no limit on the stops)"""
    import forma_amd
    rng = np.random.default_rng(nstops)
    cols = [tuple(float(v) for v in rng.uniform(0.0, 1.0, 3)) + (float(rng.uniform(0.3, 1.0)),) for _ in range(nstops)]
    comp = S.Composition()
    comp.get_mut_or_insert_default(0).insert(S.custom_square(0, 0, 96, 64)).set_props(S.Props(fill=(0.2, 0.3, 0.4, 1.0)))
    comp.get_mut_or_insert_default(1).insert(S.custom_circle(40, 30, 28)).set_props(
        S.Props(fill=S.gradient((3.0, 5.0), (90.0, 60.0), cols)))
    comp.get_mut_or_insert_default(2).insert(S.custom_circle(60, 34, 25)).set_props(
        S.Props(fill=S.gradient((60.0, 34.0), (85.0, 34.0), cols[::-1], radial=True), blend_mode="Multiply", fill_rule="EvenOdd"))
    comp.get_mut_or_insert_default(3).insert(S.custom_circle(48, 32, 20)).set_props(S.Props(clip=2))
    comp.get_mut_or_insert_default(4).insert(S.custom_square(10, 10, 90, 50)).set_props(
        S.Props(fill=S.gradient((10.0, 50.0), (90.0, 10.0), cols), blend_mode="Hue", is_clipped=True))
    comp.get_mut_or_insert_default(5).insert(S.custom_square(20, 20, 70, 60)).set_props(
        S.Props(fill=(0.9, 0.1, 0.1, 0.5), blend_mode="Screen", is_clipped=True))
    comp.get_mut_or_insert_default(6).insert(S.custom_circle(80, 50, 12)).set_props(S.Props(fill=(0.1, 0.8, 0.2, 0.7)))
    o = orc.Oracle(); c = forma_amd.Context(0)
    t = comp.tables(o)
    S.load(o, t); S.load(c, t)
    want = o.render(96, 64, clear=(1, 1, 1, 1))
    got = c.render(96, 64, clear=(1, 1, 1, 1))
    assert np.array_equal(got, want)
    c.render(96, 64, clear=(1, 1, 1, 1))                                # read-back-free frames take the same painter
    assert np.array_equal(c.render(96, 64, clear=(1, 1, 1, 1)), want)
