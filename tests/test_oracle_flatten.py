"""Oracle flattener vs the reference's path.rs unit tests (forma/src/path.rs:1023-1627).
Expected point counts / endpoints / contour flags are the literals of those tests."""
import numpy as np

from oracle import oracle as orc


def flat(prim):
    return orc.Oracle().flatten_primitives(prim)


MAX_ERROR = 1.0 / 16.0                     # path.rs:40, consts::PIXEL_WIDTH = 16


def _lerp(t, a, b):                        # path.rs:44-46
    return a + t * (b - a)


def _eval_quad(t, c):                      # the tests' own helper, path.rs:948-961
    return (_lerp(t, _lerp(t, c[0][0], c[1][0]), _lerp(t, c[1][0], c[2][0])),
            _lerp(t, _lerp(t, c[0][1], c[1][1]), _lerp(t, c[1][1], c[2][1])))


def _min_dist(p, x, y):                    # path.rs:933-946: distance of p to the LINES through consecutive points
    d10x, d10y = x[:-1] - p[0], y[:-1] - p[1]
    d21x, d21y = x[1:] - x[:-1], y[1:] - y[:-1]
    return float(np.min(np.abs(d21x * d10y - d10x * d21y) / np.hypot(d21x, d21y)))


def test_quads():  # :1023-1073
    c0, c1 = ((2, 0), (0, 1), (10, 1)), ((10, 1), (20, 1), (18, 0))
    x, y, _ = flat(orc.Primitives().push_quad(*c0).push_quad(*c1))
    assert len(x) == 9 and (x[0], y[0]) == (2.0, 0.0) and (x[8], y[8]) == (18.0, 0.0)
    assert np.hypot(x[3] - x[5], y[3] - y[5]) > 10.0
    # interior points (:1055-1074): every curve sample lies within MAX_ERROR of the flattened polyline
    for c in (c0, c1):
        assert max(_min_dist(_eval_quad(i / 50.0, c), x, y) for i in range(51)) < MAX_ERROR


def test_interior_points_stay_within_max_error_of_random_quads():
    """The property `quads` (:1055-1074) checks, on 200 seeded random quadratic Béziers up to 400 px long: pins the interior
    points of the flattener against the curve itself rather than against another restatement.  The subdivision rule
    (path.rs:252-445) targets MAX_ERROR but does not guarantee it for every curve (worst of this sample: 0.14 px), so the
    bound here is 4 x MAX_ERROR = a quarter of a pixel; the reference's own case above keeps its literal bound."""
    rng = np.random.default_rng(7)
    for _ in range(200):
        c = [tuple(float(v) for v in rng.uniform(-200, 200, 2)) for _ in range(3)]
        x, y, _ = flat(orc.Primitives().push_quad(*c))
        assert (x[0], y[0]) == (np.float32(c[0][0]), np.float32(c[0][1])) and (x[-1], y[-1]) == (np.float32(c[2][0]), np.float32(c[2][1]))
        if len(x) > 1 and np.all(np.hypot(np.diff(x), np.diff(y)) > 0):
            assert max(_min_dist(_eval_quad(i / 50.0, c), x, y) for i in range(51)) < 4 * MAX_ERROR


def test_two_splines():  # :1075-1095
    x, y, _ = flat(orc.Primitives().push_quad((0, 0), (1, 2), (2, 0)).push_quad((3, 0), (4, 4), (5, 0)))
    assert len(x) == 11
    assert [(x[i], y[i]) for i in (0, 4, 5, 10)] == [(0, 0), (2, 0), (3, 0), (5, 0)]


def test_collinear_quad():  # :1097-1109
    x, y, _ = flat(orc.Primitives().push_quad((0, 0), (2, 0.0001), (1, 0)))
    assert len(x) == 3 and abs(x[1] - 1.25) < 0.01 and abs(y[1]) < 0.01


def test_overlapping_control_point_quad():  # :1111-1127
    x, y, _ = flat(orc.Primitives().push_quad((0, 0), (0, 0), (1, 1)).push_quad((1, 1), (1, 1), (1, 1)).push_quad((1, 1), (2, 2), (2, 2)))
    assert len(x) == 2 and abs(x[1] - 2) < 0.01 and abs(y[1] - 2) < 0.01 and abs(x[0]) < 0.01


def test_rat_quad():  # :1129-1170
    w = 10.0
    x, y, _ = flat(orc.Primitives().push_quad((0, 0, 1), (1 * w, 2 * w, w), (2, 0, 1)))
    assert len(x) == 5 and abs(x[2] - 1.0) <= 0.001
    d = np.hypot(np.diff(x), np.diff(y))
    assert d[0] > 1.5 and d[1] < 0.2 and d[2] < 0.2 and d[3] > 1.5


def test_lines_and_quads():  # :1172-1201
    p = (orc.Primitives().push_line((-1, -2), (0, 0)).push_quad((0, 0), (1, 2), (2, 0)).push_line((2, 0), (3, -2))
         .push_line((3, -2), (4, 2)).push_line((4, 2), (5, -4)).push_line((5, -4), (6, 0)).push_quad((6, 0), (7, 4), (8, 0))
         .push_line((8, 0), (9, -4)))
    x, y, _ = flat(p)
    assert len(x) == 12
    assert [(x[i], y[i]) for i in (0, 4, 5, 6, 11)] == [(-1, -2), (3, -2), (4, 2), (5, -4), (9, -4)]


def test_cubic():  # :1203-1226
    x, y, _ = flat(orc.Primitives().push_cubic((0, 0), (10, 6), (-2, 6), (8, 0)))
    assert len(x) == 10
    assert x[2] > x[7] and x[3] > x[6] and x[4] > x[5]
    assert all(y[i] < y[i + 1] for i in range(4)) and all(y[i] > y[i + 1] for i in range(5, 9))


def test_rat_cubic_high_low():  # :1228-1282
    w = 10.0
    x, _, _ = flat(orc.Primitives().push_cubic((0, 0, 1), (5 * w, 3 * w, w), (-1 * w, 3 * w, w), (4, 0, 1)))
    assert len(x) == 45
    w = 0.5
    x, _, _ = flat(orc.Primitives().push_cubic((0, 0, 1), (5 * w, 3 * w, w), (-1 * w, 3 * w, w), (4, 0, 1)))
    assert len(x) == 7


def test_collinear_cubic():  # :1284-1309
    x, y, _ = flat(orc.Primitives().push_cubic((1, 0), (0, 0), (3, 0), (2, 0)))
    assert len(x) == 5 and (x[0], y[0]) == (1, 0) and (x[4], y[4]) == (2, 0)
    assert 0.5 < x[1] < 1.0 and 1.0 < x[2] < 2.0 and 2.0 < x[3] < 2.5 and not y.any()


def test_overlapping_control_point_cubic_line():  # :1311-1341
    p = (orc.Primitives().push_cubic((0, 0), (0, 0), (1, 1), (1, 1)).push_cubic((1, 1), (1, 1), (1, 1), (1, 1))
         .push_cubic((1, 1), (1, 1), (2, 2), (2, 2)))
    x, y, _ = flat(p)
    assert len(x) == 9 and np.all(np.diff(x) > 0) and np.array_equal(x, y)
    assert abs(x[0]) < 0.01 and abs(x[8] - 2) < 0.01


def test_ring():  # :1343-1376
    p = (orc.Primitives().push_cubic((0, 2), (2, 2), (2, 2), (2, 0)).push_cubic((2, 0), (2, -2), (2, -2), (0, -2))
         .push_cubic((0, -2), (-2, -2), (-2, -2), (-2, 0)).push_cubic((-2, 0), (-2, 2), (-2, 2), (0, 2)).push_contour()
         .push_cubic((0, 1), (-1, 1), (-1, 1), (-1, 0)).push_cubic((-1, 0), (-1, -1), (-1, -1), (0, -1))
         .push_cubic((0, -1), (1, -1), (1, -1), (1, 0)).push_cubic((1, 0), (1, 1), (1, 1), (0, 1)))
    _, _, nc = flat(p)
    assert len(nc) == 30 and nc.sum() == 2 and nc[16] and nc[29]


def test_ring_overlapping_start():  # :1378-1411
    p = (orc.Primitives().push_cubic((0, 1), (-1, 1), (-1, 1), (-1, 0)).push_cubic((-1, 0), (-1, -1), (-1, -1), (0, -1))
         .push_cubic((0, -1), (1, -1), (1, -1), (1, 0)).push_cubic((1, 0), (1, 1), (1, 1), (0, 1)).push_contour()
         .push_cubic((0, 1), (1, 1), (1, 1), (1, 2)).push_cubic((1, 2), (1, 3), (1, 3), (0, 3))
         .push_cubic((0, 3), (-1, 3), (-1, 3), (-1, 2)).push_cubic((-1, 2), (-1, 1), (-1, 1), (0, 1)))
    _, _, nc = flat(p)
    assert len(nc) == 26 and nc.sum() == 2 and nc[12] and nc[25]


def test_circle():  # :1413-1488
    r = 50.0
    w = float(np.sqrt(np.float32(2.0)) / np.float32(2.0))
    f = lambda v: float(np.float32(v))
    p = (orc.Primitives().push_quad((r, 0, 1), (0, 0, w), (0, r, 1))
         .push_quad((0, r, 1), (0, f(np.float32(2.0 * r) * np.float32(w)), w), (r, 2 * r, 1))
         .push_quad((r, 2 * r, 1), (f(np.float32(2.0 * r) * np.float32(w)), f(np.float32(2.0 * r) * np.float32(w)), w), (2 * r, r, 1))
         .push_quad((2 * r, r, 1), (f(np.float32(2.0 * r) * np.float32(w)), 0, w), (r, 0, 1)))
    x, y, _ = flat(p)
    assert len(x) == 66
    assert np.hypot(np.diff(x), np.diff(y)).max() < 5.0


def _circle_path(radius, tr=0.0):
    w = float(np.sqrt(np.float32(2.0)) / np.float32(2.0))
    return (orc.Path().move_to(radius + tr, 0).rat_quad_to(radius + tr, -radius, tr, -radius, w)
            .rat_quad_to(-radius + tr, -radius, -radius + tr, 0, w).rat_quad_to(-radius + tr, radius, tr, radius, w)
            .rat_quad_to(radius + tr, radius, radius + tr, 0, w).build())


def test_transform_path():  # :1490-1557
    o = orc.Oracle()
    radius = 10.0
    x, y, nc = o.flatten(_circle_path(radius))
    orig_len = len(x)
    assert not nc[:-1].any() and nc[-1]
    assert np.all(np.abs(np.hypot(x, y) - radius) <= 0.1)
    path = _circle_path(radius); path.affine = (1.0, 0.0, 0.0, 1.0, 5.0, 20.0)   # GeomPresTransform: translation
    x, y, _ = o.flatten(path)
    assert np.all(np.abs(np.hypot(x - 5.0, y - 20.0) - radius) <= 0.1)
    s = 2.0  # scaling exceeds GeomPresTransform -> control points are transformed (path.rs:733-765)
    x, y, _ = o.flatten(_circle_path(radius).transform9([s, 0, 0, 0, s, 0, 0, 0, 1]))
    assert np.all(np.abs(np.hypot(x, y) - s * radius) <= 0.1) and len(x) > orig_len


def test_perspective_transform_path():  # :1559-1626
    o = orc.Oracle()
    radius = 10.0
    x, y, _ = o.flatten(_circle_path(radius, 1000.0).transform9([1, 0, 0, 0, 1, 0, 0.001, 0, 1]))
    pts = np.stack([x, y], 1)[:-1]
    half = len(pts) // 2
    d = [np.hypot(*(pts[i] - pts[(i + half) % len(pts)])) for i in range(half)]
    assert abs(min(d) - radius / 2.0) <= 0.2 and abs(max(d) - radius) <= 0.2
