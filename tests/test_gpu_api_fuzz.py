"""The public API under random edits: two compositions — tests/ref_api.py (the reference's Composition / Layer / Renderer
bookkeeping over the oracle) and the product mirror forma_amd.api over libforma_hip.so — receive the SAME random sequence
of operations (insert / replace / remove layers, paths added and cleared, props, transforms, enable / disable, compact_geom,
new caches, crops) and render every step into buffers that are carried from frame to frame.  The buffers must stay equal
(within one code value), with and without a buffer-layer cache — which needs the same `is_unchanged` bits, the same geometry
ids after reuse and compaction, the same damage."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _path(api, rng, w, h):
    P = api.Point
    k = int(rng.integers(3, 7))
    cx, cy = float(rng.uniform(-0.2 * w, 1.2 * w)), float(rng.uniform(-0.2 * h, 1.2 * h))
    r = float(np.exp(rng.uniform(np.log(3.0), np.log(0.5 * max(w, h)))))
    ang = np.sort(rng.random(k) * 2 * np.pi)
    pts = [(float(np.float32(cx + r * np.cos(a))), float(np.float32(cy + r * np.sin(a)))) for a in ang]
    return pts, [int(v) for v in rng.integers(0, 3, k)], [float(v) for v in rng.normal(size=4 * k) * r * 0.25]


def _build(api, spec):
    P = api.Point
    pts, modes, jit = spec
    b = api.PathBuilder().move_to(P(*pts[0]))
    for j in range(1, len(pts)):
        x0, y0 = pts[j - 1]; x1, y1 = pts[j]
        if modes[j] == 0:
            b.line_to(P(x1, y1))
        elif modes[j] == 1:
            b.quad_to(P((x0 + x1) / 2 + jit[4 * j], (y0 + y1) / 2 + jit[4 * j + 1]), P(x1, y1))
        else:
            b.cubic_to(P(x0 + jit[4 * j], y0 + jit[4 * j + 1]), P(x1 + jit[4 * j + 2], y1 + jit[4 * j + 3]), P(x1, y1))
    b.line_to(P(*pts[0]))
    return b.build()


def _props(api, spec):
    kind, col, rule, blend, clipped, n = spec
    if kind == "clip":
        return api.Props(fill_rule=rule, func=api.Func.Clip(n))
    c = api.Color(*col)
    if kind == "grad":
        g = api.GradientBuilder(api.Point(col[0] * 100, col[1] * 100), api.Point(col[2] * 300 + 20, col[3] * 200 + 20))
        g.color(api.Color(col[3], col[0], col[1], 1.0)); g.color(c); g.color(api.Color(col[1], col[2], col[3], col[0]))
        fill = api.Fill.Gradient(g.build())
    else:
        fill = api.Fill.Solid(c)
    style = api.Style(is_clipped=clipped, fill=fill, blend_mode=blend)
    return api.Props(fill_rule=rule, func=api.Func.Draw(style))


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("FORMA_TEST_FUZZ_SEEDS", "6")))))
def test_random_edits_through_the_public_api(seed):
    import ref_api
    from forma_amd import api as product
    rng = np.random.default_rng(31000 + seed)
    w, h = int(rng.integers(30, 330)), int(rng.integers(30, 230))
    backends = [ref_api, product]
    comps = [a.Composition() for a in backends]
    rends = [a.Renderer() for a in backends]
    caches = [r.create_buffer_layer_cache() for r in rends]
    bufs_c = [np.zeros(w * h * 4, np.uint8) for _ in backends]         # carried with the cache
    bufs_p = [np.zeros(w * h * 4, np.uint8) for _ in backends]         # plain
    blends = ["Over", "Over", "Over", "Multiply", "Screen", "Difference", "Hue", "Luminosity"]
    live = []                                                          # orders in use

    def both(f):
        return [f(i, backends[i]) for i in range(2)]

    for step in range(int(os.environ.get("FORMA_TEST_FUZZ_STEPS", "28"))):
        for _ in range(int(rng.integers(1, 5))):
            op = int(rng.integers(0, 10))
            if op <= 2 or not live:                                    # a new or replaced layer
                order = int(rng.integers(0, 60))
                spec = _path(backends[0], rng, w, h)
                u = rng.random()
                col = tuple(float(v) for v in rng.random(4))
                if u < 0.7: col = col[:3] + ((1.0,) if rng.random() < 0.5 else (col[3],))
                pspec = ("clip" if u > 0.93 else ("grad" if u > 0.8 else "solid"), col, "EvenOdd" if rng.random() < 0.3 else "NonZero",
                         blends[int(rng.integers(0, len(blends)))], bool(rng.random() < 0.15), int(rng.integers(1, 4)))

                def mk(i, a):
                    L = comps[i].create_layer()
                    L.insert(_build(a, spec)).set_props(_props(a, pspec))
                    comps[i].insert(a.Order.new(order), L)
                both(mk)
                if order not in live: live.append(order)
            else:
                order = live[int(rng.integers(0, len(live)))]
                if op == 3:
                    both(lambda i, a: comps[i].remove(a.Order.new(order))); live.remove(order)
                elif op == 4:
                    t = [1.0, 0.0, 0.0, 1.0, float(rng.uniform(-40, 40)), float(rng.uniform(-30, 30))]
                    both(lambda i, a: comps[i].get_mut(a.Order.new(order)).set_transform(a.GeomPresTransform.try_from(t)))
                elif op == 5:
                    v = bool(rng.random() < 0.5)
                    both(lambda i, a: comps[i].get_mut(a.Order.new(order)).set_is_enabled(v))
                elif op == 6:
                    spec = _path(backends[0], rng, w, h)
                    both(lambda i, a: comps[i].get_mut(a.Order.new(order)).insert(_build(a, spec)))
                elif op == 7:
                    spec = _path(backends[0], rng, w, h)
                    both(lambda i, a: comps[i].get_mut(a.Order.new(order)).clear().insert(_build(a, spec)))
                elif op == 8:
                    col = tuple(float(v) for v in rng.random(3)) + (1.0,)
                    pspec = ("solid", col, "NonZero", "Over", False, 1)
                    both(lambda i, a: comps[i].get_mut(a.Order.new(order)).set_props(_props(a, pspec)))
                else:
                    both(lambda i, a: comps[i].compact_geom())
        if step % 9 == 8:                                              # a fresh cache: everything repaints once
            caches = [r.create_buffer_layer_cache() for r in rends]
        clear = [float(v) for v in rng.random(3)] + [1.0]
        if step % 5 != 4: clear = [0.3, 0.6, 0.9, 1.0]                 # (mostly the same clear colour: damage stays partial)
        crop = None
        if step % 7 == 6:
            x0, y0 = int(rng.integers(0, w - 1)), int(rng.integers(0, h - 1))
            crop = (x0, int(rng.integers(x0 + 1, w + 1)), y0, int(rng.integers(y0 + 1, h + 1)))
        for i, a in enumerate(backends):
            layout = a.LinearLayout(w, w * 4, h)
            rect = None if crop is None else a.Rect(range(crop[0], crop[1]), range(crop[2], crop[3]))
            rends[i].render(comps[i], a.BufferBuilder(bufs_c[i], layout).layer_cache(caches[i]).build(), a.RGBA, a.Color(*clear), rect)
            rends[i].render(comps[i], a.BufferBuilder(bufs_p[i], layout).build(), a.RGBA, a.Color(*clear), rect)
        assert len(comps[0]) == len(comps[1]), (seed, step)
        for name, bb in (("cache", bufs_c), ("plain", bufs_p)):
            d = np.abs(bb[0].astype(int) - bb[1].astype(int))
            assert d.max() <= 1, (seed, step, name, int((d > 1).sum()))
