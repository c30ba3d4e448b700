"""Synthetic workloads named by BASELINE.json `configs` (definitions: SURVEY.md §8d).  All geometry is
built through the product API (forma_amd.api) and flattened on the GPU; nothing here touches oracle/.

  C2  random_cubics      1000 random closed cubic Beziers, solid fill, 1920x1080
  C3  paris_like         30 000-layer stand-in for paris-30k.svg (the asset is not in the reference
                         checkout: /root/reference/.MISSING_LARGE_BLOBS), 3840x2160
  C4  triangles_10m      ~10 M pixel segments, 8192x8192, opaque triangles, layer = index
  C5  spaceship          animated spaceship-like scene at 4K (damage tracking through the buffer-layer cache)
  --  circles            the reference demo's `circles` mode (translucent discs, 1000x1000)
"""
from __future__ import annotations

import numpy as np

from .api import (BLEND_MODES, Color, Composition, Fill, Func, GradientBuilder, GradientType, Order, PathBuilder, Point,
                  Props, Style)


def _solid(color: Color, blend="Over") -> Props:
    return Props(func=Func.Draw(Style(fill=Fill.Solid(color), blend_mode=blend)))


def random_cubics(n=1000, width=1920, height=1080, seed=42) -> Composition:
    rng = np.random.default_rng(seed)
    comp = Composition()
    for i in range(n):
        p = rng.random((4, 2), dtype=np.float32) * np.array([width, height], np.float32)
        c = rng.random(3, dtype=np.float32)
        path = (PathBuilder().move_to(Point(float(p[0, 0]), float(p[0, 1])))
                .cubic_to(Point(float(p[1, 0]), float(p[1, 1])), Point(float(p[2, 0]), float(p[2, 1])), Point(float(p[3, 0]), float(p[3, 1])))
                .build())
        comp.get_mut_or_insert_default(Order(i)).insert(path).set_props(_solid(Color(float(c[0]), float(c[1]), float(c[2]), 1.0)))
    return comp


def _paris_like_shapes(n_layers, width, height, seed):
    """The stand-in's shapes, one dict per layer, in the order (and with exactly the random draws) `paris_like` has always
    used: the API-built composition and its SVG serialisation are two views of the same sequence."""
    rng = np.random.default_rng(seed)
    for _ in range(n_layers):
        k = int(rng.integers(4, 41))
        size = float(np.exp(rng.uniform(np.log(8.0), np.log(400.0))))
        cx, cy = float(rng.uniform(0, width)), float(rng.uniform(0, height))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        rad = size * 0.5 * rng.uniform(0.55, 1.0, k)
        xs = (cx + rad * np.cos(ang)).astype(np.float32); ys = (cy + rad * np.sin(ang)).astype(np.float32)
        curved = rng.random() < 0.3
        cmds = [("M", float(xs[0]), float(ys[0]))]
        for j in range(1, k):
            if curved:
                dx, dy = float(xs[j] - xs[j - 1]), float(ys[j] - ys[j - 1])
                nx, ny = -dy * 0.25, dx * 0.25
                cmds.append(("C", float(np.float32(xs[j - 1] + dx * 0.33 + nx)), float(np.float32(ys[j - 1] + dy * 0.33 + ny)),
                             float(np.float32(xs[j - 1] + dx * 0.66 + nx)), float(np.float32(ys[j - 1] + dy * 0.66 + ny)),
                             float(xs[j]), float(ys[j])))
            else:
                cmds.append(("L", float(xs[j]), float(ys[j])))
        col = rng.random(3, dtype=np.float32)
        alpha = 1.0 if rng.random() < 0.5 else 0.5
        blend = BLEND_MODES[int(rng.integers(1, 16))] if rng.random() < 0.05 else "Over"
        u = rng.random()
        shape = {"cmds": cmds, "color": (float(col[0]), float(col[1]), float(col[2]), alpha), "blend": blend, "gradient": None}
        if u >= 0.90:
            stops = []
            for _ in range(int(rng.integers(2, 4))):
                c2 = rng.random(3, dtype=np.float32)
                stops.append((float(c2[0]), float(c2[1]), float(c2[2]), alpha))
            shape["gradient"] = {"start": (cx - size * 0.5, cy - size * 0.5), "end": (cx + size * 0.5, cy + size * 0.25),
                                 "radial": u >= 0.98, "stops": stops}
        yield shape


def paris_like(n_layers=30000, width=3840, height=2160, seed=30000) -> Composition:
    """Labelled STAND-IN for paris-30k.svg: closed polygons of 4-40 vertices (30 % with cubic edges), bbox
    log-uniform 8-400 px, centres uniform; 90 % solid / 8 % linear / 2 % radial; 5 % non-Over blend;
    alpha in {1.0, 0.5}."""
    comp = Composition()
    for i, sh in enumerate(_paris_like_shapes(n_layers, width, height, seed)):
        b = PathBuilder()
        for c in sh["cmds"]:
            if c[0] == "M":
                b.move_to(Point(c[1], c[2]))
            elif c[0] == "L":
                b.line_to(Point(c[1], c[2]))
            else:
                b.cubic_to(Point(c[1], c[2]), Point(c[3], c[4]), Point(c[5], c[6]))
        g = sh["gradient"]
        if g is None:
            fill = Fill.Solid(Color(*sh["color"]))
        else:
            gb = GradientBuilder(Point(*g["start"]), Point(*g["end"]))
            if g["radial"]:
                gb.type(GradientType.Radial)
            for st in g["stops"]:
                gb.color(Color(*st))
            fill = Fill.Gradient(gb.build())
        comp.get_mut_or_insert_default(Order(i)).insert(b.build()).set_props(Props(func=Func.Draw(Style(fill=fill, blend_mode=sh["blend"]))))
    return comp


_CSS_BLEND = {"Over": "normal", "Multiply": "multiply", "Screen": "screen", "Overlay": "overlay", "Darken": "darken", "Lighten": "lighten",
              "ColorDodge": "color-dodge", "ColorBurn": "color-burn", "HardLight": "hard-light", "SoftLight": "soft-light",
              "Difference": "difference", "Exclusion": "exclusion", "Hue": "hue", "Saturation": "saturation", "Color": "color",
              "Luminosity": "luminosity"}


def paris_like_svg(n_layers=30000, width=3840, height=2160, seed=30000) -> str:
    """The stand-in as SVG TEXT — what `paris-30k.svg` is to the reference's demo (demo/src/demos/svg.rs) — for the loader route
    (`forma_amd.svg.Svg(text, is_text=True).compose(...)`, `bench.py --svg FILE`).  Same geometry to the bit as `paris_like`
    (every coordinate is written with `repr`, which round-trips a float32 exactly): the two compositions rasterize to the same
    pixel-segment streams.  Colours cannot round-trip (SVG colours are 8-bit sRGB, the stand-in's are linear floats), so the
    loaded scene has its own, equally distributed colours; gradients become userSpaceOnUse gradients with evenly spaced stops,
    blend modes `mix-blend-mode`, alpha `fill-opacity` / `stop-opacity`."""
    def hexcol(c):
        v = [max(0, min(255, int(round((x ** (1 / 2.2)) * 255)))) for x in c[:3]]
        return "#%02x%02x%02x" % tuple(v)
    out = ['<svg xmlns="http://www.w3.org/2000/svg" width="%d" height="%d" viewBox="0 0 %d %d">' % (width, height, width, height)]
    for i, sh in enumerate(_paris_like_shapes(n_layers, width, height, seed)):
        d = []
        for c in sh["cmds"]:
            d.append(c[0] + " ".join(repr(v) for v in c[1:]))
        d.append("Z")
        attrs = ""
        g = sh["gradient"]
        if g is None:
            attrs = 'fill="%s" fill-opacity="%s"' % (hexcol(sh["color"]), repr(sh["color"][3]))
        else:
            n = len(g["stops"])
            stops = "".join('<stop offset="%s%%" stop-color="%s" stop-opacity="%s"/>' % (repr(100.0 * j / (n - 1)), hexcol(st), repr(st[3]))
                            for j, st in enumerate(g["stops"]))
            (x1, y1), (x2, y2) = g["start"], g["end"]
            if g["radial"]:
                r = float(np.hypot(x2 - x1, y2 - y1))
                out.append('<radialGradient id="g%d" gradientUnits="userSpaceOnUse" cx="%s" cy="%s" r="%s">%s</radialGradient>'
                           % (i, repr(x1), repr(y1), repr(r), stops))
            else:
                out.append('<linearGradient id="g%d" gradientUnits="userSpaceOnUse" x1="%s" y1="%s" x2="%s" y2="%s">%s</linearGradient>'
                           % (i, repr(x1), repr(y1), repr(x2), repr(y2), stops))
            attrs = 'fill="url(#g%d)"' % i
        if sh["blend"] != "Over":
            attrs += ' style="mix-blend-mode: %s"' % _CSS_BLEND[sh["blend"]]
        out.append('<path d="%s" %s/>' % (" ".join(d), attrs))
    out.append("</svg>")
    return "\n".join(out)


def circles(count=100, width=1000, height=1000, seed=42) -> Composition:
    """The reference demo's `circles` mode (demo/src/demos/circles.rs:22-129): `count` translucent discs (alpha 0.2, radius
    10..50, four rational quadratics of weight sqrt(2)/2 each) at random positions, order = index.  The reference draws
    its randoms from `StdRng::seed_from_u64(42)` (ChaCha12, not reproducible without the `rand` crate); this uses numpy's
    generator — same distribution, not the same discs."""
    rng = np.random.default_rng(seed)
    w = float(np.sqrt(np.float32(2.0)) / np.float32(2.0))
    comp = Composition()
    for order in range(count):
        r_, g_, b_ = (float(v) for v in rng.random(3, dtype=np.float32))
        x, y = float(rng.uniform(0, width)), float(rng.uniform(0, height))
        rad = float(rng.uniform(10.0, 50.0))
        path = (PathBuilder().move_to(Point(x + rad, y)).rat_quad_to(Point(x + rad, y - rad), Point(x, y - rad), w)
                .rat_quad_to(Point(x - rad, y - rad), Point(x - rad, y), w).rat_quad_to(Point(x - rad, y + rad), Point(x, y + rad), w)
                .rat_quad_to(Point(x + rad, y + rad), Point(x + rad, y), w).build())
        comp.get_mut_or_insert_default(Order(order)).clear().insert(path).set_props(_solid(Color(r_, g_, b_, 0.2)))
    return comp


def triangles_10m(width=8192, height=8192, k=19400, seed=4) -> Composition:
    """~10 M pixel segments: K closed random triangles with vertices inside random 256-px boxes."""
    rng = np.random.default_rng(seed)
    comp = Composition()
    for i in range(k):
        ox, oy = rng.uniform(0, width - 256), rng.uniform(0, height - 256)
        p = (rng.random((3, 2)) * 256 + np.array([ox, oy])).astype(np.float32)
        c = rng.random(3, dtype=np.float32)
        path = (PathBuilder().move_to(Point(float(p[0, 0]), float(p[0, 1]))).line_to(Point(float(p[1, 0]), float(p[1, 1])))
                .line_to(Point(float(p[2, 0]), float(p[2, 1]))).build())
        comp.get_mut_or_insert_default(Order(i)).insert(path).set_props(_solid(Color(float(c[0]), float(c[1]), float(c[2]), 1.0)))
    return comp


def spaceship(width=3840, height=2160, enemies=120, stars=400, seed=43):
    """Animated scene in the spirit of the reference's spaceship demo (demo/src/demos/spaceship.rs: `ship_path` :365-417,
    `potatoe_path` :337-362, actors moved by GeomPresTransform [c, s, -s, c, x, y] :198-201), scaled to a 4K canvas:
    a static star field and planets, one ship and `enemies` dented circles that drift and spin.  Returns the
    composition and the orders of the moving layers; `spaceship_transforms(t)` gives their transforms at time t."""
    rng = np.random.default_rng(seed)
    comp = Composition()
    order = 0
    for _ in range(stars):                                  # static background: small discs and a few big planets
        big = rng.random() < 0.03
        r = float(rng.uniform(120, 400) if big else rng.uniform(2, 9))
        x, y = float(rng.uniform(0, width)), float(rng.uniform(0, height))
        w = 0.70710678
        path = (PathBuilder().move_to(Point(x + r, y)).rat_quad_to(Point(x + r, y - r), Point(x, y - r), w)
                .rat_quad_to(Point(x - r, y - r), Point(x - r, y), w).rat_quad_to(Point(x - r, y + r), Point(x, y + r), w)
                .rat_quad_to(Point(x + r, y + r), Point(x + r, y), w).build())
        c = rng.random(3)
        comp.get_mut_or_insert_default(Order(order)).insert(path).set_props(
            _solid(Color(float(c[0]) * 0.5, float(c[1]) * 0.5, float(c[2]), 1.0 if big else 0.8)))
        order += 1
    moving = []
    ship = (PathBuilder().move_to(Point(0, 50)).line_to(Point(40, 50)).line_to(Point(40, 60))
            .cubic_to(Point(47, 56), Point(54, 57), Point(60, 60)).line_to(Point(60, 50)).line_to(Point(80, 50)).line_to(Point(80, 10))
            .cubic_to(Point(67, -3), Point(50, -13), Point(30, -20)).line_to(Point(25, -51)).line_to(Point(30, -52))
            .line_to(Point(30, -70)).line_to(Point(21, -74)).cubic_to(Point(17, -90), Point(9, -102), Point(0, -107))
            .cubic_to(Point(-9, -102), Point(-17, -90), Point(-21, -74)).line_to(Point(-30, -70)).line_to(Point(-30, -52))
            .line_to(Point(-25, -51)).line_to(Point(-30, -20)).cubic_to(Point(-50, -13), Point(-67, -3), Point(-80, 10))
            .line_to(Point(-80, 50)).line_to(Point(-60, 50)).line_to(Point(-60, 60))
            .cubic_to(Point(-54, 57), Point(-47, 56), Point(-40, 60)).line_to(Point(-40, 50)).line_to(Point(0, 50)).build())
    comp.get_mut_or_insert_default(Order(order)).insert(ship).set_props(_solid(Color(0.9, 0.9, 0.2, 1.0)))
    moving.append(order); order += 1
    for _ in range(enemies):
        r = float(rng.uniform(20, 70))
        b = PathBuilder().move_to(Point(r, 0.0))
        for (cx, cy, ex, ey) in ((r, -r, 0.0, -r), (-r, -r, -r, 0.0), (-r, r, 0.0, r), (r, r, r, 0.0)):
            b.rat_quad_to(Point(cx, cy), Point(ex, ey), float(rng.uniform(0.07, 1.4)))
        c = rng.random(3)
        comp.get_mut_or_insert_default(Order(order)).insert(b.build()).set_props(
            _solid(Color(float(c[0]), float(c[1]) * 0.6, float(c[2]) * 0.4, 1.0)))
        moving.append(order); order += 1
    state = dict(pos=rng.uniform([0, 0], [width, height], (len(moving), 2)), vel=rng.uniform(-300, 300, (len(moving), 2)),
                 ang=rng.uniform(0, 6.28, len(moving)), spin=rng.uniform(-2, 2, len(moving)), size=(width, height))
    return comp, moving, state


def spaceship_transforms(state, t: float) -> np.ndarray:
    """[ux, uy, vx, vy, tx, ty] per moving layer at time t (seconds): rotation by angle(t), translation to the wrapped
    position (AffineTransform::to_array order, math/transform.rs:50)."""
    w, h = state["size"]
    pos = state["pos"] + state["vel"] * t
    pos[:, 0] = np.mod(pos[:, 0], w); pos[:, 1] = np.mod(pos[:, 1], h)
    a = state["ang"] + state["spin"] * t
    c, s_ = np.cos(a).astype(np.float32), np.sin(a).astype(np.float32)
    return np.stack([c, s_, -s_, c, pos[:, 0].astype(np.float32), pos[:, 1].astype(np.float32)], axis=1).astype(np.float32)


WORKLOADS = {
    "cubics-1080p": (random_cubics, 1920, 1080),
    "paris-like-30k-4k": (paris_like, 3840, 2160),
    "triangles-10m-8k": (triangles_10m, 8192, 8192),
    "circles-100": (circles, 1000, 1000),                                       # the demo's defaults
    "circles-20k": (lambda: circles(20000), 1000, 1000),                        # ~120 translucent layers deep per tile
}
