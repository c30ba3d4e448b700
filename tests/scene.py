"""Test-side scene builder: an independent, minimal restatement of forma's public scene API
(PathBuilder / Composition / Layer / Props, reference forma/src/lib.rs:117-154) that produces the
flat tables both the oracle and the HIP backend consume (include/forma_hip.h).

It flattens curves with the ORACLE (tests only) so that oracle-vs-HIP parity tests of stages 2-4
start from bit-identical geometry; the product's own flattener/host mirror is tested separately
against these tables.
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from oracle import oracle as orc

NONE = 0xFFFFFFFF

BLEND_MODES = ["Over", "Multiply", "Screen", "Overlay", "Darken", "Lighten", "ColorDodge", "ColorBurn",
               "HardLight", "SoftLight", "Difference", "Exclusion", "Hue", "Saturation", "Color", "Luminosity"]

RGBA = (0, 1, 2, 3)
BGRA = (2, 1, 0, 3)
RGB0 = (0, 1, 2, 4)
BGR0 = (2, 1, 0, 4)
RGB1 = (0, 1, 2, 5)
BGR1 = (2, 1, 0, 5)


def f32bits(v: float) -> int:
    return struct.unpack("<I", struct.pack("<f", v))[0]


@dataclass
class Gradient:
    start: Tuple[float, float]
    end: Tuple[float, float]
    stops: List[Tuple[Tuple[float, float, float, float], float]]
    radial: bool = False


def gradient(start, end, colors, radial=False, stops=None) -> Gradient:
    """GradientBuilder::build (reference styling.rs:112-133): unset stops are i / (n - 1)."""
    n = len(colors)
    inc = np.float32(1.0) / np.float32(n - 1)
    st = []
    for i, c in enumerate(colors):
        s = float(np.float32(i) * inc) if stops is None or stops[i] is None else stops[i]
        st.append((tuple(c), s))
    return Gradient(tuple(start), tuple(end), st, radial)


@dataclass
class Image:
    texels: np.ndarray  # (h*w, 4) uint16 bias-shifted halves
    width: int
    height: int

    @staticmethod
    def from_srgba(data: Sequence[Sequence[int]], width: int, height: int) -> "Image":
        L = orc.lib()
        out = np.zeros((len(data), 4), np.uint16)
        for i, c in enumerate(data):
            lin = [L.oracle_srgb_to_linear(c[0]), L.oracle_srgb_to_linear(c[1]), L.oracle_srgb_to_linear(c[2]),
                   float(np.float32(c[3]) * (np.float32(1.0) / np.float32(255.0)))]
            out[i] = [L.oracle_f32_to_f16(v) for v in lin]
        assert len(data) == width * height
        return Image(out, width, height)


@dataclass
class Texture:
    transform: Tuple[float, float, float, float, float, float]  # ux uy vx vy tx ty
    image: Image


@dataclass
class Props:
    fill_rule: str = "NonZero"
    clip: Optional[int] = None          # Func::Clip(n)
    fill: object = (0.0, 0.0, 0.0, 1.0)  # solid rgba | Gradient | Texture
    blend_mode: str = "Over"
    is_clipped: bool = False


def solid(color) -> Props:
    return Props(fill=tuple(color))


def encode_props(p: Props, images: List[Image]) -> List[int]:
    h = 0
    if p.fill_rule == "EvenOdd":
        h |= 1 << 6
    if p.clip is not None:
        h |= 1 << 8
        return [h, p.clip]
    h |= BLEND_MODES.index(p.blend_mode)
    if p.is_clipped:
        h |= 1 << 7
    if isinstance(p.fill, Gradient):
        g = p.fill
        h |= (2 if g.radial else 1) << 4
        h |= len(g.stops) << 16
        w = [h, 0] + [f32bits(v) for v in (g.start[0], g.start[1], g.end[0], g.end[1])]
        for c, s in g.stops:
            w += [f32bits(v) for v in (*c, s)]
        return w
    if isinstance(p.fill, Texture):
        h |= 3 << 4
        idx = None
        for i, im in enumerate(images):
            if im is p.fill.image:
                idx = i
        if idx is None:
            images.append(p.fill.image)
            idx = len(images) - 1
        return [h, 0] + [f32bits(v) for v in p.fill.transform] + [idx]
    return [h, 0] + [f32bits(v) for v in p.fill]


@dataclass
class Layer:
    paths: List[orc.Path] = field(default_factory=list)
    props: Props = field(default_factory=Props)
    transform: Optional[Tuple[float, ...]] = None   # ux uy vx vy tx ty
    enabled: bool = True
    unchanged: bool = False

    def insert(self, path: orc.Path) -> "Layer":
        self.paths.append(path); return self

    def set_props(self, props: Props) -> "Layer":
        self.props = props; return self

    def set_transform(self, t) -> "Layer":
        self.transform = tuple(t); return self


class Composition:
    def __init__(self, insertion_order: bool = False):
        self.layers = {}
        # the reference appends geometry to the SegmentBuffer when a path is inserted (segment.rs:180-198), i.e. in
        # insertion order; sorted order is merely the common case (scenes built back to front)
        self.insertion_order = insertion_order

    def get_mut_or_insert_default(self, order: int) -> Layer:
        return self.layers.setdefault(order, Layer())

    def tables(self, oracle: orc.Oracle):
        """Flatten every path (oracle) and build x, y, line_slot, geoms, styles, images."""
        xs, ys, ids = [], [], []
        orders = list(self.layers) if self.insertion_order else sorted(self.layers)
        geoms = np.zeros(len(orders), orc.GEOM_DTYPE)
        n_orders = (max(orders) + 1) if orders else 0
        offsets = np.full(n_orders, NONE, np.uint32)
        unchanged = np.zeros(n_orders, np.uint8)
        words: List[int] = []
        images: List[Image] = []
        for slot, order in enumerate(orders):
            layer = self.layers[order]
            geoms[slot]["order"] = order if layer.enabled else NONE
            if layer.transform is not None:
                geoms[slot]["flags"] = 1
                geoms[slot]["xf"] = layer.transform
            offsets[order] = len(words)
            words += encode_props(layer.props, images)
            unchanged[order] = 1 if layer.unchanged else 0
            for path in layer.paths:
                x, y, nc = oracle.flatten(path)
                if len(x) == 0:
                    continue
                # Path::push_segments_to + SegmentBuffer::push_path (path.rs:677-723, segment.rs:180-198)
                xs.append(x); ys.append(y)
                pid = np.where(nc != 0, NONE, slot).astype(np.uint32)
                pid[-1] = NONE
                ids.append(pid)
        if xs:
            x = np.concatenate(xs); y = np.concatenate(ys); lid = np.concatenate(ids)
        else:
            x = np.zeros(0, np.float32); y = np.zeros(0, np.float32); lid = np.zeros(0, np.uint32)
        line_slot = lid[: max(len(x) - 1, 0)]
        img_tab = np.zeros(len(images), orc.IMAGE_DTYPE)
        tex = []
        off = 0
        for i, im in enumerate(images):
            img_tab[i] = (off, im.width, im.height)
            tex.append(im.texels); off += len(im.texels)
        texels = np.concatenate(tex) if tex else np.zeros((0, 4), np.uint16)
        return dict(x=x, y=y, line_slot=line_slot, geoms=geoms, style_offsets=offsets,
                    style_words=np.asarray(words, np.uint32), unchanged=unchanged, images=img_tab, texels=texels)


def load(backend, t):
    """Upload tables to an oracle.Oracle or a forma_amd Renderer context (same method names)."""
    backend.set_geometry(t["x"], t["y"], t["line_slot"])
    backend.set_geoms(t["geoms"])
    backend.set_styles(t["style_offsets"], t["style_words"], t["unchanged"])
    backend.set_images(t["images"], t["texels"])


# ---- e2e scene helpers (reference e2e-tests/tests/tests.rs:41-217, test_env.rs:36-38) ----------
WIDTH = 64.0
HEIGHT = 64.0
PADDING = 8.0


def P():
    return orc.Path()


def triangle():
    return P().move_to(PADDING, PADDING).line_to(WIDTH - PADDING, PADDING).line_to(WIDTH - PADDING, HEIGHT - PADDING).build()


def custom_square(xmin, ymin, xmax, ymax):
    return P().move_to(xmin, ymin).line_to(xmin, ymax).line_to(xmax, ymax).line_to(xmax, ymin).build()


def square():
    return custom_square(PADDING, PADDING, WIDTH - PADDING, HEIGHT - PADDING)


def inner_square():
    return custom_square(PADDING * 2, PADDING * 2, WIDTH - PADDING * 2, HEIGHT - PADDING * 2)


def custom_circle(x, y, radius):
    w = float(np.sqrt(np.float32(2.0)) / np.float32(2.0))
    return (P().move_to(x + radius, y)
            .rat_quad_to(x + radius, y - radius, x, y - radius, w)
            .rat_quad_to(x - radius, y - radius, x - radius, y, w)
            .rat_quad_to(x - radius, y + radius, x, y + radius, w)
            .rat_quad_to(x + radius, y + radius, x + radius, y, w)
            .build())


def circle():
    return custom_circle(WIDTH * 0.5, HEIGHT * 0.5, WIDTH * 0.5 - PADDING)


def inner_circle():
    return custom_circle(WIDTH * 0.5, HEIGHT * 0.5, WIDTH * 0.5 - PADDING * 2)


RAINBOW = [(1.00, 0.00, 0.00, 1.0), (1.00, 0.32, 0.00, 1.0), (0.63, 0.73, 0.02, 1.0), (0.08, 0.72, 0.07, 1.0),
           (0.05, 0.70, 0.69, 1.0), (0.03, 0.58, 0.76, 1.0), (0.01, 0.21, 0.85, 1.0), (0.11, 0.01, 0.89, 1.0),
           (0.49, 0.00, 0.94, 1.0), (0.96, 0.00, 0.69, 1.0), (1.00, 0.00, 0.00, 1.0)]


def vertical_rainbow():
    return gradient((PADDING, 0.0), (WIDTH - PADDING, 0.0), RAINBOW)


def horizontal_rainbow():
    return gradient((0.0, PADDING), (0.0, WIDTH - PADDING), RAINBOW)


def e2e_scenes():
    """name -> Composition for every CPU golden of the reference (e2e-tests/tests/tests.rs:219-742)."""
    out = {}

    c = Composition()
    c.get_mut_or_insert_default(1).insert(triangle()).set_props(
        Props(fill=gradient((PADDING, 0.0), (WIDTH - PADDING, 0.0), [(0, 0, 1, 1), (1, 1, 1, 1), (1, 0, 0, 1)])))
    out["linear_gradient"] = c

    c = Composition()
    c.get_mut_or_insert_default(1).insert(circle()).set_props(
        Props(fill=gradient((WIDTH * 0.5, HEIGHT * 0.5), (WIDTH - PADDING * 2.0, HEIGHT * 0.5),
                            [(0, 0, 1, 1), (1, 1, 1, 1), (1, 0, 0, 1)], radial=True)))
    out["radial_gradient"] = c

    for color, name in [((0, 0, 1, 1), "blue"), ((0, 0, 0.5, 1), "dark_blue"), ((1, 0, 0, 1), "red"),
                        ((0.5, 0, 0, 1), "dark_red"), ((0, 1, 0, 1), "green"), ((0, 0.5, 0, 1), "dark_green"),
                        ((0, 0, 0, 0.5), "transparent_black")]:
        c = Composition()
        c.get_mut_or_insert_default(1).insert(square()).set_props(solid(color))
        out["solid_color__" + name] = c

    c = Composition()
    c.get_mut_or_insert_default(1).insert(custom_square(PADDING, PADDING, PADDING + 1.0, PADDING + 1.0)).set_props(solid((0, 0, 0, 1)))
    out["pixel"] = c

    c = Composition()
    layer = c.get_mut_or_insert_default(0).set_props(solid((0, 0, 0, 1)))
    step = float(np.float32(2.0) + np.float32(1.0) / np.float32(32.0))
    for xi in range(32):
        for yi in range(32):
            x0 = float(np.float32(xi) * np.float32(step)); y0 = float(np.float32(yi) * np.float32(step))
            layer.insert(custom_square(x0, y0, x0 + 1.0, y0 + 1.0))
    out["covers"] = c

    c = Composition()
    image = Image.from_srgba([[0, 0, 0, 255], [255, 0, 0, 255], [0, 255, 0, 255], [255, 255, 0, 255], [0, 0, 255, 255],
                              [255, 0, 255, 255], [0, 255, 255, 255], [255, 255, 255, 255], [0, 0, 0, 255]], 3, 3)
    order = 0
    for xi in range(8):
        for yi in range(8):
            x0 = xi * 8.0; y0 = yi * 8.0
            tx = -x0 - 2.0 + xi * 0.25; ty = -y0 - 2.0 + yi * 0.25
            c.get_mut_or_insert_default(order).insert(custom_square(x0, y0, x0 + 7.0, y0 + 7.0)).set_props(
                Props(fill_rule="EvenOdd", fill=Texture((1.0, 0.0, 0.0, 1.0, tx, ty), image)))
            order += 1
    out["texture"] = c

    for bm in BLEND_MODES:
        c = Composition()
        c.get_mut_or_insert_default(0).insert(square()).set_props(Props(fill=horizontal_rainbow()))
        c.get_mut_or_insert_default(1).insert(triangle()).set_props(Props(fill=vertical_rainbow(), blend_mode=bm))
        out["blend_modes__" + bm] = c

    for fr in ["EvenOdd", "NonZero"]:
        c = Composition()
        path = (P().move_to(PADDING, PADDING).line_to(WIDTH / 2 + PADDING, HEIGHT / 2 + PADDING)
                .line_to(WIDTH / 2 - PADDING, HEIGHT / 2 + PADDING).line_to(WIDTH - PADDING, PADDING)
                .line_to(WIDTH - PADDING, HEIGHT - PADDING).line_to(PADDING, HEIGHT - PADDING).build())
        c.get_mut_or_insert_default(0).insert(path).set_props(Props(fill_rule=fr, fill=(0.0, 0.0, 0.0, 0.8)))
        out["fill_rules__" + fr] = c

    c = Composition()
    c.get_mut_or_insert_default(0).insert(square()).set_props(solid((0, 0, 0, 0.7)))
    c.get_mut_or_insert_default(1).insert(triangle()).set_props(Props(clip=4))
    c.get_mut_or_insert_default(2).insert(square()).set_props(Props(fill=(0.5, 0.5, 1.0, 0.7), is_clipped=True))
    c.get_mut_or_insert_default(4).insert(circle()).set_props(Props(fill=(1.0, 0.5, 0.5, 0.7)))
    c.get_mut_or_insert_default(5).insert(inner_square()).set_props(Props(fill=(0.5, 0.5, 1.0, 0.6), is_clipped=True))
    c.get_mut_or_insert_default(6).insert(inner_circle()).set_props(Props(fill=(0.5, 1.0, 0.5, 0.6), is_clipped=True))
    out["clipping"] = c

    c = Composition()
    c.get_mut_or_insert_default(0).insert(square()).set_props(solid((0, 0, 0, 0.7)))
    c.get_mut_or_insert_default(1).insert(inner_circle()).set_props(Props(clip=1))
    c.get_mut_or_insert_default(2).insert(triangle()).set_props(Props(fill=(0.5, 0.5, 1.0, 0.7), is_clipped=True))
    out["clipping2"] = c
    return out


# ---- synthetic workloads -----------------------------------------------------------
def random_cubics(n=1000, width=1920, height=1080, seed=42, alpha=1.0):
    """C2: n layers, one closed random cubic each, solid fill, NonZero, Over."""
    rng = np.random.default_rng(seed)
    c = Composition()
    for i in range(n):
        p = rng.random((4, 2), dtype=np.float32) * np.array([width, height], np.float32)
        col = rng.random(3, dtype=np.float32)
        path = P().move_to(*map(float, p[0])).cubic_to(*map(float, p[1]), *map(float, p[2]), *map(float, p[3])).build()
        c.get_mut_or_insert_default(i).insert(path).set_props(solid((float(col[0]), float(col[1]), float(col[2]), alpha)))
    return c


def random_mixed(n=300, width=512, height=384, seed=7):
    """A small stress scene touching every feature: fills, fill rules, blend modes, clips, transforms,
    shapes crossing every canvas edge."""
    rng = np.random.default_rng(seed)
    c = Composition()
    img = Image.from_srgba([[int(v) for v in rng.integers(0, 256, 4)] for _ in range(16)], 4, 4)
    order = 0
    for i in range(n):
        k = int(rng.integers(3, 9))
        cx, cy = rng.random(2) * np.array([width * 1.4, height * 1.4]) - np.array([width * 0.2, height * 0.2])
        r = float(np.exp(rng.uniform(np.log(4.0), np.log(160.0))))
        path = P()
        ang = np.sort(rng.random(k) * 2 * np.pi)
        pts = [(float(np.float32(cx + r * np.cos(a))), float(np.float32(cy + r * np.sin(a)))) for a in ang]
        path.move_to(*pts[0])
        for j in range(1, k):
            mode = rng.integers(0, 3)
            if mode == 0:
                path.line_to(*pts[j])
            elif mode == 1:
                mx = float(np.float32((pts[j - 1][0] + pts[j][0]) / 2 + rng.normal() * r * 0.3))
                my = float(np.float32((pts[j - 1][1] + pts[j][1]) / 2 + rng.normal() * r * 0.3))
                path.quad_to(mx, my, *pts[j])
            else:
                a = (float(np.float32(pts[j - 1][0] + rng.normal() * r * 0.3)), float(np.float32(pts[j - 1][1] + rng.normal() * r * 0.3)))
                b = (float(np.float32(pts[j][0] + rng.normal() * r * 0.3)), float(np.float32(pts[j][1] + rng.normal() * r * 0.3)))
                path.cubic_to(*a, *b, *pts[j])
        path.build()
        u = rng.random()
        col = tuple(float(v) for v in rng.random(4, dtype=np.float32))
        if u < 0.05:
            props = Props(clip=int(rng.integers(1, 4)))
        else:
            if u < 0.55:
                fill = (col[0], col[1], col[2], 1.0 if rng.random() < 0.5 else col[3])
            elif u < 0.75:
                cols = [tuple(float(v) for v in rng.random(4, dtype=np.float32)) for _ in range(int(rng.integers(2, 5)))]
                fill = gradient((float(cx - r), float(cy - r)), (float(cx + r), float(cy + r * 0.5)), cols, radial=bool(rng.random() < 0.4))
            elif u < 0.85:
                fill = Texture((0.1, 0.02, -0.03, 0.1, float(-cx * 0.1 + 2), float(-cy * 0.1 + 2)), img)
            else:
                fill = (col[0], col[1], col[2], 0.5)
            props = Props(fill_rule="EvenOdd" if rng.random() < 0.3 else "NonZero", fill=fill,
                          blend_mode=BLEND_MODES[int(rng.integers(0, 16))] if rng.random() < 0.4 else "Over",
                          is_clipped=bool(rng.random() < 0.15))
        layer = c.get_mut_or_insert_default(order).insert(path).set_props(props)
        if rng.random() < 0.2:
            th = rng.uniform(-0.3, 0.3)
            cs, sn = float(np.float32(np.cos(th))), float(np.float32(np.sin(th)))
            layer.set_transform((cs, -sn, sn, cs, float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20))))
        order += int(rng.integers(1, 3))
    return c
