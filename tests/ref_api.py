"""Test-side restatement of forma's public scene / renderer API ON TOP OF THE ORACLE (test infrastructure).

Same names and call shapes as the product mirror `forma_amd.api`, so that the reference's own composition-level
tests (forma/src/composition/mod.rs:482-1428, cpu/buffer/mod.rs doc test) can be written once and run against
both backends: here (CPU, pins the oracle + this bookkeeping to the reference's expected buffers) and through
`forma_amd.api` on the GPU (pins the product to the same vectors).  Nothing here imports the product.

Follows: composition/mod.rs:52-395 (Composition, compact_geom), composition/layer.rs:26-260 (Layer, is_unchanged
SmallBitSet), segment.rs:95-275 (GeomId, SegmentBuffer::{len, push_path, retain}), cpu/renderer.rs:55-224,
cpu/buffer/mod.rs:98-197 (BufferLayerCache, IdDropper), utils/small_bit_set.rs:17-57.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

import scene as S
from oracle import oracle as orc

LAYER_LIMIT = (1 << 21) - 1
MAX_WIDTH, MAX_HEIGHT = 1 << 16, 1 << 15
LINES_GARBAGE_THRESHOLD = 2                      # composition/mod.rs:33
NONE = 0xFFFFFFFF

RGBA, BGRA, RGB0, BGR0, RGB1, BGR1 = S.RGBA, S.BGRA, S.RGB0, S.BGR0, S.RGB1, S.BGR1


@dataclass(frozen=True)
class Point:
    x: float
    y: float


@dataclass(frozen=True)
class Color:
    r: float = 0.0
    g: float = 0.0
    b: float = 0.0
    a: float = 1.0


class OrderError(ValueError):
    pass


class Order(int):
    def __new__(cls, v):
        if v < 0 or v > LAYER_LIMIT:
            raise OrderError(f"exceeded layer limit ({LAYER_LIMIT})")
        return super().__new__(cls, v)

    @staticmethod
    def new(v):
        return Order(v)


class FillRule:
    NonZero, EvenOdd = "NonZero", "EvenOdd"


class GradientType:
    Linear, Radial = "Linear", "Radial"


class BlendMode:
    pass


for _n in S.BLEND_MODES:
    setattr(BlendMode, _n, _n)


class GradientBuilder:                           # styling.rs:80-134
    def __init__(self, start: Point, end: Point):
        self._type, self._start, self._end, self._stops = GradientType.Linear, start, end, []

    def type(self, t):
        self._type = t; return self

    def color(self, c: Color):
        self._stops.append((c, None)); return self

    def color_with_stop(self, c: Color, stop: float):
        self._stops.append((c, stop)); return self

    def build(self):
        if len(self._stops) < 2:
            return None
        return S.gradient((self._start.x, self._start.y), (self._end.x, self._end.y),
                          [(c.r, c.g, c.b, c.a) for c, _ in self._stops], radial=self._type == GradientType.Radial,
                          stops=[s for _, s in self._stops])


class Fill:
    @staticmethod
    def Solid(c: Color):
        return ("solid", c)

    @staticmethod
    def Gradient(g):
        return ("gradient", g)


@dataclass(frozen=True)
class Style:
    is_clipped: bool = False
    fill: tuple = ("solid", Color())
    blend_mode: str = "Over"


class Func:
    @staticmethod
    def Draw(style: Style = Style()):
        return ("draw", style)

    @staticmethod
    def Clip(n):
        return ("clip", int(n))


@dataclass(frozen=True)
class Props:
    fill_rule: str = FillRule.NonZero
    func: tuple = ("draw", Style())

    def to_scene(self) -> S.Props:
        kind, payload = self.func
        if kind == "clip":
            return S.Props(fill_rule=self.fill_rule, clip=payload)
        fk, fv = payload.fill
        fill = (fv.r, fv.g, fv.b, fv.a) if fk == "solid" else fv
        return S.Props(fill_rule=self.fill_rule, fill=fill, blend_mode=payload.blend_mode, is_clipped=payload.is_clipped)


class GeomPresTransformError(ValueError):
    pass


class GeomPresTransform:                         # math/transform.rs:151-222
    def __init__(self, ux=1.0, uy=0.0, vx=0.0, vy=1.0, tx=0.0, ty=0.0):
        self.t = (ux, uy, vx, vy, tx, ty)

    @staticmethod
    def try_from(a: Sequence[float]):            # [ux, vx, uy, vy, tx, ty], transform.rs:80-91
        f = [float(np.float32(v)) for v in a]
        ux, vx, uy, vy, tx, ty = f
        max_x = np.float32(1.0) + np.float32(1.0 / 16.0) / np.float32(MAX_WIDTH)
        max_y = np.float32(1.0) + np.float32(1.0 / 16.0) / np.float32(MAX_HEIGHT)
        if (np.float32(ux) * np.float32(ux) + np.float32(uy) * np.float32(uy) > max_x
                or np.float32(vx) * np.float32(vx) + np.float32(vy) * np.float32(vy) > max_y):
            raise GeomPresTransformError("exceeded scaling factor")
        return GeomPresTransform(ux, uy, vx, vy, tx, ty)

    def is_identity(self):
        return self.t == (1.0, 0.0, 0.0, 1.0, 0.0, 0.0)


class Path:
    def __init__(self, p: orc.Path):
        self._p = p


class PathBuilder:                               # path.rs:773-925
    def __init__(self):
        self._p = orc.Path()

    def move_to(self, p):
        self._p.move_to(p.x, p.y); return self

    def line_to(self, p):
        self._p.line_to(p.x, p.y); return self

    def quad_to(self, p1, p2):
        self._p.quad_to(p1.x, p1.y, p2.x, p2.y); return self

    def cubic_to(self, p1, p2, p3):
        self._p.cubic_to(p1.x, p1.y, p2.x, p2.y, p3.x, p3.y); return self

    def rat_quad_to(self, p1, p2, w):
        self._p.rat_quad_to(p1.x, p1.y, p2.x, p2.y, w); return self

    def build(self) -> Path:
        built, self._p = self._p.build(), orc.Path()
        return Path(built)


class _SegmentBuffer:                            # segment.rs:152-275
    def __init__(self):
        self.x: List[float] = []
        self.y: List[float] = []
        self.ids: List[Optional[int]] = []

    def len(self):
        return sum(1 for i in self.ids if i is not None)

    def push_path(self, gid: int, path: Path, flattener: orc.Oracle):
        x, y, nc = flattener.flatten(path._p)      # Path::push_segments_to, path.rs:677-723
        self.x += [float(v) for v in x]; self.y += [float(v) for v in y]
        self.ids += [None if c else gid for c in nc]
        want = max(len(self.x) - 1, 0)             # ids.resize(x.len() - 1, Some(id))
        self.ids = self.ids[:want] + [gid] * (want - len(self.ids))
        if self.ids and self.ids[-1] is not None:
            self.ids.append(None)

    def retain(self, keep):                        # segment.rs:236-273
        nx, ny, ni = [], [], []
        prev = None
        for i in range(len(self.x)):
            gid = self.ids[i]
            ref = gid if gid is not None else prev
            assert ref is not None, "consecutive None values should not exist in ids"
            prev = gid
            if keep(ref):
                nx.append(self.x[i]); ny.append(self.y[i]); ni.append(gid)
        self.x, self.y, self.ids = nx, ny, ni


class _Shared:
    def __init__(self):
        self.segment_buffer = _SegmentBuffer()
        self.geom_id_to_order: Dict[int, Optional[int]] = {}
        self.next_geom_id = 1
        self.flattener = orc.Oracle()

    def new_geom_id(self):
        g = self.next_geom_id; self.next_geom_id += 1; return g


class Layer:                                     # composition/layer.rs
    def __init__(self, shared: _Shared):
        self._shared = shared
        self.is_enabled_ = True
        self.affine_transform: Optional[GeomPresTransform] = None
        self.order: Optional[int] = None
        self.geom_id_ = shared.new_geom_id()
        self.props_ = Props()
        self.is_unchanged_ = 0                   # SmallBitSet (u32)
        self.lines_count = 0

    def insert(self, path: Path):                # layer.rs:90-111
        sb = self._shared.segment_buffer
        old = sb.len()
        sb.push_path(self.geom_id_, path, self._shared.flattener)
        self._shared.geom_id_to_order[self.geom_id_] = self.order
        self.lines_count += sb.len() - old
        self.is_unchanged_ = 0
        return self

    def clear(self):                             # layer.rs:113-129
        self._shared.geom_id_to_order.pop(self.geom_id_, None)
        self.geom_id_ = self._shared.new_geom_id()
        self._shared.geom_id_to_order[self.geom_id_] = self.order
        self.lines_count = 0
        self.is_unchanged_ = 0
        return self

    def _set_order(self, order):                 # layer.rs:131-141
        if order is not None and self.order != order:
            self.order = order
            self.is_unchanged_ = 0
        self._shared.geom_id_to_order[self.geom_id_] = order

    def geom_id(self):
        return self.geom_id_

    def is_unchanged(self, cache_id):
        return bool((self.is_unchanged_ >> cache_id) & 1)

    def set_is_unchanged(self, cache_id, v):
        if v:
            self.is_unchanged_ |= 1 << cache_id
        else:
            self.is_unchanged_ &= ~(1 << cache_id)

    def is_enabled(self):
        return self.is_enabled_

    def set_is_enabled(self, v):
        self.is_enabled_ = v; return self

    def disable(self):
        return self.set_is_enabled(False)

    def enable(self):
        return self.set_is_enabled(True)

    def set_transform(self, t: GeomPresTransform):   # layer.rs:206-217
        new = None if t.is_identity() else t
        old = self.affine_transform
        if (old is None) != (new is None) or (old is not None and old.t != new.t):
            self.is_unchanged_ = 0
            self.affine_transform = new
        return self

    def set_props(self, props: Props):           # layer.rs:225-233
        if self.props_ != props:
            self.is_unchanged_ = 0
            self.props_ = props
        return self


class Composition:                               # composition/mod.rs
    def __init__(self):
        self._shared = _Shared()
        self.layers: Dict[int, Layer] = {}

    def create_layer(self):
        return Layer(self._shared)

    def is_empty(self):
        return not self.layers

    def __len__(self):
        return len(self.layers)

    def insert(self, order, layer):
        assert layer._shared is self._shared, "Layer was crated by a different Composition"
        layer._set_order(int(order))
        old = self.layers.get(int(order))
        self.layers[int(order)] = layer
        if old is not None and old is not layer:
            old._set_order(None)
            return old
        return None if old is None else old

    def remove(self, order):
        layer = self.layers.pop(int(order), None)
        if layer is not None:
            layer._set_order(None)
        return layer

    def get_order_if_stored(self, geom_id):
        return self._shared.geom_id_to_order.get(geom_id)

    def get(self, order):
        return self.layers.get(int(order))

    get_mut = get

    def get_mut_or_insert_default(self, order):
        if int(order) not in self.layers:
            self.insert(order, self.create_layer())
        return self.layers[int(order)]

    def builder_len(self):
        return self._shared.segment_buffer.len()

    def actual_len(self):
        return sum(l.lines_count for l in self.layers.values())

    def compact_geom(self):                      # composition/mod.rs:372-384
        if self.builder_len() >= self.actual_len() * LINES_GARBAGE_THRESHOLD:
            g2o = self._shared.geom_id_to_order
            self._shared.segment_buffer.retain(lambda gid: gid in g2o)


class LinearLayout:                              # cpu/buffer/layout/mod.rs:167-222
    def __init__(self, width, width_stride, height):
        if width * 4 > width_stride:
            raise AssertionError(f"width exceeds width stride: {width} * 4 > {width_stride}")
        self.width, self.width_stride, self.height = width, width_stride, height


@dataclass
class Rect:
    horizontal: range
    vertical: range


class BufferLayerCache:                          # cpu/buffer/mod.rs:165-197
    def __init__(self, cache_id, renderer):
        self.id = cache_id
        self._renderer = renderer

    def clear(self):
        self._renderer._oracle.cache_clear(self.id)

    def __del__(self):                           # IdDropper :98-111
        try:
            self._renderer._slots &= ~(1 << self.id)
            self._renderer._oracle.cache_clear(self.id)
        except Exception:
            pass


@dataclass
class Buffer:
    buffer: np.ndarray
    layout: LinearLayout
    layer_cache: Optional[BufferLayerCache] = None
    flusher: Optional[object] = None


class BufferBuilder:
    def __init__(self, buffer, layout):
        self._b = Buffer(buffer, layout)

    def layer_cache(self, cache):
        self._b.layer_cache = cache; return self

    def flusher(self, flusher):
        self._b.flusher = flusher; return self

    def build(self):
        return self._b


class Renderer:                                  # cpu/renderer.rs:55-224 over the oracle
    def __init__(self, device: int = 0):
        self._oracle = orc.Oracle()
        self._slots = 0                          # SmallBitSet buffers_with_caches

    def create_buffer_layer_cache(self):         # :68-73, SmallBitSet::first_empty_slot
        slot = 0
        while (self._slots >> slot) & 1:
            slot += 1
        if slot >= 32:
            return None
        self._slots |= 1 << slot
        return BufferLayerCache(slot, self)

    def render(self, composition: Composition, buffer: Buffer, channels=RGBA, clear_color: Color = Color(1, 1, 1, 1),
               crop: Optional[Rect] = None):
        lay = buffer.layout
        composition.compact_geom()               # :113
        sh = composition._shared
        sb = sh.segment_buffer
        # SegmentBuffer view + the two hash maps of fill_cpu_view (segment.rs:141-149, 298-340) as dense tables
        gids = sorted({g for g in sb.ids if g is not None})
        slot_of = {g: i for i, g in enumerate(gids)}
        geoms = np.zeros(max(len(gids), 1), orc.GEOM_DTYPE)
        geoms["order"] = NONE
        for g, slot in slot_of.items():
            order = sh.geom_id_to_order.get(g)
            layer = composition.layers.get(order) if order is not None else None
            if layer is None or not layer.is_enabled_:
                continue
            geoms[slot]["order"] = order
            if layer.affine_transform is not None:
                geoms[slot]["flags"] = 1
                geoms[slot]["xf"] = layer.affine_transform.t
        n = len(sb.x)
        line_slot = np.asarray([NONE if g is None else slot_of[g] for g in sb.ids[: max(n - 1, 0)]], np.uint32)
        cache_id = buffer.layer_cache.id if buffer.layer_cache is not None else None
        n_orders = (max(composition.layers) + 1) if composition.layers else 0
        offsets = np.full(n_orders, NONE, np.uint32)
        unchanged = np.zeros(n_orders, np.uint8)
        words: List[int] = []
        images: list = []
        for order, layer in composition.layers.items():
            offsets[order] = len(words)
            words += S.encode_props(layer.props_.to_scene(), images)
            unchanged[order] = 1 if (cache_id is not None and layer.is_unchanged(cache_id)) else 0
        o = self._oracle
        o.set_geometry(np.asarray(sb.x, np.float32), np.asarray(sb.y, np.float32), line_slot)
        o.set_geoms(geoms)
        o.set_styles(offsets, np.asarray(words, np.uint32), unchanged)
        rect = None if crop is None else (crop.horizontal.start, crop.horizontal.stop, crop.vertical.start, crop.vertical.stop)
        dst = buffer.buffer
        assert dst.dtype == np.uint8 and dst.size >= lay.width_stride * lay.height
        fl = None if buffer.flusher is None else buffer.flusher.flush
        o.render(lay.width, lay.height, channels=channels, clear=(clear_color.r, clear_color.g, clear_color.b, clear_color.a),
                 crop=rect, cache_id=-1 if cache_id is None else cache_id, dst=dst.reshape(-1), stride=lay.width_stride, flusher=fl)
        if cache_id is not None:                 # :217-223
            for layer in composition.layers.values():
                layer.set_is_unchanged(cache_id, layer.is_enabled_)
