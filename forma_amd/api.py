"""Python face of forma's public API for the MI355X backend — same names, argument meaning and
error behaviour as the reference (`forma::prelude`, reference forma/src/lib.rs:117-154):

    PathBuilder / Path / Point            forma/src/path.rs:773-925, math/point.rs
    Order, Color, FillRule, BlendMode, GradientBuilder, Gradient, GradientType, Image, Texture,
    Fill, Style, Func, Props              forma/src/styling.rs, utils/order.rs
    GeomPresTransform, AffineTransform    forma/src/math/transform.rs
    Composition / Layer                   forma/src/composition/{mod,layer}.rs
    hip.Renderer  (drop-in for cpu::Renderer: new / create_buffer_layer_cache / render,
                   forma/src/cpu/renderer.rs:61-224), BufferBuilder, LinearLayout, RGBA.., Rect

Path construction and the sequential half of curve flattening run in the C++ host library
(csrc/host_path.cpp); everything per-frame runs in HIP kernels behind the C ABI.  There is no CPU
fallback and no dependency on oracle/.
"""
from __future__ import annotations

import ctypes as C
import struct
import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import FormaError
from .context import Context, GEOM_DTYPE, IMAGE_DTYPE

LAYER_LIMIT = (1 << 21) - 1          # consts.rs:107-109
LINES_GARBAGE_THRESHOLD = 2          # composition/mod.rs:33
MAX_WIDTH, MAX_HEIGHT = 1 << 16, 1 << 15
NONE = 0xFFFFFFFF

_host_bound = False


def _host():
    global _host_bound
    L = _lib.lib()
    if not _host_bound:
        vp, f, sz = C.c_void_p, C.c_float, C.c_size_t
        L.forma_host_builder_new.restype = vp
        L.forma_host_builder_free.argtypes = [vp]
        L.forma_host_move_to.argtypes = [vp, f, f]
        L.forma_host_line_to.argtypes = [vp, f, f]
        L.forma_host_quad_to.argtypes = [vp, f, f, f, f]
        L.forma_host_cubic_to.argtypes = [vp, f, f, f, f, f, f]
        L.forma_host_rat_quad_to.argtypes = [vp, f, f, f, f, f]
        L.forma_host_rat_cubic_to.argtypes = [vp, f, f, f, f, f, f, f, f]
        L.forma_host_build.argtypes = [vp]; L.forma_host_build.restype = vp
        L.forma_host_path_free.argtypes = [vp]
        L.forma_host_path_transform.argtypes = [vp, vp]; L.forma_host_path_transform.restype = vp
        L.forma_host_path_points.argtypes = [vp]; L.forma_host_path_points.restype = sz
        L.forma_host_path_lines.argtypes = [vp]; L.forma_host_path_lines.restype = sz
        L.forma_host_batch_new.restype = vp
        L.forma_host_batch_free.argtypes = [vp]
        L.forma_host_batch_add.argtypes = [vp, vp, C.c_uint32]
        L.forma_host_batch_points.argtypes = [vp]; L.forma_host_batch_points.restype = sz
        L.forma_host_batch_flatten.argtypes = [vp, vp, vp, vp, vp]; L.forma_host_batch_flatten.restype = C.c_int
        L.forma_host_batch_tables.argtypes = [vp, vp]
        _host_bound = True
    return L


# ---- math ---------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Point:
    x: float
    y: float


@dataclass(frozen=True)
class AffineTransform:                       # math/transform.rs:22-57
    ux: float = 1.0
    uy: float = 0.0
    vx: float = 0.0
    vy: float = 1.0
    tx: float = 0.0
    ty: float = 0.0

    def to_array(self):
        return [self.ux, self.uy, self.vx, self.vy, self.tx, self.ty]

    def transform(self, p: "Point") -> "Point":        # math/transform.rs:43-48 (f32, fused multiply-adds)
        f = np.float32
        def fma(a, b, c):                               # exact for f32 operands: the product of two f32 fits an f64
            return f(np.float64(f(a)) * np.float64(f(b)) + np.float64(f(c)))
        return Point(float(fma(self.ux, p.x, fma(self.vx, p.y, self.tx))), float(fma(self.uy, p.x, fma(self.vy, p.y, self.ty))))


class GeomPresTransformError(ValueError):
    pass


class GeomPresTransform:
    """Affine transform that does not scale geometry up (math/transform.rs:151-222)."""

    def __init__(self, t: AffineTransform = AffineTransform()):
        self.t = t

    @staticmethod
    def try_from(a: Sequence[float]) -> "GeomPresTransform":   # [ux, vx, uy, vy, tx, ty], transform.rs:80-91
        f = [float(np.float32(v)) for v in a]
        t = AffineTransform(ux=f[0], uy=f[2], vx=f[1], vy=f[3], tx=f[4], ty=f[5])
        max_x = np.float32(1.0) + np.float32(1.0 / 16.0) / np.float32(MAX_WIDTH)
        max_y = np.float32(1.0) + np.float32(1.0 / 16.0) / np.float32(MAX_HEIGHT)
        sx = np.float32(t.ux) * np.float32(t.ux) + np.float32(t.uy) * np.float32(t.uy) > max_x
        sy = np.float32(t.vx) * np.float32(t.vx) + np.float32(t.vy) * np.float32(t.vy) > max_y
        if sx or sy:
            raise GeomPresTransformError(f"exceeded scaling factor (x: {bool(sx)}, y: {bool(sy)})")
        return GeomPresTransform(t)

    def is_identity(self) -> bool:
        return self.t == AffineTransform()

    def transform(self, p: "Point") -> "Point":
        return self.t.transform(p)

    def to_array(self):
        t = self.t
        return [t.ux, t.vx, t.uy, t.vy, t.tx, t.ty]


# ---- paths ----------------------------------------------------------------------------------------
class Path:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:
            _host().forma_host_path_free(self._h)
        except Exception:
            pass

    def transform(self, t9: Sequence[float]) -> "Path":           # path.rs:725-771
        t = np.ascontiguousarray(t9, np.float32)
        assert t.size == 9
        return Path(_host().forma_host_path_transform(self._h, t.ctypes.data))


class PathBuilder:
    def __init__(self):
        self._h = _host().forma_host_builder_new()

    def __del__(self):
        try:
            _host().forma_host_builder_free(self._h)
        except Exception:
            pass

    def move_to(self, p: Point):
        _host().forma_host_move_to(self._h, p.x, p.y); return self

    def line_to(self, p: Point):
        _host().forma_host_line_to(self._h, p.x, p.y); return self

    def quad_to(self, p1: Point, p2: Point):
        _host().forma_host_quad_to(self._h, p1.x, p1.y, p2.x, p2.y); return self

    def cubic_to(self, p1: Point, p2: Point, p3: Point):
        _host().forma_host_cubic_to(self._h, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y); return self

    def rat_quad_to(self, p1: Point, p2: Point, weight: float):
        _host().forma_host_rat_quad_to(self._h, p1.x, p1.y, p2.x, p2.y, weight); return self

    def rat_cubic_to(self, p1: Point, p2: Point, p3: Point, w1: float, w2: float):
        _host().forma_host_rat_cubic_to(self._h, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y, w1, w2); return self

    def build(self) -> Path:
        return Path(_host().forma_host_build(self._h))


# ---- styling (forma/src/styling.rs) --------------------------------------------------------------
class OrderError(ValueError):
    pass


class Order(int):
    MAX = LAYER_LIMIT

    def __new__(cls, v: int):
        if v < 0 or v > LAYER_LIMIT:
            raise OrderError(f"exceeded layer limit ({LAYER_LIMIT})")        # utils/order.rs:27-31
        return super().__new__(cls, v)

    @staticmethod
    def new(v: int) -> "Order":
        return Order(v)

    def as_u32(self) -> int:
        return int(self)


@dataclass(frozen=True)
class Color:
    r: float = 0.0
    g: float = 0.0
    b: float = 0.0
    a: float = 1.0


class FillRule:
    NonZero = "NonZero"
    EvenOdd = "EvenOdd"


class GradientType:
    Linear = "Linear"
    Radial = "Radial"


BLEND_MODES = ["Over", "Multiply", "Screen", "Overlay", "Darken", "Lighten", "ColorDodge", "ColorBurn", "HardLight",
               "SoftLight", "Difference", "Exclusion", "Hue", "Saturation", "Color", "Luminosity"]


class BlendMode:
    pass


for _i, _n in enumerate(BLEND_MODES):
    setattr(BlendMode, _n, _n)


@dataclass(frozen=True)
class Gradient:
    type: str
    start: Point
    end: Point
    stops: Tuple[Tuple[Color, float], ...]


class GradientBuilder:                        # styling.rs:80-134
    def __init__(self, start: Point, end: Point):
        self._type = GradientType.Linear
        self._start, self._end = start, end
        self._stops: List[Tuple[Color, Optional[float]]] = []

    def type(self, t):
        self._type = t; return self

    def color(self, color: Color):
        self._stops.append((color, None)); return self

    def color_with_stop(self, color: Color, stop: float):
        if not (0.0 <= stop <= 1.0):
            raise ValueError("gradient stops must be between 0.0 and 1.0")
        self._stops.append((color, stop)); return self

    def build(self) -> Optional[Gradient]:
        if len(self._stops) < 2:
            return None
        inc = np.float32(1.0) / np.float32(len(self._stops) - 1)
        stops = tuple((c, float(np.float32(i) * inc) if s is None else float(np.float32(s))) for i, (c, s) in enumerate(self._stops))
        return Gradient(self._type, self._start, self._end, stops)


class ImageError(ValueError):
    pass


def _to_linear(l8: np.ndarray) -> np.ndarray:                 # styling.rs:252-259 (powf = libm powf)
    l = l8.astype(np.float32) * (np.float32(1.0) / np.float32(255.0))
    lo = l * (np.float32(1.0) / np.float32(12.92))
    hi = np.power((l + np.float32(0.055)) * (np.float32(1.0) / np.float32(1.055)), np.float32(2.4), dtype=np.float32)
    return np.where(l <= np.float32(0.04045), lo, hi).astype(np.float32)


def _to_f16(v: np.ndarray) -> np.ndarray:                     # bias-shifted half, styling.rs:242-250
    b = np.ascontiguousarray(v, np.float32).view(np.uint32)
    return np.where(v != 0, ((b - np.uint32(0x38000000)) >> np.uint32(13)) & np.uint32(0xFFFF), 0).astype(np.uint16)


class Image:
    _next_id = 0

    def __init__(self, texels: np.ndarray, width: int, height: int):
        if width * height != len(texels):
            raise ImageError(f"buffer has {len(texels)} pixels, which does not match the specified width ({width}) and height ({height})")
        self.texels, self.width_, self.height_ = texels, width, height
        self.id = Image._next_id; Image._next_id += 1

    @staticmethod
    def from_srgba(data, width: int, height: int) -> "Image":
        d = np.asarray(data, np.uint8).reshape(-1, 4)
        lin = np.empty(d.shape, np.float32)
        lin[:, :3] = _to_linear(d[:, :3])
        lin[:, 3] = d[:, 3].astype(np.float32) * (np.float32(1.0) / np.float32(255.0))
        return Image(_to_f16(lin), width, height)

    @staticmethod
    def from_linear_rgba(data, width: int, height: int) -> "Image":
        return Image(_to_f16(np.asarray(data, np.float32).reshape(-1, 4)), width, height)

    def width(self):
        return self.width_

    def height(self):
        return self.height_


@dataclass(frozen=True)
class Texture:
    transform: AffineTransform
    image: Image


class Fill:
    @staticmethod
    def Solid(color: Color):
        return ("solid", color)

    @staticmethod
    def Gradient(g: Gradient):
        return ("gradient", g)

    @staticmethod
    def Texture(t: Texture):
        return ("texture", t)


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
    def Clip(n: int):
        return ("clip", int(n))


@dataclass(frozen=True)
class Props:
    fill_rule: str = FillRule.NonZero
    func: tuple = ("draw", Style())


def _f32bits(v: float) -> int:
    return struct.unpack("<I", struct.pack("<f", v))[0]


def _encode_props(p: Props, images: List[Image]) -> List[int]:
    """Props -> style words (layout: include/forma_hip.h)."""
    h = (1 << 6) if p.fill_rule == FillRule.EvenOdd else 0
    kind, payload = p.func
    if kind == "clip":
        return [h | (1 << 8), payload]
    st: Style = payload
    h |= BLEND_MODES.index(st.blend_mode)
    if st.is_clipped:
        h |= 1 << 7
    fk, fv = st.fill
    if fk == "gradient":
        g: Gradient = fv
        h |= (2 if g.type == GradientType.Radial else 1) << 4
        h |= len(g.stops) << 16
        w = [h, 0] + [_f32bits(v) for v in (g.start.x, g.start.y, g.end.x, g.end.y)]
        for c, s in g.stops:
            w += [_f32bits(v) for v in (c.r, c.g, c.b, c.a, s)]
        return w
    if fk == "texture":
        t: Texture = fv
        idx = next((i for i, im in enumerate(images) if im is t.image), None)
        if idx is None:
            images.append(t.image); idx = len(images) - 1
        return [h | (3 << 4), 0] + [_f32bits(v) for v in t.transform.to_array()] + [idx]
    c: Color = fv
    return [h, 0] + [_f32bits(v) for v in (c.r, c.g, c.b, c.a)]


# ---- composition (forma/src/composition/{mod,layer,state}.rs) -------------------------------------------
class _Shared:
    def __init__(self):
        self.pushes: List[Tuple[int, Path, int]] = []   # (geom_id, path, lines) in SegmentBuffer order
        self.geom_id_to_order: Dict[int, Optional[int]] = {}
        self.next_geom_id = 1
        self.geometry_version = 0
        self.table_version = 0         # bumped by every change that the per-frame layer / style tables depend on
        self.unchanged_version = 0     # bumped when a render call changes some layer's is_unchanged set

    def new_geom_id(self) -> int:
        g = self.next_geom_id; self.next_geom_id += 1; return g


def _drop_layer(shared: "_Shared", gid_cell):
    shared.geom_id_to_order.pop(gid_cell[0], None)
    shared.table_version += 1


class Layer:
    def __init__(self, shared: _Shared):
        self._shared = shared
        self.is_enabled_ = True
        self.affine_transform: Optional[GeomPresTransform] = None
        self.order: Optional[int] = None
        self.geom_id_ = shared.new_geom_id()
        self.props_ = Props()
        self.is_unchanged_: set = set()
        self.lines_count = 0
        # `impl Drop for Layer` (layer.rs:356-363): a layer that goes away takes its geometry id out of
        # geom_id_to_order, which is what lets compact_geom collect its lines.  The id lives in a one-element
        # list because Layer::clear replaces it.
        self._gid_cell = [self.geom_id_]
        self._drop = weakref.finalize(self, _drop_layer, shared, self._gid_cell)

    def geom_id(self) -> int:                                 # layer.rs:143-145
        return self.geom_id_

    def insert(self, path: Path) -> "Layer":                  # layer.rs:90-111
        n = _host().forma_host_path_points(path._h)
        lines = _host().forma_host_path_lines(path._h)        # ids that are Some: what SegmentBuffer::len counts
        self._shared.pushes.append((self.geom_id_, path, lines))
        self._shared.geom_id_to_order[self.geom_id_] = self.order
        if n:
            self._shared.geometry_version += 1
        self._shared.table_version += 1
        self.lines_count += lines
        self.is_unchanged_.clear()
        return self

    def clear(self) -> "Layer":                               # layer.rs:113-129
        self._shared.geom_id_to_order.pop(self.geom_id_, None)
        self.geom_id_ = self._shared.new_geom_id()
        self._gid_cell[0] = self.geom_id_
        self._shared.geom_id_to_order[self.geom_id_] = self.order
        self._shared.geometry_version += 1
        self._shared.table_version += 1
        self.lines_count = 0
        self.is_unchanged_.clear()
        return self

    def _set_order(self, order: Optional[int]):               # layer.rs:131-141
        if order is not None and self.order != order:
            self.order = order
            self.is_unchanged_.clear()
        self._shared.geom_id_to_order[self.geom_id_] = order
        self._shared.table_version += 1

    def is_enabled(self) -> bool:
        return self.is_enabled_

    def set_is_enabled(self, v: bool) -> "Layer":
        if self.is_enabled_ != v:
            self._shared.table_version += 1
        self.is_enabled_ = v; return self

    def disable(self):
        return self.set_is_enabled(False)

    def enable(self):
        return self.set_is_enabled(True)

    def transform(self) -> GeomPresTransform:
        return self.affine_transform or GeomPresTransform()

    def set_transform(self, t: GeomPresTransform) -> "Layer":  # layer.rs:206-217
        new = None if t.is_identity() else t
        old = self.affine_transform
        if (old is None) != (new is None) or (old is not None and old.t != new.t):
            self.is_unchanged_.clear()
            self.affine_transform = new
            self._shared.table_version += 1
        return self

    def props(self) -> Props:
        return self.props_

    def set_props(self, props: Props) -> "Layer":             # layer.rs:225-233
        if self.props_ != props:
            self.is_unchanged_.clear()
            self.props_ = props
            self._shared.table_version += 1
        return self


class Composition:
    def __init__(self):
        self._shared = _Shared()
        self.layers: Dict[int, Layer] = {}

    def create_layer(self) -> Layer:
        return Layer(self._shared)

    def is_empty(self) -> bool:
        return not self.layers

    def __len__(self):
        return len(self.layers)

    def insert(self, order: Order, layer: Layer) -> Optional[Layer]:     # composition/mod.rs:146-161
        if layer._shared is not self._shared:
            raise AssertionError("Layer was crated by a different Composition")
        layer._set_order(int(order))
        old = self.layers.get(int(order))
        self.layers[int(order)] = layer
        self._shared.table_version += 1
        if old is not None and old is not layer:
            old._set_order(None)
        return old

    def remove(self, order: Order) -> Optional[Layer]:
        layer = self.layers.pop(int(order), None)
        if layer is not None:
            layer._set_order(None)
        return layer

    def get_order_if_stored(self, geom_id: int) -> Optional[Order]:      # composition/mod.rs:234-241
        o = self._shared.geom_id_to_order.get(geom_id)
        return None if o is None else Order(o)

    def builder_len(self) -> int:                                         # composition/mod.rs:345-352
        return sum(n for _, _, n in self._shared.pushes)

    def actual_len(self) -> int:                                          # composition/mod.rs:354-356
        return sum(l.lines_count for l in self.layers.values())

    def compact_geom(self) -> bool:
        """Geometry garbage collection (composition/mod.rs:372-384): drops the pushes of geometry ids nobody can reach
        any more once they make up at least half of the store.  Returns True if anything was dropped."""
        sh = self._shared
        if self.builder_len() >= self.actual_len() * LINES_GARBAGE_THRESHOLD:
            live = [p for p in sh.pushes if p[0] in sh.geom_id_to_order]
            if len(live) != len(sh.pushes):
                sh.pushes = live
                sh.geometry_version += 1
                return True
        return False

    def get(self, order: Order) -> Optional[Layer]:
        return self.layers.get(int(order))

    def get_mut(self, order: Order) -> Optional[Layer]:
        return self.layers.get(int(order))

    def get_mut_or_insert_default(self, order: Order) -> Layer:
        if int(order) not in self.layers:
            self.insert(order, self.create_layer())
        return self.layers[int(order)]

    def layers_iter(self):
        return self.layers.items()


# ---- buffers (forma/src/cpu/buffer) ---------------------------------------------------------------------
RGBA = (0, 1, 2, 3)      # cpu/channel.rs:57-62
BGRA = (2, 1, 0, 3)
RGB0 = (0, 1, 2, 4)
BGR0 = (2, 1, 0, 4)
RGB1 = (0, 1, 2, 5)
BGR1 = (2, 1, 0, 5)


@dataclass
class Rect:                                   # pixels; rounded out to tiles by the backend (renderer.rs:43-52)
    horizontal: range
    vertical: range


class TileFill:
    """What `Layout.write` receives (cpu/buffer/layout/mod.rs:36-44): ("solid", [u8; 4]) or ("full", colors) with colors a
    [256][4] uint8 array in COLUMN-major order (index = x * TILE_HEIGHT + y), exactly the reference's `TileFill::Full`."""

    @staticmethod
    def Solid(color):
        return ("solid", color)

    @staticmethod
    def Full(colors):
        return ("full", colors)


class Layout:
    """A buffer's layout description (reference trait `Layout`, cpu/buffer/layout/mod.rs:51-163).  Subclass it to describe
    a non-linear buffer: the renderer then hands every written tile to `write` (the generic, host-side path);
    `LinearLayout` is the fast path (one strided device-to-host copy straight into the caller's buffer)."""

    def width(self) -> int:
        raise NotImplementedError

    def height(self) -> int:
        raise NotImplementedError

    def slices_per_tile(self) -> int:
        raise NotImplementedError

    def slices(self, buffer: np.ndarray) -> list:
        """writable views of `buffer`, tile-major: tiles ordered by (tile_y, tile_x), `slices_per_tile()` views each"""
        raise NotImplementedError

    @staticmethod
    def write(slices: list, flusher, fill) -> None:
        raise NotImplementedError

    def width_in_tiles(self) -> int:
        return (self.width() + 15) >> 4

    def height_in_tiles(self) -> int:
        return (self.height() + 15) >> 4


class LinearLayout(Layout):                   # cpu/buffer/layout/mod.rs:167-295
    def __init__(self, width: int, width_stride: int, height: int):
        if width * 4 > width_stride:
            raise AssertionError(f"width exceeds width stride: {width} * 4 > {width_stride}")
        self._w, self.width_stride, self._h = width, width_stride, height

    def width(self) -> int:
        return self._w

    def height(self) -> int:
        return self._h

    def slices_per_tile(self) -> int:
        return 16

    def slices(self, buffer: np.ndarray) -> list:                        # :186-213: rows of every tile, sorted by (tile_y, tile_x)
        flat = buffer.reshape(-1)
        assert self._h * self.width_stride <= flat.size, "height * width_stride exceeds buffer length"
        out = []
        for ty in range(self.height_in_tiles()):
            for tx in range(self.width_in_tiles()):
                for y in range(ty * 16, ty * 16 + 16):
                    if y >= self._h:
                        out.append(flat[0:0])                            # (edge tiles: fewer rows; keep slices_per_tile entries)
                        continue
                    o = y * self.width_stride + tx * 64
                    out.append(flat[o: o + min(64, (self._w - tx * 16) * 4)])
        return out

    @staticmethod
    def write(slices: list, flusher, fill) -> None:                      # :264-295
        kind, payload = fill
        for y, row in enumerate(slices):
            px = row.reshape(-1, 4)
            if kind == "solid":
                px[:] = np.asarray(payload, np.uint8)
            else:
                px[:] = np.asarray(payload, np.uint8).reshape(-1, 4)[np.arange(len(px)) * 16 + y]
        if flusher is not None:
            for row in slices:
                if len(row):
                    flusher.flush(row[:64])


class BufferLayerCache:                       # cpu/buffer/mod.rs:165-197
    def __init__(self, cache_id: int, renderer: "Renderer"):
        self.id = cache_id
        self._renderer = renderer

    def clear(self):
        self._renderer._ctx.cache_clear(self.id)

    def __del__(self):                        # IdDropper (cpu/buffer/mod.rs:98-111): the id returns to the renderer's pool
        try:
            r = self._renderer
            if self.id in r._caches:
                r._caches.discard(self.id)
                r._ctx.cache_clear(self.id)   # the next owner of the id starts from an empty cache
                r._tables_key = None
        except Exception:
            pass


@dataclass
class Buffer:
    buffer: np.ndarray
    layout: LinearLayout
    layer_cache: Optional[BufferLayerCache] = None
    flusher: Optional[object] = None


class BufferBuilder:
    def __init__(self, buffer: np.ndarray, layout: LinearLayout):
        self._b = Buffer(buffer, layout)

    def layer_cache(self, cache: BufferLayerCache):
        self._b.layer_cache = cache; return self

    def flusher(self, flusher):
        self._b.flusher = flusher; return self

    def build(self) -> Buffer:
        return self._b


# ---- the renderer (drop-in for cpu::Renderer) -----------------------------------------------------------------
class Renderer:
    """`forma::hip::Renderer`: same three methods as `cpu::Renderer` (cpu/renderer.rs:61-224)."""

    def __init__(self, device: int = 0, devices=None, frames_in_flight: int = 1):
        """`Renderer::new()` (cpu/renderer.rs:63-65).  `devices`: one renderer over several GPUs of this process
        (forma_hip_create_multi — the Rust shim's `Renderer::with_devices`); `frames_in_flight`: device-resident frames
        are pipelined inside the renderer (forma_hip_set_frames_in_flight)."""
        self._ctx = Context(device, devices=devices, frames_in_flight=frames_in_flight)
        self._caches = set()
        self._geom_version = None
        self._geom_owner = None
        self._slot_of: Dict[int, int] = {}
        self.last_timings = None
        self.host_tables: Dict[str, np.ndarray] = {}     # last uploaded scene tables (inspection / tests / bench)
        self._tables_key = None                          # (composition, table version, unchanged version, cache) on the device
        self._marked_key = None                          # (composition, table version, cache) whose layers are all marked unchanged

    def create_buffer_layer_cache(self) -> Optional[BufferLayerCache]:     # at most 32 (SmallBitSet)
        for i in range(32):
            if i not in self._caches:
                self._caches.add(i)
                return BufferLayerCache(i, self)
        return None

    # -- geometry store: flatten every live path in one HIP launch, keep the result resident on the device
    def _upload_geometry(self, comp: Composition):
        sh = comp._shared
        live = [(g, p) for g, p, _ in sh.pushes]      # garbage is kept until compact_geom drops it; its slots get order NONE
        slot_of: Dict[int, int] = {}
        H = _host()
        batch = H.forma_host_batch_new()
        try:
            for g, p in live:
                slot = slot_of.setdefault(g, len(slot_of))
                H.forma_host_batch_add(batch, p._h, slot)
            n = H.forma_host_batch_points(batch)
            x = np.zeros(n, np.float32); y = np.zeros(n, np.float32); ls = np.zeros(n, np.uint32)
            if n:
                rc = H.forma_host_batch_flatten(batch, self._ctx._h, x.ctypes.data, y.ctypes.data, ls.ctypes.data)
                self._ctx._check(rc)
        finally:
            H.forma_host_batch_free(batch)
        self._ctx.set_geometry(x, y, ls[: max(n - 1, 0)])
        self.host_tables.update(x=x, y=y, line_slot=ls[: max(n - 1, 0)].copy())
        self._slot_of = slot_of
        self._geom_version = sh.geometry_version
        self._geom_owner = sh

    def _upload_tables(self, comp: Composition, cache_id: Optional[int]):
        sh = comp._shared
        geoms = np.zeros(max(len(self._slot_of), 1), GEOM_DTYPE)
        geoms["order"] = NONE
        by_geom = {l.geom_id_: l for l in comp.layers.values()}
        for g, slot in self._slot_of.items():
            layer = by_geom.get(g)
            order = sh.geom_id_to_order.get(g)
            if layer is None or order is None or not layer.is_enabled_:
                continue
            geoms[slot]["order"] = order
            if layer.affine_transform is not None:
                geoms[slot]["flags"] = 1
                geoms[slot]["xf"] = layer.affine_transform.t.to_array()
        n_orders = (max(comp.layers) + 1) if comp.layers else 0
        offsets = np.full(n_orders, NONE, np.uint32)
        unchanged = np.zeros(n_orders, np.uint8)
        words: List[int] = []
        images: List[Image] = []
        interned: Dict[Props, int] = {}
        for order, layer in comp.layers.items():
            off = interned.get(layer.props_)
            if off is None:
                off = len(words); interned[layer.props_] = off
                words += _encode_props(layer.props_, images)
            offsets[order] = off
            unchanged[order] = 1 if (cache_id is not None and cache_id in layer.is_unchanged_) else 0
        self._ctx.set_geoms(geoms)
        self._ctx.set_styles(offsets, np.asarray(words, np.uint32), unchanged)
        img_tab = np.zeros(len(images), IMAGE_DTYPE)
        tex, off = [], 0
        for i, im in enumerate(images):
            img_tab[i] = (off, im.width_, im.height_)
            tex.append(im.texels); off += len(im.texels)
        texels = np.concatenate(tex) if tex else np.zeros((0, 4), np.uint16)
        self._ctx.set_images(img_tab, texels)
        self.host_tables.update(geoms=geoms, style_offsets=offsets, style_words=np.asarray(words, np.uint32),
                                unchanged=unchanged, images=img_tab, texels=texels)

    def render(self, composition: Composition, buffer: Buffer, channels=RGBA, clear_color: Color = Color(1, 1, 1, 1),
               crop: Optional[Rect] = None, timings: bool = False):
        lay = buffer.layout
        W, H = lay.width(), lay.height()
        if W > MAX_WIDTH or H > MAX_HEIGHT:
            raise FormaError(-1, "canvas exceeds MAX_WIDTH x MAX_HEIGHT")
        composition.compact_geom()                            # renderer.rs:113
        sh = composition._shared
        if self._geom_owner is not sh or self._geom_version != sh.geometry_version:
            self._upload_geometry(composition)
        cache_id = buffer.layer_cache.id if buffer.layer_cache else None
        # the layer / style / unchanged tables stay resident: they are rebuilt only when something they depend on changed
        key = (sh, sh.table_version, sh.unchanged_version if cache_id is not None else -1, cache_id)
        if key != self._tables_key:
            self._upload_tables(composition, cache_id)
            self._tables_key = key
        dst = buffer.buffer
        rect = None if crop is None else (crop.horizontal.start, crop.horizontal.stop, crop.vertical.start, crop.vertical.stop)
        clear = (clear_color.r, clear_color.g, clear_color.b, clear_color.a)
        cid = -1 if cache_id is None else cache_id
        if type(lay) is LinearLayout:                         # fast path: one strided copy of what was written
            assert dst.dtype == np.uint8 and dst.size >= lay.width_stride * H
            out = self._ctx.render(W, H, channels=channels, clear=clear, crop=rect, cache_id=cid, dst=dst.reshape(-1),
                                   stride=lay.width_stride, timings=timings)
            if buffer.flusher is not None:                    # Flusher::flush on every row slice of every written tile
                flat = dst.reshape(-1)                        # (layout/mod.rs:283-294, painter/mod.rs:537-548)
                tw = (W + 15) >> 4
                for t in np.flatnonzero(self._ctx.tiles_written(W, H)):
                    ty, tx = divmod(int(t), tw)
                    n = min(64, (W - tx * 16) * 4)
                    for y in range(ty * 16, min(ty * 16 + 16, H)):
                        o = y * lay.width_stride + tx * 64
                        buffer.flusher.flush(flat[o: o + n])
        else:                                                 # generic Layout: every written tile goes through Layout::write
            out = self._ctx.render(W, H, channels=channels, clear=clear, crop=rect, cache_id=cid, dst=None, timings=timings,
                                   device_only=True)
            img = self._ctx.read_image(W, H).reshape(H, W, 4)
            slices = lay.slices(dst)
            spt = lay.slices_per_tile()
            tw = (W + 15) >> 4
            for t in np.flatnonzero(self._ctx.tiles_written(W, H)):
                ty, tx = divmod(int(t), tw)
                tile = np.zeros((16, 16, 4), np.uint8)        # [x][y]: column-major like TileFill::Full
                blk = img[ty * 16: ty * 16 + 16, tx * 16: tx * 16 + 16]
                tile[: blk.shape[1], : blk.shape[0]] = blk.transpose(1, 0, 2)
                type(lay).write(slices[int(t) * spt: (int(t) + 1) * spt], buffer.flusher, TileFill.Full(tile.reshape(256, 4)))
        if timings:
            self.last_timings = out[1] if isinstance(out, tuple) else out
        if cache_id is not None:                              # renderer.rs:217-223
            mkey = (sh, sh.table_version, cache_id)
            if mkey != self._marked_key:                      # (nothing to do when this exact state was marked already)
                changed = False
                for layer in composition.layers.values():
                    if layer.is_enabled_:
                        if cache_id not in layer.is_unchanged_:
                            layer.is_unchanged_.add(cache_id); changed = True
                    elif cache_id in layer.is_unchanged_:
                        layer.is_unchanged_.discard(cache_id); changed = True
                if changed:
                    sh.unchanged_version += 1
                self._marked_key = mkey
