"""ctypes binding of the CPU oracle (oracle/libforma_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py.  The product package `forma_amd` never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libforma_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "forma_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class GeomT(C.Structure):
    _fields_ = [("order", C.c_uint32), ("flags", C.c_uint32), ("xf", C.c_float * 6)]


class ImageT(C.Structure):
    _fields_ = [("texel_offset", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32)]


class RectT(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("x1", C.c_uint32), ("y0", C.c_uint32), ("y1", C.c_uint32)]


GEOM_DTYPE = np.dtype([("order", "<u4"), ("flags", "<u4"), ("xf", "<f4", (6,))])
IMAGE_DTYPE = np.dtype([("texel_offset", "<u8"), ("width", "<u4"), ("height", "<u4")])
NONE = 0xFFFFFFFF

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, f, i32, u32, sz, u8p = C.c_void_p, C.c_float, C.c_int, C.c_uint32, C.c_size_t, C.c_void_p
        L.oracle_create.restype = vp
        L.oracle_destroy.argtypes = [vp]
        L.oracle_set_threads.argtypes = [vp, i32]
        L.oracle_max_threads.restype = i32
        L.oracle_path_new.restype = vp
        L.oracle_path_free.argtypes = [vp]
        L.oracle_path_move_to.argtypes = [vp, f, f]
        L.oracle_path_line_to.argtypes = [vp, f, f]
        L.oracle_path_quad_to.argtypes = [vp, f, f, f, f]
        L.oracle_path_cubic_to.argtypes = [vp, f, f, f, f, f, f]
        L.oracle_path_rat_quad_to.argtypes = [vp, f, f, f, f, f]
        L.oracle_path_rat_cubic_to.argtypes = [vp, f, f, f, f, f, f, f, f]
        L.oracle_path_close.argtypes = [vp]
        L.oracle_path_transform9.argtypes = [vp, vp]
        L.oracle_path_flatten.argtypes = [vp, vp, vp]
        L.oracle_path_flatten.restype = sz
        L.oracle_flatten_get.argtypes = [vp, vp, vp, vp]
        L.oracle_path_counts.argtypes = [vp, vp]
        L.oracle_path_counts.restype = sz
        L.oracle_path_get.argtypes = [vp, vp, vp, vp, vp]
        L.oracle_prim_new.restype = vp
        L.oracle_prim_free.argtypes = [vp]
        L.oracle_prim_push_contour.argtypes = [vp]
        L.oracle_prim_push_line.argtypes = [vp, vp]
        L.oracle_prim_push_quad.argtypes = [vp, vp]
        L.oracle_prim_push_cubic.argtypes = [vp, vp]
        L.oracle_prim_flatten.argtypes = [vp, vp]
        L.oracle_prim_flatten.restype = sz
        L.oracle_set_geometry.argtypes = [vp, vp, vp, vp, sz]
        L.oracle_set_geoms.argtypes = [vp, vp, sz]
        L.oracle_set_styles.argtypes = [vp, vp, sz, vp, sz, vp]
        L.oracle_set_images.argtypes = [vp, vp, sz, vp, sz]
        L.oracle_prepare_lines.argtypes = [vp, f, f] + [vp] * 10
        L.oracle_rasterize.argtypes = [vp]
        L.oracle_rasterize.restype = sz
        L.oracle_sort.argtypes = [vp]
        L.oracle_sort.restype = sz
        L.oracle_get_segments.argtypes = [vp, i32, vp]
        L.oracle_sort_array.argtypes = [vp, sz]
        L.oracle_pixel_segment_new.argtypes = [u32, i32, i32, i32, i32, i32, i32]
        L.oracle_pixel_segment_new.restype = C.c_uint64
        L.oracle_find.argtypes = [i32, f, f, f, f]
        L.oracle_find.restype = f
        L.oracle_coverage.argtypes = [C.c_int32, i32]
        L.oracle_coverage.restype = f
        L.oracle_srgb_bytes.argtypes = [vp, vp]
        L.oracle_linear_to_srgb.argtypes = [f]
        L.oracle_linear_to_srgb.restype = f
        L.oracle_to_u8.argtypes = [f]
        L.oracle_to_u8.restype = u32
        L.oracle_blend_simd.argtypes = [i32, vp, vp, vp]
        L.oracle_blend_scalar.argtypes = [i32, vp, vp, vp]
        L.oracle_blend_fn.argtypes = [i32, i32, vp, vp]
        L.oracle_blend_fn.restype = f
        L.oracle_f16_to_f32.argtypes = [C.c_uint16]
        L.oracle_f16_to_f32.restype = f
        L.oracle_f32_to_f16.argtypes = [f]
        L.oracle_f32_to_f16.restype = C.c_uint16
        L.oracle_srgb_to_linear.argtypes = [C.c_uint8]
        L.oracle_srgb_to_linear.restype = f
        L.oracle_gradient_column.argtypes = [vp, f, f, vp]
        L.oracle_paint.argtypes = [vp, vp, sz, vp, u32, u32, sz, vp, vp, vp, i32, vp]
        L.oracle_paint.restype = i32
        L.oracle_cache_clear.argtypes = [vp, i32]
        L.oracle_render.argtypes = [vp, vp, u32, u32, sz, vp, vp, vp, i32, vp]
        L.oracle_render.restype = i32
        L.oracle_last_n.argtypes = [vp]
        L.oracle_last_n.restype = sz
        L.oracle_time_frame.argtypes = [vp, u32, u32, i32, vp, vp, vp, vp]
        L.oracle_time_frame.restype = i32
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Path:
    """oracle restatement of PathBuilder + Path (reference forma/src/path.rs:776-925)."""

    def __init__(self):
        self._h = lib().oracle_path_new()
        self.affine = None  # GeomPresTransform as 6 floats ux,uy,vx,vy,tx,ty

    def __del__(self):
        try:
            lib().oracle_path_free(self._h)
        except Exception:
            pass

    def move_to(self, x, y):
        lib().oracle_path_move_to(self._h, x, y); return self

    def line_to(self, x, y):
        lib().oracle_path_line_to(self._h, x, y); return self

    def quad_to(self, ax, ay, bx, by):
        lib().oracle_path_quad_to(self._h, ax, ay, bx, by); return self

    def cubic_to(self, ax, ay, bx, by, cx, cy):
        lib().oracle_path_cubic_to(self._h, ax, ay, bx, by, cx, cy); return self

    def rat_quad_to(self, ax, ay, bx, by, w):
        lib().oracle_path_rat_quad_to(self._h, ax, ay, bx, by, w); return self

    def rat_cubic_to(self, ax, ay, bx, by, cx, cy, w1, w2):
        lib().oracle_path_rat_cubic_to(self._h, ax, ay, bx, by, cx, cy, w1, w2); return self

    def build(self):
        lib().oracle_path_close(self._h); return self

    def transform9(self, t9):
        t = np.ascontiguousarray(t9, dtype=np.float32)
        lib().oracle_path_transform9(self._h, _p(t)); return self

    def raw(self):
        n_cmds = C.c_size_t(0)
        n = lib().oracle_path_counts(self._h, C.byref(n_cmds))
        x = np.empty(n, np.float32); y = np.empty(n, np.float32); w = np.empty(n, np.float32)
        cmds = np.empty(n_cmds.value, np.uint8)
        lib().oracle_path_get(self._h, _p(x), _p(y), _p(w), _p(cmds))
        return x, y, w, cmds


class Primitives:
    """oracle restatement of `Primitives` (reference forma/src/path.rs:190-558), weights default 1."""

    def __init__(self):
        self._h = lib().oracle_prim_new()

    def __del__(self):
        try:
            lib().oracle_prim_free(self._h)
        except Exception:
            pass

    @staticmethod
    def _pts(pts):
        out = []
        for p in pts:
            out += [p[0], p[1], p[2] if len(p) > 2 else 1.0]
        return np.asarray(out, np.float32)

    def push_contour(self):
        lib().oracle_prim_push_contour(self._h); return self

    def push_line(self, *pts):
        a = self._pts(pts); lib().oracle_prim_push_line(self._h, _p(a)); return self

    def push_quad(self, *pts):
        a = self._pts(pts); lib().oracle_prim_push_quad(self._h, _p(a)); return self

    def push_cubic(self, *pts):
        a = self._pts(pts); lib().oracle_prim_push_cubic(self._h, _p(a)); return self


class Oracle:
    def __init__(self, threads: int = 1):
        self._h = lib().oracle_create()
        lib().oracle_set_threads(self._h, threads)
        self.n_points = 0

    def __del__(self):
        try:
            lib().oracle_destroy(self._h)
        except Exception:
            pass

    def set_threads(self, t):
        lib().oracle_set_threads(self._h, t)

    # ---- stage 1
    def flatten(self, path: Path):
        aff = None if path.affine is None else np.ascontiguousarray(path.affine, np.float32)
        n = lib().oracle_path_flatten(self._h, path._h, _p(aff))
        x = np.empty(n, np.float32); y = np.empty(n, np.float32); nc = np.empty(n, np.uint8)
        lib().oracle_flatten_get(self._h, _p(x), _p(y), _p(nc))
        return x, y, nc

    def flatten_primitives(self, prim: "Primitives"):
        n = lib().oracle_prim_flatten(self._h, prim._h)
        x = np.empty(n, np.float32); y = np.empty(n, np.float32); nc = np.empty(n, np.uint8)
        lib().oracle_flatten_get(self._h, _p(x), _p(y), _p(nc))
        return x, y, nc

    # ---- scene tables
    def set_geometry(self, x, y, line_slot):
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
        ls = np.ascontiguousarray(line_slot, np.uint32)
        assert len(x) == len(y) and len(ls) == max(len(x) - 1, 0)
        self.n_points = len(x)
        lib().oracle_set_geometry(self._h, _p(x), _p(y), _p(ls), len(x))

    def set_geoms(self, geoms):
        g = np.ascontiguousarray(geoms, GEOM_DTYPE)
        lib().oracle_set_geoms(self._h, _p(g), len(g))

    def set_styles(self, offsets, words, unchanged=None):
        o = np.ascontiguousarray(offsets, np.uint32); w = np.ascontiguousarray(words, np.uint32)
        u = None if unchanged is None else np.ascontiguousarray(unchanged, np.uint8)
        lib().oracle_set_styles(self._h, _p(o), len(o), _p(w), len(w), _p(u))

    def set_images(self, images, texels):
        im = np.ascontiguousarray(images, IMAGE_DTYPE)
        tx = np.ascontiguousarray(texels, np.uint16).reshape(-1, 4)
        lib().oracle_set_images(self._h, _p(im), len(im), _p(tx), len(tx))

    # ---- stages
    def prepare_lines(self, width, height):
        n = max(self.n_points - 1, 0)
        out = {k: np.zeros(n, np.float32) for k in ("x0", "y0", "dx", "dy", "a", "b", "c", "d")}
        out["orders"] = np.zeros(n, np.uint32); out["lengths"] = np.zeros(n, np.uint32)
        lib().oracle_prepare_lines(self._h, float(width), float(height), _p(out["orders"]), _p(out["x0"]), _p(out["y0"]),
                                   _p(out["dx"]), _p(out["dy"]), _p(out["a"]), _p(out["b"]), _p(out["c"]), _p(out["d"]),
                                   _p(out["lengths"]))
        return out

    def rasterize(self):
        n = lib().oracle_rasterize(self._h)
        out = np.empty(n, np.uint64)
        lib().oracle_get_segments(self._h, 0, _p(out))
        return out

    def sort(self):
        n = lib().oracle_sort(self._h)
        out = np.empty(n, np.uint64)
        lib().oracle_get_segments(self._h, 1, _p(out))
        return out

    def paint(self, segs, width, height, channels=(0, 1, 2, 3), clear=(1, 1, 1, 0), crop=None, cache_id=-1,
              dst=None, stride=None, dump_tiles=False):
        segs = np.ascontiguousarray(segs, np.uint64)
        stride = stride or width * 4
        if dst is None:
            dst = np.zeros((height, stride), np.uint8)
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear, np.float32)
        rect = None if crop is None else RectT(*crop)
        dump = None
        if dump_tiles:
            dump = np.zeros((((height + 15) // 16), ((width + 15) // 16), 256, 4), np.float32)
        rc = lib().oracle_paint(self._h, _p(segs), len(segs), _p(dst), width, height, stride, _p(ch), _p(cl),
                                None if rect is None else C.addressof(rect), cache_id, _p(dump))
        assert rc == 0
        return (dst, dump) if dump_tiles else dst

    def render(self, width, height, channels=(0, 1, 2, 3), clear=(1, 1, 1, 0), crop=None, cache_id=-1,
               dst=None, stride=None):
        stride = stride or width * 4
        if dst is None:
            dst = np.zeros((height, stride), np.uint8)
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear, np.float32)
        rect = None if crop is None else RectT(*crop)
        rc = lib().oracle_render(self._h, _p(dst), width, height, stride, _p(ch), _p(cl),
                                 None if rect is None else C.addressof(rect), cache_id, None)
        assert rc == 0
        return dst

    def segments(self, which):
        n = lib().oracle_last_n(self._h)
        out = np.empty(n, np.uint64)
        lib().oracle_get_segments(self._h, which, _p(out))
        return out

    def cache_clear(self, cache_id):
        lib().oracle_cache_clear(self._h, cache_id)

    def time_frame(self, width, height, iters=1):
        t = [C.c_double(0) for _ in range(4)]
        rc = lib().oracle_time_frame(self._h, width, height, iters, *[C.byref(v) for v in t])
        assert rc == 0
        return dict(prepare=t[0].value, rasterize=t[1].value, sort=t[2].value, paint=t[3].value)


# ---- field extractors of the packed u64 (reference cpu/pixel_segment.rs:90-138) ---------------
def seg_fields(v):
    v = np.asarray(v, np.uint64)
    cover = ((v & np.uint64(0x3F)).astype(np.int64) ^ 0x20) - 0x20
    dam = ((v >> np.uint64(6)) & np.uint64(0x3F)).astype(np.int64)
    return dict(
        tile_y=((v >> np.uint64(53)).astype(np.int64) - 1),
        tile_x=(((v >> np.uint64(41)) & np.uint64(0xFFF)).astype(np.int64) - 1),
        layer=((v >> np.uint64(20)) & np.uint64(0x1FFFFF)).astype(np.int64),
        local_x=((v >> np.uint64(16)) & np.uint64(0xF)).astype(np.int64),
        local_y=((v >> np.uint64(12)) & np.uint64(0xF)).astype(np.int64),
        double_area=dam * cover,
        cover=cover,
    )
