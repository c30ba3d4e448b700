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
FLUSH_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_void_p)   # Flusher::flush(slice) (buffer/layout/mod.rs:29-34)


def _flush_cb(flusher):
    """ctypes trampoline: `flusher(view)` gets a writable uint8 numpy view of the row slice."""
    def cb(ptr, n, _user):
        flusher(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)))
    return FLUSH_FN(cb)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, f, i32, u32, sz, u8p = C.c_void_p, C.c_float, C.c_int, C.c_uint32, C.c_size_t, C.c_void_p
        L.oracle_create.restype = vp
        L.oracle_destroy.argtypes = [vp]
        L.oracle_set_threads.argtypes = [vp, i32]
        L.oracle_set_stage_threads.argtypes = [vp, i32, i32, i32, i32]
        L.oracle_set_stage_threads.restype = None
        L.oracle_set_optimizer.argtypes = [vp, i32]
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
        L.oracle_texture_column.argtypes = [vp, vp, sz, vp, f, f, vp]
        L.oracle_point_angle.argtypes = [f, f, vp]; L.oracle_point_angle.restype = i32
        L.oracle_psi_new.argtypes = [vp, sz]; L.oracle_psi_new.restype = vp
        L.oracle_psi_free.argtypes = [vp]
        L.oracle_psi_next.argtypes = [vp, vp, vp]; L.oracle_psi_next.restype = i32
        L.oracle_psi_next_back.argtypes = [vp, vp, vp]; L.oracle_psi_next_back.restype = i32
        L.oracle_psi_len.argtypes = [vp]; L.oracle_psi_len.restype = u32
        L.oracle_psi_split_at.argtypes = [vp, sz]; L.oracle_psi_split_at.restype = vp
        L.oracle_paint.argtypes = [vp, vp, sz, vp, u32, u32, sz, vp, vp, vp, i32, vp]
        L.oracle_paint.restype = i32
        L.oracle_cache_clear.argtypes = [vp, i32]
        L.oracle_render.argtypes = [vp, vp, u32, u32, sz, vp, vp, vp, i32, vp]
        L.oracle_render.restype = i32
        L.oracle_last_n.argtypes = [vp]
        L.oracle_last_n.restype = sz
        L.oracle_time_frame.argtypes = [vp, u32, u32, i32, vp, vp, vp, vp]
        L.oracle_time_frame.restype = i32
        L.oracle_paint_flush.argtypes = [vp, vp, sz, vp, u32, u32, sz, vp, vp, vp, i32, FLUSH_FN, vp]
        L.oracle_paint_flush.restype = i32
        L.oracle_render_flush.argtypes = [vp, vp, u32, u32, sz, vp, vp, vp, i32, FLUSH_FN, vp]
        L.oracle_render_flush.restype = i32
        L.oracle_wb_new.argtypes = [vp]; L.oracle_wb_new.restype = vp
        L.oracle_wb_free.argtypes = [vp]
        L.oracle_wb_init.argtypes = [vp, vp, vp, sz]
        L.oracle_wb_cached_tile_set.argtypes = [vp, i32, i32, u32, i32, vp]
        L.oracle_wb_cached_tile_get.argtypes = [vp, vp, vp, vp, vp]
        L.oracle_wb_context.argtypes = [vp, u32, u32, vp, sz, i32, vp, vp, vp]
        L.oracle_wb_populate.argtypes = [vp]
        L.oracle_wb_next_tile.argtypes = [vp]
        L.oracle_wb_pass.argtypes = [vp, i32, vp]; L.oracle_wb_pass.restype = i32
        L.oracle_wb_drive.argtypes = [vp, vp]; L.oracle_wb_drive.restype = i32
        L.oracle_wb_colors.argtypes = [vp, vp]
        L.oracle_wb_ids.argtypes = [vp, vp, sz, i32]; L.oracle_wb_ids.restype = sz
        L.oracle_wb_skip_clipping.argtypes = [vp, u32]; L.oracle_wb_skip_clipping.restype = i32
        L.oracle_wb_seg_range.argtypes = [vp, u32, vp, vp]; L.oracle_wb_seg_range.restype = i32
        L.oracle_wb_queue_index.argtypes = [vp, u32]; L.oracle_wb_queue_index.restype = i32
        L.oracle_wb_queue.argtypes = [vp, vp, vp, sz]; L.oracle_wb_queue.restype = sz
        L.oracle_cover_is_empty.argtypes = [vp, i32]; L.oracle_cover_is_empty.restype = i32
        L.oracle_cover_is_full.argtypes = [vp, i32]; L.oracle_cover_is_full.restype = i32
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

    def tables(self):
        """populate_buffers (path.rs:400-445) + the per-quad / per-spline arrays: the fields of the C ABI's
        forma_flatten_tables_t, as a dict of numpy arrays (what a host hands to forma_hip_flatten)"""
        L = lib()
        L.oracle_prim_tables.restype = C.c_size_t
        L.oracle_prim_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 19
        nq, ns = C.c_size_t(0), C.c_size_t(0)
        n = L.oracle_prim_tables(self._h, C.byref(nq), C.byref(ns), *([None] * 17))
        nq, ns = nq.value, ns.value
        u = lambda k: np.zeros(max(k, 1), np.uint32)
        f = lambda k: np.zeros(max(k, 1), np.float32)
        t = {"point_commands": u(n), "point_indices": u(n), "quad_indices": u(n), "qx": f(3 * nq), "qy": f(3 * nq), "qw": f(3 * nq),
             "x0": f(nq), "dx_recip": f(nq), "k0": f(nq), "dk": f(nq), "curvatures_recip": f(nq), "partial_spline": u(nq),
             "partial_curv": f(nq), "sp0x": f(ns), "sp0y": f(ns), "sp2x": f(ns), "sp2y": f(ns)}
        order = ["point_commands", "point_indices", "quad_indices", "qx", "qy", "qw", "x0", "dx_recip", "k0", "dk", "curvatures_recip",
                 "partial_spline", "partial_curv", "sp0x", "sp0y", "sp2x", "sp2y"]
        L.oracle_prim_tables(self._h, None, None, *[_p(t[k]) for k in order])
        t["n_points"], t["n_quads"], t["n_splines"] = int(n), nq, ns
        return t


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

    def set_optimizer(self, enabled: bool):
        """test switch: with False every tile is painted layer by layer, no optimizer pass (layer_workbench/mod.rs:236-248 skipped)"""
        lib().oracle_set_optimizer(self._h, 1 if enabled else 0)

    def set_threads(self, t):
        lib().oracle_set_threads(self._h, t)

    def set_stage_threads(self, prepare=0, rasterize=0, sort=0, paint=0):
        """time_frame only: a thread count per stage (0 = the frame's)"""
        lib().oracle_set_stage_threads(self._h, int(prepare), int(rasterize), int(sort), int(paint))

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
              dst=None, stride=None, dump_tiles=False, flusher=None):
        segs = np.ascontiguousarray(segs, np.uint64)
        stride = stride or width * 4
        if dst is None:
            dst = np.zeros((height, stride), np.uint8)
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear, np.float32)
        rect = None if crop is None else RectT(*crop)
        dump = None
        if dump_tiles:
            dump = np.zeros((((height + 15) // 16), ((width + 15) // 16), 256, 4), np.float32)
        if flusher is not None:
            cb = _flush_cb(flusher)
            rc = lib().oracle_paint_flush(self._h, _p(segs), len(segs), _p(dst), width, height, stride, _p(ch), _p(cl),
                                          None if rect is None else C.addressof(rect), cache_id, cb, None)
        else:
            rc = lib().oracle_paint(self._h, _p(segs), len(segs), _p(dst), width, height, stride, _p(ch), _p(cl),
                                    None if rect is None else C.addressof(rect), cache_id, _p(dump))
        assert rc == 0
        return (dst, dump) if dump_tiles else dst

    def render(self, width, height, channels=(0, 1, 2, 3), clear=(1, 1, 1, 0), crop=None, cache_id=-1,
               dst=None, stride=None, flusher=None):
        stride = stride or width * 4
        if dst is None:
            dst = np.zeros((height, stride), np.uint8)
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear, np.float32)
        rect = None if crop is None else RectT(*crop)
        if flusher is not None:
            cb = _flush_cb(flusher)
            rc = lib().oracle_render_flush(self._h, _p(dst), width, height, stride, _p(ch), _p(cl),
                                           None if rect is None else C.addressof(rect), cache_id, cb, None)
        else:
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


class Workbench:
    """Harness over the oracle's LayerWorkbench + Painter restatement (reference
    cpu/painter/layer_workbench/mod.rs:147-342, passes/*.rs, cpu/painter/mod.rs:232-483) so that the reference's own
    unit tests of those files can be replayed against the oracle.  Layer props / is_unchanged come from the owning
    Oracle's style table (`Oracle.set_styles(offsets, words, unchanged)`)."""
    CONTINUE, BREAK_NONE, BREAK_SOLID = 0, 1, 2          # ControlFlow<OptimizerTileWriteOp>
    OP_NONE, OP_SOLID, OP_COLOR_BUFFER = 0, 1, 2         # TileWriteOp

    def __init__(self, oracle: "Oracle"):
        self._o = oracle
        self._h = lib().oracle_wb_new(oracle._h)

    def __del__(self):
        try:
            lib().oracle_wb_free(self._h)
        except Exception:
            pass

    def init(self, carries):
        """carries: [(layer_id, 16 x i8 cover)]"""
        layers = np.asarray([c[0] for c in carries], np.uint32)
        covers = np.asarray([c[1] for c in carries], np.int8).reshape(-1, 16) if carries else np.zeros((0, 16), np.int8)
        covers = np.ascontiguousarray(covers)
        lib().oracle_wb_init(self._h, _p(layers), _p(covers), len(layers))

    def cached_tile(self, use=True, layer_count=None, solid_color=None):
        sc = np.asarray(solid_color if solid_color is not None else [0, 0, 0, 0], np.uint8)
        lib().oracle_wb_cached_tile_set(self._h, int(use), int(layer_count is not None), int(layer_count or 0),
                                        int(solid_color is not None), _p(sc))

    def cached_tile_state(self):
        has_lc, lc, has_sc = C.c_int(0), C.c_uint32(0), C.c_int(0)
        sc = np.zeros(4, np.uint8)
        lib().oracle_wb_cached_tile_get(self._h, C.byref(has_lc), C.byref(lc), C.byref(has_sc), _p(sc))
        return (lc.value if has_lc.value else None), (sc.tolist() if has_sc.value else None)

    def context(self, segments=(), tile_x=0, tile_y=0, cached_clear_color=None, channels=(0, 1, 2, 3), clear_color=(0, 0, 0, 0)):
        segs = np.ascontiguousarray(segments, np.uint64)
        cc = np.asarray(cached_clear_color if cached_clear_color is not None else [0, 0, 0, 0], np.float32)
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear_color, np.float32)
        lib().oracle_wb_context(self._h, tile_x, tile_y, _p(segs), len(segs), int(cached_clear_color is not None), _p(cc), _p(ch), _p(cl))

    def populate_layers(self):
        lib().oracle_wb_populate(self._h)

    def next_tile(self):
        lib().oracle_wb_next_tile(self._h)

    def _pass(self, which):
        solid = np.zeros(4, np.float32)
        r = lib().oracle_wb_pass(self._h, which, _p(solid))
        return (r, tuple(float(v) for v in solid)) if r == 2 else (r, None)

    def tile_unchanged_pass(self):
        return self._pass(0)

    def skip_trivial_clips_pass(self):
        return self._pass(1)

    def skip_fully_covered_layers_pass(self):
        return self._pass(2)

    def drive_tile_painting(self):
        solid = np.zeros(4, np.uint8)
        op = lib().oracle_wb_drive(self._h, _p(solid))
        return (op, solid.tolist()) if op == 1 else (op, None)

    def colors(self):
        out = np.zeros((256, 4), np.float32)
        lib().oracle_wb_colors(self._h, _p(out))
        return out

    def ids(self, masked=True):
        out = np.zeros(4096, np.uint32)
        n = lib().oracle_wb_ids(self._h, _p(out), len(out), int(masked))
        return out[:n].tolist()

    def skip_clipping_contains(self, layer_id):
        return bool(lib().oracle_wb_skip_clipping(self._h, layer_id))

    def segment_range(self, layer_id):
        lo, hi = C.c_size_t(0), C.c_size_t(0)
        ok = lib().oracle_wb_seg_range(self._h, layer_id, C.byref(lo), C.byref(hi))
        return (lo.value, hi.value) if ok else None

    def queue_index(self, layer_id):
        i = lib().oracle_wb_queue_index(self._h, layer_id)
        return None if i < 0 else i

    def queue(self):
        layers = np.zeros(4096, np.uint32); covers = np.zeros((4096, 16), np.int8)
        n = lib().oracle_wb_queue(self._h, _p(layers), _p(covers), 4096)
        return [(int(layers[i]), covers[i].tolist()) for i in range(n)]


def cover_is_empty(cover16, even_odd=False):
    c = np.asarray(cover16, np.int8)
    return bool(lib().oracle_cover_is_empty(_p(c), int(even_odd)))


def cover_is_full(cover16, even_odd=False):
    c = np.asarray(cover16, np.int8)
    return bool(lib().oracle_cover_is_full(_p(c), int(even_odd)))


def pixel_segment(layer, tile_x, tile_y, local_x, local_y, dam, cover):
    """PixelSegment::new (reference cpu/pixel_segment.rs:36-71)"""
    return int(lib().oracle_pixel_segment_new(layer, tile_x, tile_y, local_x, local_y, dam, cover))


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
