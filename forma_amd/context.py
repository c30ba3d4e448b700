"""Thin 1:1 Python wrapper of the C ABI (include/forma_hip.h) — numpy arrays in, numpy arrays out."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FormaError, RectT, TimingsT

GEOM_DTYPE = np.dtype([("order", "<u4"), ("flags", "<u4"), ("xf", "<f4", (6,))])
IMAGE_DTYPE = np.dtype([("texel_offset", "<u8"), ("width", "<u4"), ("height", "<u4")])


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One forma_hip_ctx (reference `cpu::Renderer`, cpu/renderer.rs:55-73): one GPU and one HIP stream, or — `devices` —
    several GPUs behind the same calls (forma_hip_create_multi: line-sharded rasterization, one RCCL all-to-all of pixel
    segments, band-local sort + paint, every device writing its rows of the caller's buffer)."""

    LAYOUTS = {"auto": 0, "exchange": 1, "bands": 2}

    def __init__(self, device: int = 0, devices=None, frames_in_flight: int = 1, layout=None):
        self._L = _lib.lib()
        h = C.c_void_p()
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = self._L.forma_hip_create_multi(C.byref(h), devs, len(devices))
            device = int(devices[0])
        else:
            rc = self._L.forma_hip_create(C.byref(h), device)
        if rc != 0:
            raise FormaError(rc, "forma_hip_create (needs a visible MI355X; there is no CPU fallback)")
        self._h = h
        self.device = device
        self.devices = None if devices is None else [int(d) for d in devices]
        self.n_points = 0
        if layout is not None:
            self.set_layout(layout)
        if frames_in_flight != 1:
            self.set_frames_in_flight(frames_in_flight)

    def set_layout(self, layout):
        """forma_hip_multi_layout: how a multi-device context splits a frame — 'exchange' (line shares + one all-to-all of
        pixel segments), 'bands' (no exchange: every device culls the scene to its band of tile rows) or 'auto'."""
        self._check(self._L.forma_hip_multi_layout(self._h, self.LAYOUTS[layout] if isinstance(layout, str) else int(layout)))

    def set_frames_in_flight(self, n: int):
        """forma_hip_set_frames_in_flight: device-resident, cache-less frames are enqueued on n frame slots; see sync()"""
        self._check(self._L.forma_hip_set_frames_in_flight(self._h, int(n)))

    def info(self):
        """devices, frame slots and — for a multi-device context — the exchange transport ('rccl' / 'copy')."""
        from ._lib import ContextInfoT
        ci = ContextInfoT()
        self._check(self._L.forma_hip_context_info(self._h, C.byref(ci)))
        return {"n_devices": int(ci.n_devices), "frames_in_flight": int(ci.frames_in_flight),
                "transport": {0: "none", 1: "rccl", 2: "copy"}.get(int(ci.transport), "?"),
                "layout": {0: "single", 1: "exchange", 2: "bands"}.get(int(ci.layout), "?") if ci.n_devices > 1 else "single",
                "devices": [int(ci.devices[i]) for i in range(ci.n_devices)]}

    def kernel_times(self):
        """forma_hip_kernel_times: the kernels of the last timed render call in launch order, each with the duration its own
        launch events measured: [(name, stage, start_us, us)]"""
        from ._lib import KernelTimeT
        arr = (KernelTimeT * 96)()
        n = C.c_size_t(0)
        self._check(self._L.forma_hip_kernel_times(self._h, arr, 96, C.byref(n)))
        return [(arr[i].name.decode(), int(arr[i].stage), float(arr[i].start_us), float(arr[i].us)) for i in range(min(n.value, 96))]

    def trim(self):
        """forma_hip_trim: give the per-frame device memory back (scene and caches stay)"""
        self._check(self._L.forma_hip_trim(self._h))

    def sync(self):
        """forma_hip_sync: wait for every enqueued frame, raise the first error one of them produced"""
        self._check(self._L.forma_hip_sync(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.forma_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise FormaError(rc, self._L.forma_hip_last_error(self._h).decode())

    # ---- scene upload
    def set_geometry(self, x, y, line_slot):
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
        ls = np.ascontiguousarray(line_slot, np.uint32)
        assert len(x) == len(y) and len(ls) == max(len(x) - 1, 0)
        self.n_points = len(x)
        self._check(self._L.forma_hip_set_geometry(self._h, _p(x), _p(y), _p(ls), len(x)))

    def set_geoms(self, geoms):
        g = np.ascontiguousarray(geoms, GEOM_DTYPE)
        self._check(self._L.forma_hip_set_geoms(self._h, _p(g), len(g)))

    def set_styles(self, offsets, words, unchanged=None):
        o = np.ascontiguousarray(offsets, np.uint32); w = np.ascontiguousarray(words, np.uint32)
        u = None if unchanged is None else np.ascontiguousarray(unchanged, np.uint8)
        self._check(self._L.forma_hip_set_styles(self._h, _p(o), len(o), _p(w), len(w), _p(u)))

    def set_images(self, images, texels):
        im = np.ascontiguousarray(images, IMAGE_DTYPE)
        tx = np.ascontiguousarray(texels, np.uint16).reshape(-1, 4)
        self._check(self._L.forma_hip_set_images(self._h, _p(im), len(im), _p(tx), len(tx)))

    # ---- stages
    def flatten_tables(self, t):
        """forma_hip_flatten on caller-supplied work items (the fields of forma_flatten_tables_t as numpy arrays + n_points,
        n_quads, n_splines): the parallel map of Primitives::into_segments, path.rs:487-534"""
        from ._lib import FlattenTablesT
        ft = FlattenTablesT()
        keep = []
        for name, _ in FlattenTablesT._fields_:
            if name.startswith("n_"):
                setattr(ft, name, int(t[name]))
            else:
                a = np.ascontiguousarray(t[name]); keep.append(a)
                setattr(ft, name, a.ctypes.data)
        n = int(t["n_points"])
        x = np.zeros(max(n, 1), np.float32); y = np.zeros(max(n, 1), np.float32)
        self._check(self._L.forma_hip_flatten(self._h, C.byref(ft), _p(x), _p(y)))
        return x[:n], y[:n]

    def prepare_lines(self, width, height):
        n = max(self.n_points - 1, 0)
        out = {k: np.zeros(n, np.float32) for k in ("x0", "y0", "dx", "dy", "a", "b", "c", "d")}
        out["orders"] = np.zeros(n, np.uint32); out["lengths"] = np.zeros(n, np.uint32)
        self._check(self._L.forma_hip_prepare_lines(self._h, width, height, _p(out["orders"]), _p(out["x0"]), _p(out["y0"]),
                                                    _p(out["dx"]), _p(out["dy"]), _p(out["a"]), _p(out["b"]), _p(out["c"]),
                                                    _p(out["d"]), _p(out["lengths"])))
        return out

    def rasterize_lines(self, lines):
        n = len(lines["lengths"])
        N = int(lines["lengths"][-1]) if n else 0
        out = np.zeros(N, np.uint64); got = C.c_size_t(0)
        arr = {k: np.ascontiguousarray(v) for k, v in lines.items()}
        self._check(self._L.forma_hip_rasterize(self._h, n, _p(arr["orders"]), _p(arr["x0"]), _p(arr["y0"]), _p(arr["dx"]),
                                                _p(arr["dy"]), _p(arr["a"]), _p(arr["b"]), _p(arr["c"]), _p(arr["d"]),
                                                _p(arr["lengths"]), _p(out), N, C.byref(got)))
        assert got.value == N
        return out

    def sort_array(self, segs, digit_bits=0):
        v = np.ascontiguousarray(segs, np.uint64).copy()
        self._check(self._L.forma_hip_sort(self._h, _p(v), len(v), digit_bits))
        return v

    def paint(self, segs, width, height, channels=(0, 1, 2, 3), clear=(1, 1, 1, 0), crop=None, dst=None, stride=None):
        segs = np.ascontiguousarray(segs, np.uint64)
        stride = stride or width * 4
        if dst is None:
            dst = np.zeros((height, stride), np.uint8)
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear, np.float32)
        rect = None if crop is None else RectT(*crop)
        self._check(self._L.forma_hip_paint(self._h, _p(segs), len(segs), _p(dst), width, height, stride, _p(ch), _p(cl),
                                            None if rect is None else C.addressof(rect)))
        return dst

    def render(self, width, height, channels=(0, 1, 2, 3), clear=(1, 1, 1, 0), crop=None, cache_id=-1, dst=None,
               stride=None, timings=False, device_only=False):
        stride = stride or width * 4
        if dst is None and not device_only:
            dst = np.zeros((height, stride), np.uint8)
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear, np.float32)
        rect = None if crop is None else RectT(*crop)
        t = TimingsT() if timings else None
        self._check(self._L.forma_hip_render(self._h, None if device_only else _p(dst), width, height, stride, _p(ch), _p(cl),
                                             None if rect is None else C.addressof(rect), cache_id,
                                             None if t is None else C.addressof(t)))
        if timings:
            return dst, t.as_dict()
        return dst

    def render_enqueue(self, width, height, dst, channels=(0, 1, 2, 3), clear=(1, 1, 1, 0), crop=None, stride=None):
        """forma_hip_render_enqueue: the frame AND its copy into `dst` are enqueued on the next frame slot; `dst` is complete after
        sync() or after frames_in_flight further enqueues"""
        stride = stride or width * 4
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear, np.float32)
        rect = None if crop is None else RectT(*crop)
        self._check(self._L.forma_hip_render_enqueue(self._h, _p(dst), width, height, stride, _p(ch), _p(cl),
                                                     None if rect is None else C.addressof(rect)))

    def register_buffer(self, arr):
        """forma_hip_register_buffer: page-lock a numpy buffer the renderer writes often (keep it alive until unregister / close)"""
        self._check(self._L.forma_hip_register_buffer(self._h, _p(arr), arr.nbytes))

    def unregister_buffer(self, arr):
        self._check(self._L.forma_hip_unregister_buffer(self._h, _p(arr)))

    def cache_clear(self, cache_id):                    # BufferLayerCache::clear
        self._check(self._L.forma_hip_cache_clear(self._h, cache_id))

    def segments(self, which):
        n = C.c_size_t(0)
        rc = self._L.forma_hip_read_segments(self._h, which, None, 0, C.byref(n))
        if n.value == 0:
            self._check(rc)
            return np.zeros(0, np.uint64)
        out = np.zeros(n.value, np.uint64)
        self._check(self._L.forma_hip_read_segments(self._h, which, _p(out), n.value, C.byref(n)))
        return out

    def tiles_written(self, width, height):
        """one byte per tile of the last frame: 1 = the frame wrote the tile (forma_hip_tiles_written)"""
        n = ((width + 15) // 16) * ((height + 15) // 16)
        flags = np.zeros(n, np.uint8)
        self._check(self._L.forma_hip_tiles_written(self._h, _p(flags), n))
        return flags

    def read_image(self, width, height):
        dst = np.zeros((height, width * 4), np.uint8)
        self._check(self._L.forma_hip_read_image(self._h, _p(dst), width * 4))
        return dst

    # ---- multi-GPU
    def set_band(self, row0, row1):
        self._check(self._L.forma_hip_set_band(self._h, row0, row1))

    def rasterize_frame(self, width, height, timings=False):
        t = TimingsT() if timings else None
        self._check(self._L.forma_hip_rasterize_frame(self._h, width, height, None if t is None else C.addressof(t)))
        return t.as_dict() if timings else None

    def segments_device(self, which):
        p = C.c_void_p(); n = C.c_size_t(0)
        self._check(self._L.forma_hip_segments_device(self._h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def reserve_segments(self, n):
        p = C.c_void_p()
        self._check(self._L.forma_hip_reserve_segments(self._h, n, C.byref(p)))
        return p.value

    def device_view(self, ptr, n):
        """torch int64 view (no copy) of n u64 at device address `ptr` of this context's GPU, for torch.distributed."""
        import torch

        class _Span:
            pass
        sp = _Span()
        sp.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(sp, device=torch.device("cuda", self.device))

    def unsorted_view(self):
        ptr, n = self.segments_device(0)
        return self.device_view(ptr, n) if n else None

    def reserve_view(self, n):
        return self.device_view(self.reserve_segments(max(int(n), 1)), max(int(n), 1))[: int(n)]

    def sort_paint_frame(self, n, width, height, channels=(0, 1, 2, 3), clear=(1, 1, 1, 0), crop=None, dst=None,
                         stride=None, timings=False, device_only=True):
        stride = stride or width * 4
        if dst is None and not device_only:
            dst = np.zeros((height, stride), np.uint8)
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear, np.float32)
        rect = None if crop is None else RectT(*crop)
        t = TimingsT() if timings else None
        self._check(self._L.forma_hip_sort_paint_frame(self._h, n, None if device_only else _p(dst), width, height, stride,
                                                       _p(ch), _p(cl), None if rect is None else C.addressof(rect),
                                                       None if t is None else C.addressof(t)))
        return (dst, t.as_dict()) if timings else dst

    # ---- multi-GPU, exchange layout (include/forma_hip.h, last section)
    def stream_handle(self) -> int:
        p = C.c_void_p()
        self._check(self._L.forma_hip_stream(self._h, C.byref(p)))
        return p.value or 0

    def exchange_plan(self, row_edges, pair_capacity):
        e = np.ascontiguousarray(row_edges, np.uint32)
        self._check(self._L.forma_hip_exchange_plan(self._h, _p(e), len(e) - 1, int(pair_capacity)))
        self._xplan = (len(e) - 1, int(pair_capacity))

    def exchange_views(self):
        """torch views (no copies) of the send / receive buckets (int64, n_ranks * words_per_pair: a bucket is pair_capacity
        segments + its header word {count | overflow << 32}) and words_per_pair"""
        n, _cap = self._xplan
        ps, pr, w = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        self._check(self._L.forma_hip_exchange_buffers(self._h, C.byref(ps), C.byref(pr), C.byref(w)))
        return self.device_view(ps.value, n * w.value), self.device_view(pr.value, n * w.value), int(w.value)

    def rasterize_bucket_frame(self, width, height, timings=False):
        t = TimingsT() if timings else None
        self._check(self._L.forma_hip_rasterize_bucket_frame(self._h, width, height, None if t is None else C.addressof(t)))
        return t.as_dict() if timings else None

    def gather_sort_paint_frame(self, width, height, channels=(0, 1, 2, 3), clear=(1, 1, 1, 0), crop=None, dst=None, stride=None,
                                timings=False, device_only=True):
        stride = stride or width * 4
        if dst is None and not device_only:
            dst = np.zeros((height, stride), np.uint8)
        ch = np.asarray(channels, np.uint8); cl = np.asarray(clear, np.float32)
        rect = None if crop is None else RectT(*crop)
        t = TimingsT() if timings else None
        self._check(self._L.forma_hip_gather_sort_paint_frame(self._h, None if device_only else _p(dst), width, height, stride,
                                                              _p(ch), _p(cl), None if rect is None else C.addressof(rect),
                                                              None if t is None else C.addressof(t)))
        return (dst, t.as_dict()) if timings else dst
