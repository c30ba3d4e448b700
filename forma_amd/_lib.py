"""ctypes loader of the in-tree libforma_hip.so (hand-written HIP for gfx950).

The product path has no CPU fallback: if the shared library is missing or no MI355X is visible,
construction of a context raises.  Nothing in this package imports `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# FORMA_HIP_LIB: another build of the library (tools/ab_fast.py A/B runs of kernel variants on one box); default = the in-tree one
SO_PATH = os.environ.get("FORMA_HIP_LIB") or os.path.join(CSRC, "libforma_hip.so")

NONE = 0xFFFFFFFF

ERRORS = {0: "FORMA_OK", -1: "FORMA_E_ARG", -2: "FORMA_E_HIP", -3: "FORMA_E_NO_DEVICE", -4: "FORMA_E_CAPACITY",
          -5: "FORMA_E_STATE", -6: "FORMA_E_INTERNAL", -7: "FORMA_E_COMM"}


class FormaError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")


class GeomT(C.Structure):
    _fields_ = [("order", C.c_uint32), ("flags", C.c_uint32), ("xf", C.c_float * 6)]


class ImageT(C.Structure):
    _fields_ = [("texel_offset", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32)]


class RectT(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("x1", C.c_uint32), ("y0", C.c_uint32), ("y1", C.c_uint32)]


class TimingsT(C.Structure):
    _fields_ = [("prepare_us", C.c_float), ("rasterize_us", C.c_float), ("sort_us", C.c_float), ("sort_pass_us", C.c_float),
                ("carry_us", C.c_float), ("paint_us", C.c_float), ("total_us", C.c_float), ("d2h_us", C.c_float),
                ("n_lines", C.c_uint32), ("n_segments", C.c_uint32), ("n_sort_passes", C.c_uint32), ("n_runs", C.c_uint32),
                ("n_tile_entries", C.c_uint32), ("n_tiles_written", C.c_uint32), ("exchange_us", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ContextInfoT(C.Structure):
    _fields_ = [("n_devices", C.c_uint32), ("frames_in_flight", C.c_uint32), ("transport", C.c_uint32), ("layout", C.c_uint32),
                ("devices", C.c_int32 * 8)]


class KernelTimeT(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("start_us", C.c_float), ("us", C.c_float), ("stage", C.c_uint32), ("reserved", C.c_uint32)]


class SortPlanT(C.Structure):
    _fields_ = [("n_passes", C.c_uint32), ("biased", C.c_uint32), ("shift", C.c_uint32 * 12), ("mask", C.c_uint32 * 12), ("bias", C.c_uint32 * 12)]


class FlattenTablesT(C.Structure):
    _fields_ = [("point_commands", C.c_void_p), ("point_indices", C.c_void_p), ("quad_indices", C.c_void_p), ("n_points", C.c_size_t),
                ("qx", C.c_void_p), ("qy", C.c_void_p), ("qw", C.c_void_p), ("x0", C.c_void_p), ("dx_recip", C.c_void_p),
                ("k0", C.c_void_p), ("dk", C.c_void_p), ("curvatures_recip", C.c_void_p), ("partial_spline", C.c_void_p),
                ("partial_curv", C.c_void_p), ("n_quads", C.c_size_t), ("sp0x", C.c_void_p), ("sp0y", C.c_void_p),
                ("sp2x", C.c_void_p), ("sp2y", C.c_void_p), ("n_splines", C.c_size_t)]


# every symbol include/forma_hip.h declares: (name, restype, argtypes)
_vp, _sz, _u32, _i = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
SYMBOLS = {
    "forma_hip_create": (_i, [C.POINTER(_vp), _i]),
    "forma_hip_create_multi": (_i, [C.POINTER(_vp), C.POINTER(_i), _i]),
    "forma_hip_destroy": (None, [_vp]),
    "forma_hip_last_error": (C.c_char_p, [_vp]),
    "forma_hip_version": (C.c_char_p, []),
    "forma_hip_set_geometry": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "forma_hip_set_geoms": (_i, [_vp, _vp, _sz]),
    "forma_hip_set_styles": (_i, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "forma_hip_set_images": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "forma_hip_flatten": (_i, [_vp, _vp, _vp, _vp]),
    "forma_hip_prepare_lines": (_i, [_vp, _u32, _u32] + [_vp] * 10),
    "forma_hip_rasterize": (_i, [_vp, _sz] + [_vp] * 10 + [_vp, _sz, _vp]),
    "forma_hip_sort": (_i, [_vp, _vp, _sz, _i]),
    "forma_hip_paint": (_i, [_vp, _vp, _sz, _vp, _u32, _u32, _sz, _vp, _vp, _vp]),
    "forma_hip_render": (_i, [_vp, _vp, _u32, _u32, _sz, _vp, _vp, _vp, _i, _vp]),
    "forma_hip_render_enqueue": (_i, [_vp, _vp, _u32, _u32, _sz, _vp, _vp, _vp]),
    "forma_hip_register_buffer": (_i, [_vp, _vp, _sz]),
    "forma_hip_unregister_buffer": (_i, [_vp, _vp]),
    "forma_hip_cache_clear": (_i, [_vp, _i]),
    "forma_hip_set_frames_in_flight": (_i, [_vp, _i]),
    "forma_hip_multi_layout": (_i, [_vp, _i]),
    "forma_hip_sync": (_i, [_vp]),
    "forma_hip_context_info": (_i, [_vp, _vp]),
    "forma_hip_sort_plan": (_i, [C.c_uint64, _i, _i, _vp, _vp]),
    "forma_hip_kernel_times": (_i, [_vp, _vp, _sz, _vp]),
    "forma_hip_trim": (_i, [_vp]),
    "forma_hip_read_segments": (_i, [_vp, _i, _vp, _sz, _vp]),
    "forma_hip_read_image": (_i, [_vp, _vp, _sz]),
    "forma_hip_tiles_written": (_i, [_vp, _vp, _sz]),
    "forma_hip_set_band": (_i, [_vp, _u32, _u32]),
    "forma_hip_segments_device": (_i, [_vp, _i, _vp, _vp]),
    "forma_hip_rasterize_frame": (_i, [_vp, _u32, _u32, _vp]),
    "forma_hip_reserve_segments": (_i, [_vp, _sz, _vp]),
    "forma_hip_sort_paint_frame": (_i, [_vp, _sz, _vp, _u32, _u32, _sz, _vp, _vp, _vp, _vp]),
    "forma_hip_stream": (_i, [_vp, _vp]),
    "forma_hip_exchange_plan": (_i, [_vp, _vp, _u32, _u32]),
    "forma_hip_exchange_buffers": (_i, [_vp, _vp, _vp, _vp]),
    "forma_hip_rasterize_bucket_frame": (_i, [_vp, _u32, _u32, _vp]),
    "forma_hip_gather_sort_paint_frame": (_i, [_vp, _vp, _u32, _u32, _sz, _vp, _vp, _vp, _vp]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "forma_hip.h"))
    stale = force or not os.path.exists(SO_PATH) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC, "-j4", "-s"])
    return SO_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise FormaError(-2, f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                 f"(there is no CPU fallback)")
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("FORMA_HIP_LIB") and not hasattr(L, name):
                continue                                  # (tools/ab_fast.py: an older build of the library, for A/B timing only)
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
