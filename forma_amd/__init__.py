"""forma_amd — MI355X (gfx950) backend for forma's 4-stage raster pipeline.

Layout: `csrc/` hand-written HIP kernels + the C ABI (include/forma_hip.h), `_lib.py` the ctypes
loader, `context.py` a thin 1:1 wrapper of the C ABI for tests/bench, `scenes.py` the synthetic
workloads of BASELINE.json.  No CPU fallback and no dependency on oracle/.
"""
from ._lib import FormaError, NONE, SO_PATH, build, lib  # noqa: F401
from .context import Context  # noqa: F401
