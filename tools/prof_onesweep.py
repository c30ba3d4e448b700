import subprocess, sys, ctypes
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
from forma_amd import _lib
_lib.lib().forma_debug_dump_prof()
