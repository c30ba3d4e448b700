"""Regenerates tests/golden/e2e_cpu_64x64.npz from the reference's CPU PNG goldens
(/root/reference/e2e-tests/expected/tests__*__cpu.png, 64x64 RGBA, written by the reference's own
CPU backend, e2e-tests/tests/test_env.rs:40-59,262-299).  Run in the build container only — the
GPU box has no /root/reference."""
import glob
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/e2e-tests/expected"
out = {}
for p in sorted(glob.glob(os.path.join(SRC, "tests__*__cpu.png"))):
    name = os.path.basename(p)[len("tests__"):-len("__cpu.png")]
    out[name] = np.asarray(Image.open(p).convert("RGBA"), np.uint8)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_cpu_64x64.npz"), **out)
print(len(out), "goldens")
