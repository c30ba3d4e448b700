"""examples/render_e2e.c — a COMPILED consumer of the C ABI (C11, gcc, links libforma_hip.so; no Python or ctypes in the process
that renders).  The closest stand-in this image allows for SURVEY §8 f2 ("the reference-side binding actually links"): the
program builds the scene tables of the reference's polygon-only e2e scenes (e2e-tests/tests/tests.rs) itself and renders them
as e2e-tests/tests/test_env.rs:40-59 does (RGBA, clear {1, 1, 1, 0}).

CPU: it compiles against include/forma_hip.h without warnings, links, and without a GPU ends with FORMA_E_NO_DEVICE (-3) —
no fallback.  GPU: every image it writes equals the oracle's for the same scene built by tests/scene.py (an independent
construction of the tables) and lies within the reference's tolerance of the CPU golden PNG."""
import os
import subprocess

import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "render_e2e.c")
NAMES = ["linear_gradient", "solid_color__red", "solid_color__transparent_black", "pixel", "fill_rules__EvenOdd",
         "fill_rules__NonZero", "covers", "blend_modes__Multiply"]


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    import forma_amd
    forma_amd.build()                                                    # (libforma_hip.so in-tree; a no-op when it is current)
    out = str(tmp_path_factory.mktemp("example_c") / "render_e2e")
    libdir = os.path.join(ROOT, "forma_amd", "csrc")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + libdir,
           "-lforma_hip", "-Wl,-rpath," + libdir, "-o", out]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return out


def test_example_compiles_links_and_refuses_without_a_gpu(binary, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the GPU test runs the program")
    p = subprocess.run([binary, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "forma_hip_create" in p.stderr and "-3" in p.stderr, (p.returncode, p.stderr)
    assert not os.listdir(tmp_path)                                      # nothing rendered: there is no CPU fallback


@pytest.mark.gpu
def test_example_images_match_oracle_and_goldens(binary, tmp_path):
    p = subprocess.run([binary, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout, p.stderr)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "e2e_cpu_64x64.npz"))
    scenes = S.e2e_scenes()
    o = orc.Oracle()
    for name in NAMES:
        got = np.fromfile(os.path.join(str(tmp_path), name + ".rgba"), np.uint8).reshape(64, 64 * 4)
        S.load(o, scenes[name].tables(o))
        want = o.render(64, 64)                                          # clear (1, 1, 1, 0), RGBA: the oracle's defaults = test_env.rs
        d = np.abs(got.astype(int) - want.reshape(64, 256).astype(int))
        assert d.max() == 0, (name, "differs from the oracle", int(d.max()), int((d > 0).sum()))
        dg = np.abs(got.reshape(64, 64, 4).astype(int) - gold[name].astype(int))
        assert dg.max() <= 8, (name, int(dg.max()))                      # e2e-tests/tests/test_env.rs:278
    assert "ok" in p.stdout.splitlines()[-1]
