"""Oracle vs the reference's 32 CPU e2e goldens (e2e-tests/expected/*__cpu.png, tolerance 8 in
e2e-tests/tests/test_env.rs:278).  28 of them are reproduced bit-exactly; the four `clip_color!`
blend modes differ by <= 5 because the reference build used AVX `rcp_ps`."""
import os

import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_cpu_64x64.npz"))
RCP_MODES = {"blend_modes__Hue": 4, "blend_modes__Saturation": 5, "blend_modes__Color": 3, "blend_modes__Luminosity": 3}
SCENES = S.e2e_scenes()


def test_all_goldens_have_scenes():
    assert sorted(GOLD.files) == sorted(SCENES)


@pytest.mark.parametrize("name", sorted(SCENES))
def test_e2e_golden(name):
    o = orc.Oracle()
    S.load(o, SCENES[name].tables(o))
    got = o.render(64, 64, channels=S.RGBA, clear=(1, 1, 1, 0)).reshape(64, 64, 4)
    d = np.abs(got.astype(int) - GOLD[name].astype(int))
    assert d.max() <= RCP_MODES.get(name, 0), (name, d.max(), int((d > 0).sum()))
