"""CPU-only checks (no GPU needed): the C-ABI library loads and exports every symbol include/forma_hip.h declares,
the product path fails loudly without a GPU, and the host-side mirror of forma's API behaves like the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "forma_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(forma_hip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_symbol():
    from forma_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    assert sorted(_lib.SYMBOLS) == names, "ctypes binding table and header disagree"
    L = C.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/forma_hip.h but not exported by libforma_hip.so"
    L.forma_hip_version.restype = C.c_char_p
    assert b"gfx950" in L.forma_hip_version()


def test_no_cpu_fallback():
    """Without a visible MI355X the product path raises; it never computes on the CPU."""
    import forma_amd
    from forma_amd._lib import FormaError
    L = forma_amd.lib()
    h = C.c_void_p()
    rc = L.forma_hip_create(C.byref(h), 0)
    if rc == 0:                      # a GPU is present: nothing to check here
        L.forma_hip_destroy(h)
        pytest.skip("GPU present")
    assert rc == -3                  # FORMA_E_NO_DEVICE
    with pytest.raises(FormaError):
        forma_amd.Context(0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "forma_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "forma_oracle" not in src, f


def test_order_and_transform_limits():
    from forma_amd import api
    assert api.Order(api.LAYER_LIMIT).as_u32() == api.LAYER_LIMIT
    with pytest.raises(api.OrderError):                      # utils/order.rs:44-66
        api.Order(api.LAYER_LIMIT + 1)
    api.GeomPresTransform.try_from([0.5, 0.0, 0.0, 0.5, 3.0, 4.0])
    with pytest.raises(api.GeomPresTransformError):          # math/transform.rs:160-222: no up-scaling
        api.GeomPresTransform.try_from([2.0, 0.0, 0.0, 2.0, 0.0, 0.0])


def test_linear_layout_checks_stride():
    from forma_amd import api
    api.LinearLayout(64, 64 * 4, 64)
    with pytest.raises(Exception):                           # cpu/buffer/layout/mod.rs:188-193
        api.LinearLayout(64, 63 * 4, 64)


def test_style_words_match_the_abi_encoding():
    """Props -> u32 words (include/forma_hip.h): header bit fields and payload sizes."""
    from forma_amd import api
    col = api.Color(0.25, 0.5, 0.75, 1.0)
    w = api._encode_props(api.Props(func=api.Func.Draw(api.Style(fill=api.Fill.Solid(col), blend_mode="Multiply"))), [])
    assert len(w) == 6 and (w[0] & 0xF) == api.BLEND_MODES.index("Multiply") and ((w[0] >> 4) & 3) == 0
    assert np.array_equal(np.asarray(w[2:6], np.uint32).view(np.float32), np.float32([0.25, 0.5, 0.75, 1.0]))
    g = api.GradientBuilder(api.Point(0, 0), api.Point(10, 0))
    g.color(api.Color(1, 0, 0, 1)); g.color(api.Color(0, 0, 1, 1))
    w = api._encode_props(api.Props(fill_rule=api.FillRule.EvenOdd, func=api.Func.Draw(api.Style(fill=api.Fill.Gradient(g.build())))), [])
    assert ((w[0] >> 4) & 3) == 1 and ((w[0] >> 6) & 1) == 1 and (w[0] >> 16) == 2 and len(w) == 6 + 5 * 2
    w = api._encode_props(api.Props(func=api.Func.Clip(3)), [])
    assert ((w[0] >> 8) & 1) == 1 and w[1] == 3


def test_host_path_builder_closes_contours():
    """PathBuilder::build appends the closing line (path.rs:596-615); runs in the C++ host library, no GPU."""
    from forma_amd import api
    p = api.PathBuilder().move_to(api.Point(0, 0)).line_to(api.Point(4, 0)).line_to(api.Point(4, 4)).build()
    from forma_amd.api import _host
    # 3 points + the closing line back to the start = 4 flattened points, all produced without a GPU
    assert _host().forma_host_path_points(p._h) == 4
