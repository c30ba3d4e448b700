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


def _c_struct_fields(hdr, name):
    """field names of `typedef struct ... { ... } name;` in declaration order (`a[N], b[N]` declares two)"""
    end = re.search(r"\}\s*" + name + r"\s*;", hdr)
    assert end, name
    body = hdr[hdr.rindex("{", 0, end.start()) + 1:end.start()]
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith("const ") else decl.split(None, 2)[2]
        fields += [re.sub(r"\[.*?\]", "", d).replace("*", "").strip() for d in names.split(",")]
    return fields


def test_rust_shim_declares_the_same_abi():
    """rust/forma_hip/ffi.rs (the `extern "C"` block a forma maintainer compiles; no toolchain here) names exactly the entry
    points of include/forma_hip.h, and its `#[repr(C)]` mirrors have the header's fields in the header's order."""
    ffi = open(os.path.join(ROOT, "rust", "forma_hip", "ffi.rs")).read()
    assert sorted(set(re.findall(r"pub fn (forma_hip_[a-z_0-9]+)", ffi))) == declared_symbols()
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name in ("forma_geom_t", "forma_rect_t", "forma_image_t", "forma_timings_t", "forma_context_info_t", "forma_sort_plan_t", "forma_kernel_time_t"):
        r = re.search(r"pub struct " + name + r"\s*\{(.*?)\n\}", ffi, flags=re.S)
        assert r, name + " missing in ffi.rs"
        assert re.findall(r"pub (\w+)\s*:", r.group(1)) == _c_struct_fields(hdr, name), name


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


def test_header_is_plain_c_and_struct_layouts_match_ctypes(tmp_path):
    """The boundary is a C ABI: include/forma_hip.h must compile as C11 (what cgo / bindgen / a Rust `extern "C"` block
    consume), and the struct sizes / field offsets the Python binding assumes must be the compiler's."""
    import shutil
    import subprocess
    from forma_amd import context
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    prog = tmp_path / "abi.c"
    prog.write_text(r'''
#include <stddef.h>
#include <stdio.h>
#include "forma_hip.h"
int main(void) {
    printf("geom %zu %zu %zu %zu\n", sizeof(forma_geom_t), offsetof(forma_geom_t, order), offsetof(forma_geom_t, flags), offsetof(forma_geom_t, xf));
    printf("rect %zu\n", sizeof(forma_rect_t));
    printf("image %zu\n", sizeof(forma_image_t));
    printf("timings %zu %zu %zu\n", sizeof(forma_timings_t), offsetof(forma_timings_t, n_lines), offsetof(forma_timings_t, n_tile_entries));
    printf("info %zu %zu\n", sizeof(forma_context_info_t), offsetof(forma_context_info_t, devices));
    printf("plan %zu %zu %zu\n", sizeof(forma_sort_plan_t), offsetof(forma_sort_plan_t, mask), offsetof(forma_sort_plan_t, bias));
    printf("ktime %zu %zu %zu\n", sizeof(forma_kernel_time_t), offsetof(forma_kernel_time_t, start_us), offsetof(forma_kernel_time_t, stage));
    return 0;
}
''')
    exe = tmp_path / "abi"
    subprocess.run([gcc, "-std=c11", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)],
                   check=True)
    out = dict(line.split(" ", 1) for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    g = context.GEOM_DTYPE
    assert out["geom"].split() == [str(g.itemsize), str(g.fields["order"][1]), str(g.fields["flags"][1]), str(g.fields["xf"][1])]
    assert int(out["image"]) == context.IMAGE_DTYPE.itemsize
    assert int(out["rect"]) == 16
    from forma_amd import _lib
    t = _lib.TimingsT
    assert out["timings"].split() == [str(C.sizeof(t)), str(t.n_lines.offset), str(t.n_tile_entries.offset)]
    assert out["info"].split() == [str(C.sizeof(_lib.ContextInfoT)), str(_lib.ContextInfoT.devices.offset)]
    assert out["plan"].split() == [str(C.sizeof(_lib.SortPlanT)), str(_lib.SortPlanT.mask.offset), str(_lib.SortPlanT.bias.offset)]
    assert out["ktime"].split() == [str(C.sizeof(_lib.KernelTimeT)), str(_lib.KernelTimeT.start_us.offset), str(_lib.KernelTimeT.stage.offset)]


def _plan(live, layer_sorted, digit_bits, rng=None):
    from forma_amd import _lib
    L = _lib.lib()
    out = _lib.SortPlanT()
    r = None if rng is None else (C.c_uint32 * 4)(*rng)
    assert L.forma_hip_sort_plan(C.c_uint64(live), int(layer_sorted), digit_bits, r, C.byref(out)) == 0
    return out


@pytest.mark.parametrize("digit_bits", [0, 4, 8, 9])
def test_sort_plans_order_like_a_stable_sort_on_the_key(digit_bits):
    """The digit plan of the segment sort is host logic (`forma_hip_sort_plan`: no device): digits packed over the LIVE key bits
    only, the layer digits dropped when the stream is already non-decreasing in layer, nine-bit digits where they save a pass,
    a tile field taken relative to its minimum where the span is known and that saves another.  The property every plan must
    have, checked here on the CPU with numpy's stable sort standing in for the counting passes: applied least significant pass
    first, the passes order any keys with those live bits (and that span) exactly like ONE stable sort on bits 20..63
    (pixel_segment.rs:161-171) — for random live masks, canvases of 2^k tiles (field value 2^k: the extra bit), both layer
    orders."""
    rng = np.random.default_rng(900 + digit_bits)
    for trial in range(60):
        tiles_w = int(rng.choice([3, 16, 64, 240, 256, 512, 1024, 4095]))
        tiles_h = int(rng.choice([2, 16, 68, 135, 256, 512, 2047]))
        n_layers = int(rng.choice([1, 2, 200, 70000, (1 << 21) - 1]))
        n = 4000
        x0 = int(rng.integers(0, tiles_w)); x1 = int(rng.integers(x0, tiles_w + 1))          # tile_x + 1 in [x0, x1]  (0 = left of the canvas)
        y0 = int(rng.integers(1, tiles_h + 1)); y1 = int(rng.integers(y0, tiles_h + 1))
        tx = rng.integers(x0, x1 + 1, n, dtype=np.uint64)
        ty = rng.integers(y0, y1 + 1, n, dtype=np.uint64)
        layer = rng.integers(0, n_layers, n, dtype=np.uint64)
        layer_sorted = bool(trial & 1)
        if layer_sorted:
            layer = np.sort(layer)                                       # non-decreasing along the stream, ties everywhere
        low = rng.integers(0, 1 << 20, n, dtype=np.uint64)               # local_x / local_y / area / cover: never sorted on
        v = (ty << np.uint64(53)) | (tx << np.uint64(41)) | (layer << np.uint64(20)) | low
        key = v >> np.uint64(20)
        live = int(np.bitwise_or.reduce(key) ^ np.bitwise_and.reduce(key))
        for use_range in (False, True):
            r = (int(tx.min()), int(tx.max()), int(ty.min()), int(ty.max())) if use_range else None
            P = _plan(live, layer_sorted, digit_bits, r)
            assert P.n_passes <= 12
            assert use_range or not P.biased
            out = v.copy()
            for p in range(P.n_passes):
                assert 20 <= P.shift[p] < 64 and P.mask[p] < (16 if digit_bits == 4 else 512)
                d = ((out >> np.uint64(P.shift[p])) - np.uint64(P.bias[p])) & np.uint64(P.mask[p])
                out = out[np.argsort(d, kind="stable")]
            want = v[np.argsort(key, kind="stable")]
            assert np.array_equal(out, want), (trial, use_range, tiles_w, tiles_h, n_layers, layer_sorted, P.n_passes, list(P.shift)[:P.n_passes])
        # what the plan is for: no digit is spent on bits that do not vary
        plain = _plan(live, layer_sorted, digit_bits)
        covered = 0
        for p in range(plain.n_passes):
            covered |= plain.mask[p] << (plain.shift[p] - 20)
        need = live & (~0x1FFFFF if layer_sorted else ~0)
        assert covered & need == need


def test_sort_plan_saves_the_pass_of_a_power_of_two_canvas():
    """8192 x 8192 pixels = 512 x 512 tiles: the fields store 1..512, ten live bits each — three 8-bit passes, or (9-bit digits,
    fields relative to their minima) two; 3840 x 2160 is two 8-bit passes either way."""
    live = ((0x3FF << 33) | (0x3FF << 21))                               # tile_y, tile_x: ten live bits each (key bit = segment bit - 20)
    assert _plan(live, True, 0).n_passes == 3
    P = _plan(live, True, 0, (1, 512, 1, 512))
    assert P.n_passes == 2 and P.biased and sorted(P.mask[:2]) == [511, 511]
    live4k = ((0xFF << 33) | (0xFF << 21))
    assert _plan(live4k, True, 0).n_passes == 2 and _plan(live4k, True, 0, (1, 240, 1, 135)).n_passes == 2
    assert _plan(live4k | 0x7FFF, False, 0).n_passes == 4                # 15 live layer bits on top: two more digits


class _FakeCtx:
    """Stands in for forma_amd.Context in host-logic tests: records table uploads, renders nothing."""
    def __init__(self):
        self.uploads = 0
        self.unchanged = None
        self._h = None

    def set_geometry(self, *a): pass
    def set_geoms(self, g): self.uploads += 1
    def set_styles(self, off, words, unchanged): self.unchanged = None if unchanged is None else np.array(unchanged)
    def set_images(self, *a): pass

    def render(self, w, h, **kw):
        return (None, {}) if kw.get("timings") else None


def test_renderer_keeps_tables_resident_until_something_changes():
    """Renderer.render re-uploads the per-frame layer / style tables only when the composition changed (or a cached frame
    changed the layers' is_unchanged bits, renderer.rs:217-223) — host logic, exercised with a recording context."""
    from forma_amd import api
    r = api.Renderer.__new__(api.Renderer)
    r._ctx = _FakeCtx(); r._caches = set(); r._geom_owner = None; r._geom_version = -1; r._slot_of = {}
    r.last_timings = {}; r.host_tables = {}; r._tables_key = None; r._marked_key = None
    r._upload_geometry = lambda comp: (setattr(r, "_geom_owner", comp._shared), setattr(r, "_geom_version", comp._shared.geometry_version),
                                       setattr(r, "_slot_of", {l.geom_id(): i for i, l in enumerate(comp.layers.values())}))
    comp = api.Composition()
    tri = api.PathBuilder().move_to(api.Point(1, 1)).line_to(api.Point(9, 1)).line_to(api.Point(9, 9)).build()
    for o in range(3):
        comp.get_mut_or_insert_default(api.Order(o)).insert(tri)
    buf = lambda cache=None: (api.BufferBuilder(np.zeros(64 * 64 * 4, np.uint8), api.LinearLayout(64, 256, 64)).layer_cache(cache).build()
                              if cache else api.BufferBuilder(np.zeros(64 * 64 * 4, np.uint8), api.LinearLayout(64, 256, 64)).build())
    r.render(comp, buf()); r.render(comp, buf()); r.render(comp, buf())
    assert r._ctx.uploads == 1                                    # static scene: one upload
    comp.get_mut(api.Order(1)).set_props(api.Props(fill_rule=api.FillRule.EvenOdd))
    r.render(comp, buf()); r.render(comp, buf())
    assert r._ctx.uploads == 2                                    # a style changed: one more
    comp.get_mut(api.Order(1)).set_props(api.Props(fill_rule=api.FillRule.EvenOdd))   # same value: not a change
    r.render(comp, buf())
    assert r._ctx.uploads == 2
    cache = api.BufferLayerCache(0, r)
    r.render(comp, buf(cache))                                    # first cached frame: nothing is "unchanged" yet
    assert r._ctx.uploads == 3 and not r._ctx.unchanged.any()
    r.render(comp, buf(cache))                                    # second: every layer is
    assert r._ctx.uploads == 4 and r._ctx.unchanged.all()
    r.render(comp, buf(cache)); r.render(comp, buf(cache))
    assert r._ctx.uploads == 4                                    # ... and it stays that way
    comp.get_mut(api.Order(2)).set_transform(api.GeomPresTransform.try_from([1, 0, 0, 1, 2, 0]))
    r.render(comp, buf(cache))
    assert r._ctx.uploads == 5 and list(r._ctx.unchanged) == [1, 1, 0]
    comp.get_mut(api.Order(0)).disable()
    r.render(comp, buf(cache))
    assert r._ctx.uploads == 6
    other = api.Composition()
    other.get_mut_or_insert_default(api.Order(0)).insert(tri)
    r.render(other, buf()); r.render(comp, buf(cache))
    assert r._ctx.uploads == 8                                    # another composition in between: both re-upload


def test_dropped_layers_release_their_geometry():
    """`impl Drop for Layer` (composition/layer.rs:356-363) removes the layer's geometry id from geom_id_to_order, so
    compact_geom (composition/mod.rs:372-384) can collect the lines of layers an application removed or replaced —
    without it the store of an app that creates and removes layers every frame grows without bound."""
    import gc
    from forma_amd import api
    comp = api.Composition()
    tri = api.PathBuilder().move_to(api.Point(1, 1)).line_to(api.Point(9, 1)).line_to(api.Point(9, 9)).build()
    comp.get_mut_or_insert_default(api.Order(0)).insert(tri)
    for frame in range(6):                                       # a fresh layer at order 1 every frame, the old one dropped
        comp.insert(api.Order(1), comp.create_layer().insert(tri))
        gc.collect()
        assert len(comp._shared.geom_id_to_order) == 2, frame     # order 0's id + the live order-1 layer's id
        comp.compact_geom()
        assert comp.builder_len() <= 2 * max(comp.actual_len(), 1) + 3
    removed = comp.remove(api.Order(1))
    assert removed is not None and comp._shared.geom_id_to_order[removed.geom_id()] is None   # held by the caller: still known
    gid = removed.geom_id()
    del removed
    gc.collect()
    assert gid not in comp._shared.geom_id_to_order
    comp.compact_geom()
    assert comp.builder_len() == comp.actual_len() == 3           # only order 0's triangle (3 lines) is left
    lay = comp.get_mut(api.Order(0))
    old = lay.geom_id()
    lay.clear()                                                   # Layer::clear takes a new id; the drop must release the NEW one
    new = lay.geom_id()
    assert old not in comp._shared.geom_id_to_order and new in comp._shared.geom_id_to_order
    comp.remove(api.Order(0)); del lay
    gc.collect()
    assert not comp._shared.geom_id_to_order


def test_every_debug_switch_is_documented():
    """forma_amd/csrc/debug.h is the one place the library reads the environment: every token its parser accepts is described
    in the header's own comment and listed in tools/README.md (the switches the suite is run under)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "forma_amd", "csrc", "debug.h")).read()
    head = src[:src.index("#pragma once")]
    toks = set(re.findall(r"FD_FLAG\((\w+)\)", src.split("#define FD_FLAG", 1)[1])) | set(re.findall(r'strcmp\(tok, "(\w+)"\)', src))
    toks.discard("name")
    assert len(toks) >= 25
    readme = open(os.path.join(root, "tools", "README.md")).read()
    for t in sorted(toks):
        assert t in head, f"{t}: not described in debug.h's header comment"
        assert t in readme, f"{t}: not listed in tools/README.md"
