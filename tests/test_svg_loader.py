"""SVG-subset loader (reference demo/src/demos/svg.rs): host logic only, no GPU.  The PathBuilder is
replaced by a recorder so the emitted forma path commands can be compared with hand-derived values."""
import math

import numpy as np
import pytest

from forma_amd import api, svg


class Recorder:
    def __init__(self):
        self.cmds = []

    def move_to(self, p): self.cmds.append(("M", p.x, p.y)); return self
    def line_to(self, p): self.cmds.append(("L", p.x, p.y)); return self
    def quad_to(self, a, b): self.cmds.append(("Q", a.x, a.y, b.x, b.y)); return self
    def cubic_to(self, a, b, c): self.cmds.append(("C", a.x, a.y, b.x, b.y, c.x, c.y)); return self
    def rat_quad_to(self, a, b, w): self.cmds.append(("R", a.x, a.y, b.x, b.y, w)); return self

    def build(self):
        return self

    def transform(self, t9):
        self.scale = list(t9)
        return self


@pytest.fixture()
def rec(monkeypatch):
    monkeypatch.setattr(svg.api, "PathBuilder", Recorder)


def doc(body):
    return f'<svg xmlns="http://www.w3.org/2000/svg">{body}</svg>'


def test_path_data_grammar():
    segs = list(svg.path_segments("M10-20.5.5e1 3l1,1 2 2h3V4zm1 1a1 1 0 01 10 10"))
    assert segs == [("M", True, (10.0, -20.5)), ("L", True, (5.0, 3.0)), ("L", False, (1.0, 1.0)), ("L", False, (2.0, 2.0)),
                    ("H", False, (3.0,)), ("V", True, (4.0,)), ("Z", False, ()), ("M", False, (1.0, 1.0)),
                    ("A", False, (1.0, 1.0, 0.0, 0.0, 1.0, 10.0, 10.0))]
    assert list(svg.path_segments("L 1 1")) == []                       # must start with a moveto
    assert list(svg.path_segments("M 0 0 L 1")) == [("M", True, (0.0, 0.0))]   # stops at the first error


def test_colors_and_to_linear():
    assert svg.parse_color("#fff") == (255, 255, 255)
    assert svg.parse_color("#1a2B3c") == (0x1A, 0x2B, 0x3C)
    assert svg.parse_color("rgb(10, 20,30)") == (10, 20, 30)
    assert svg.parse_color("rgb(100%,0%,50%)") == (255, 0, 127)
    assert svg.parse_color("CornflowerBlue") == (0x64, 0x95, 0xED)
    assert svg.parse_color("none") is None and svg.parse_color("url(#a)") is None
    c = svg.to_linear((255, 0, 10))
    assert c.r == pytest.approx(1.0, abs=1e-6) and c.g == 0.0
    assert c.b == pytest.approx((10 / 255) / 12.92, rel=1e-6)
    assert svg.to_linear((128, 128, 128)).r == pytest.approx(((128 / 255 + 0.055) / 1.055) ** 2.4, rel=1e-5)


def test_transform_list():
    t = svg.parse_transform("translate(10 20) scale(2)")
    assert t.apply(1.0, 1.0) == (12.0, 22.0)                             # scale first, then translate
    t = svg.parse_transform("matrix(1 2 3 4 5 6)")
    assert t.apply(1.0, 1.0) == (1 + 3 + 5, 2 + 4 + 6)
    x, y = svg.parse_transform("rotate(90)").apply(1.0, 0.0)
    assert x == pytest.approx(0.0, abs=1e-12) and y == pytest.approx(1.0)
    x, y = svg.parse_transform("rotate(90, 1, 1)").apply(2.0, 1.0)
    assert (x, y) == pytest.approx((1.0, 2.0))
    assert svg.parse_transform("bogus(1)") is None


def test_path_commands_relative_smooth_and_close(rec):
    s = svg.Svg(doc('<path d="M10 10 l10 0 v10 h-10 z l5 5 Q 20 20 30 10 T 50 10 C 1 2 3 4 5 6 s 1 1 2 2 S 9 9 10 10" fill="#ff0000"/>'),
                2.0, is_text=True)
    (p, rule, fill, blend), = s.paths
    assert p.scale == [2.0, 0, 0, 0, 2.0, 0, 0, 0, 1.0]
    assert rule == api.FillRule.NonZero and blend == "Over"
    assert fill == api.Fill.Solid(api.Color(1.0, 0.0, 0.0, 1.0))
    assert p.cmds == [
        ("M", 10, 10), ("L", 20, 10), ("L", 20, 20), ("L", 10, 20),
        ("L", 15, 15),                                  # z moved the pen back to (10,10); PathBuilder closes by itself
        ("Q", 20, 20, 30, 10), ("Q", 40, 0, 50, 10),    # T reflects (20,20) about (30,10)
        ("C", 1, 2, 3, 4, 5, 6), ("C", 7, 8, 6, 7, 7, 8),   # s reflects (3,4) about (5,6)
        ("C", 7, 8, 9, 9, 10, 10),                      # S after s reflects the stored reflection (7,8) about (7,8)
    ]


def test_stroked_and_dataless_paths_are_skipped(rec):
    s = svg.Svg(doc('<path d="M0 0L1 1" stroke="#000"/><path fill="red"/><path d="M0 0L1 1" stroke="none"/>'), is_text=True)
    assert len(s.paths) == 1


def test_groups_innermost_transform_fill_and_opacity_product(rec):
    s = svg.Svg(doc('<g transform="translate(100 0)" fill="blue" opacity="0.5"><g transform="scale(2)" opacity="0.5">'
                    '<path d="M1 1L2 2"/><rect x="1" y="2" width="3" height="4" fill="#00ff00" fill-opacity="0.25"/></g>'
                    '<path d="M1 1L2 2"/></g><path d="M1 1 L2 2"/>'), is_text=True)
    inner, rect, outer, bare = s.paths
    assert inner[0].cmds == [("M", 2, 2), ("L", 4, 4)]                  # only the innermost transform, not composed
    assert inner[2] == api.Fill.Solid(api.Color(0.0, 0.0, 1.0, 0.25))   # group fill, opacity product
    assert rect[0].cmds == [("M", 1, 2), ("L", 1, 6), ("L", 4, 6), ("L", 4, 2), ("L", 1, 2)]   # rect ignores the transform
    assert rect[2] == api.Fill.Solid(api.Color(0.0, 1.0, 0.0, 0.25))
    assert outer[0].cmds == [("M", 101, 1), ("L", 102, 2)] and outer[2][1].a == 0.5
    assert bare[0].cmds == [("M", 1, 1), ("L", 2, 2)]
    assert bare[2] == api.Fill.Solid(api.Color(0.0, 0.0, 0.0, 1.0))     # no colour anywhere: opaque black


def test_gradients_blend_modes_and_fill_rule(rec):
    s = svg.Svg(doc('<linearGradient id="a" gradientUnits="userSpaceOnUse" x1="0" y1="0" x2="10" y2="0">'
                    '<stop offset="0%" stop-color="#ff0000"/><stop offset="100%" stop-color="#0000ff" stop-opacity="0.5"/></linearGradient>'
                    '<radialGradient id="b" gradientUnits="userSpaceOnUse" cx="5" cy="6" r="7">'
                    '<stop offset="25%" stop-color="white"/><stop offset="75%"/></radialGradient>'
                    '<path d="M0 0L9 0L9 9" fill="url(#a)" fill-rule="evenodd" style="opacity:1; mix-blend-mode: color-dodge"/>'
                    '<path d="M0 0L9 0L9 9" fill="url(#b)" style="mix-blend-mode:bogus"/>'
                    '<path d="M0 0L9 0L9 9" fill="url(#missing)"/>'), is_text=True)
    a, b, c = s.paths
    assert a[1] == api.FillRule.EvenOdd and a[3] == "ColorDodge" and b[3] == "Over"
    kind, g = a[2]
    assert kind == "gradient" and g.type == api.GradientType.Linear and (g.start, g.end) == (api.Point(0, 0), api.Point(10, 0))
    assert [st for _, st in g.stops] == [0.0, 1.0] and g.stops[1][0] == api.Color(0.0, 0.0, 1.0, 0.5)
    kind, g = b[2]
    assert g.type == api.GradientType.Radial and (g.start, g.end) == (api.Point(5, 6), api.Point(12, 6))
    assert [st for _, st in g.stops] == [0.25, 0.75] and g.stops[1][0] == api.Color(0.0, 0.0, 0.0, 1.0)
    assert c[2] == api.Fill.Solid(api.Color(0.0, 0.0, 0.0, 1.0))        # unknown url(): colour does not parse -> black


def test_arc_becomes_quarter_turn_rational_quads(rec):
    # half circle of radius 10 around (10, 0), from (0,0) to (20,0), sweep flag 1 (through y < 0 in SVG's y-down frame)
    s = svg.Svg(doc('<path d="M0 0 A10 10 0 0 1 20 0"/>'), is_text=True)
    cmds = s.paths[0][0].cmds
    assert [c[0] for c in cmds] == ["M", "R", "R"]
    w = math.cos(math.pi / 4)
    (_, c1x, c1y, e1x, e1y, w1), (_, c2x, c2y, e2x, e2y, w2) = cmds[1:]
    assert w1 == pytest.approx(w, rel=1e-6) and w2 == pytest.approx(w, rel=1e-6)
    assert (e1x, e1y) == pytest.approx((10.0, -10.0), abs=1e-4) and (c1x, c1y) == pytest.approx((0.0, -10.0), abs=1e-4)
    assert (e2x, e2y) == pytest.approx((20.0, 0.0), abs=1e-4) and (c2x, c2y) == pytest.approx((20.0, -10.0), abs=1e-4)
    # degenerate arcs (zero radius, coincident end points) emit nothing and leave the pen where it was
    s = svg.Svg(doc('<path d="M0 0 A0 10 0 0 1 20 0 L 5 5 a3 3 0 0 0 0 0 l1 1"/>'), is_text=True)
    assert s.paths[0][0].cmds == [("M", 0, 0), ("L", 5, 5), ("L", 6, 6)]


def test_compose_inserts_layer_per_path_in_order():
    s = svg.Svg(doc('<path d="M0 0L8 0L8 8" fill="red"/><rect width="4" height="4" fill="#00f" style="mix-blend-mode:multiply"/>'),
                is_text=True)
    comp = s.compose(api.Composition())
    assert len(comp) == 2
    assert comp.get(api.Order(0)).props().func[1].fill == api.Fill.Solid(api.Color(1.0, 0.0, 0.0, 1.0))
    assert comp.get(api.Order(1)).props().func[1].blend_mode == "Multiply"
    assert comp.get(api.Order(1)).transform().is_identity()
