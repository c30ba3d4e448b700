"""SVG-subset loader: the caller that feeds BASELINE config 3 (`paris-30k.svg`) into the hot path.

Behavioural mirror of the reference demo's loader (demo/src/demos/svg.rs:27-924, colours demo/src/main.rs:134-151),
written for this host layer: one streaming pass over the XML (expat), a table-driven path-data interpreter, and forma
paths / props built through `forma_amd.api`.  What the reference does, and so what this does:

  * `<g>`: a stack of {transform, fill, opacity}.  A point is mapped by the *innermost group that has a transform*
    (transforms of outer groups are NOT composed, svg.rs:226-231); fill = innermost group fill; opacity = product.
  * `<path>`: skipped when it has a `stroke` other than "none" or no `d`.  M/L/H/V/Q/T/C/S/A/Z, absolute and relative;
    arcs become <= 90 degree rational quadratics with weight cos(sweep/2) (svg.rs:276-335).  `Z` only moves the pen
    back to the sub-path start (svg.rs:731-737); forma's PathBuilder closes contours itself.
  * `<rect>`: x, y, width, height — not mapped through the group transform (svg.rs:769-773).
  * `<linearGradient>` / `<radialGradient>` with gradientUnits="userSpaceOnUse" and `<stop offset="NN%">`;
    gradient coordinates are used as written (neither group transform nor `scale`).
  * fill: `url(#id)` of a known gradient, else the colour (`fill`, else `stop-color`, else the group fill), sRGB ->
    linear; alpha = `opacity` / `stop-opacity` / `fill-opacity`, else the product of the group opacities; a path with
    no parsable colour anywhere is opaque black (svg.rs:255-273).
  * `style="mix-blend-mode: ..."` -> BlendMode; `fill-rule="evenodd"` -> FillRule.EvenOdd.
  * finally every path is scaled by `scale` (Path::transform, svg.rs:217-220); layer i = path i, Order(i).

Two reference quirks are kept because they decide pixels: the arc's x-axis-rotation is passed to cos/sin as written
(svg.rs:62-63 — degrees treated as radians), and the endpoint's rotated y uses the already rotated x (svg.rs:65-69).
"""
from __future__ import annotations

import math
import re
import xml.parsers.expat
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import api

f32 = np.float32
_PI = f32(math.pi)

# ---------------------------------------------------------------------------------------------------------------------
# colours (svgtypes::Color: #rgb, #rrggbb, rgb(r,g,b), rgb(r%,g%,b%), CSS keywords)
_NAMED = {
    "aliceblue": 0xF0F8FF, "antiquewhite": 0xFAEBD7, "aqua": 0x00FFFF, "aquamarine": 0x7FFFD4, "azure": 0xF0FFFF,
    "beige": 0xF5F5DC, "bisque": 0xFFE4C4, "black": 0x000000, "blanchedalmond": 0xFFEBCD, "blue": 0x0000FF,
    "blueviolet": 0x8A2BE2, "brown": 0xA52A2A, "burlywood": 0xDEB887, "cadetblue": 0x5F9EA0, "chartreuse": 0x7FFF00,
    "chocolate": 0xD2691E, "coral": 0xFF7F50, "cornflowerblue": 0x6495ED, "cornsilk": 0xFFF8DC, "crimson": 0xDC143C,
    "cyan": 0x00FFFF, "darkblue": 0x00008B, "darkcyan": 0x008B8B, "darkgoldenrod": 0xB8860B, "darkgray": 0xA9A9A9,
    "darkgreen": 0x006400, "darkgrey": 0xA9A9A9, "darkkhaki": 0xBDB76B, "darkmagenta": 0x8B008B,
    "darkolivegreen": 0x556B2F, "darkorange": 0xFF8C00, "darkorchid": 0x9932CC, "darkred": 0x8B0000,
    "darksalmon": 0xE9967A, "darkseagreen": 0x8FBC8F, "darkslateblue": 0x483D8B, "darkslategray": 0x2F4F4F,
    "darkslategrey": 0x2F4F4F, "darkturquoise": 0x00CED1, "darkviolet": 0x9400D3, "deeppink": 0xFF1493,
    "deepskyblue": 0x00BFFF, "dimgray": 0x696969, "dimgrey": 0x696969, "dodgerblue": 0x1E90FF, "firebrick": 0xB22222,
    "floralwhite": 0xFFFAF0, "forestgreen": 0x228B22, "fuchsia": 0xFF00FF, "gainsboro": 0xDCDCDC, "ghostwhite": 0xF8F8FF,
    "gold": 0xFFD700, "goldenrod": 0xDAA520, "gray": 0x808080, "grey": 0x808080, "green": 0x008000,
    "greenyellow": 0xADFF2F, "honeydew": 0xF0FFF0, "hotpink": 0xFF69B4, "indianred": 0xCD5C5C, "indigo": 0x4B0082,
    "ivory": 0xFFFFF0, "khaki": 0xF0E68C, "lavender": 0xE6E6FA, "lavenderblush": 0xFFF0F5, "lawngreen": 0x7CFC00,
    "lemonchiffon": 0xFFFACD, "lightblue": 0xADD8E6, "lightcoral": 0xF08080, "lightcyan": 0xE0FFFF,
    "lightgoldenrodyellow": 0xFAFAD2, "lightgray": 0xD3D3D3, "lightgreen": 0x90EE90, "lightgrey": 0xD3D3D3,
    "lightpink": 0xFFB6C1, "lightsalmon": 0xFFA07A, "lightseagreen": 0x20B2AA, "lightskyblue": 0x87CEFA,
    "lightslategray": 0x778899, "lightslategrey": 0x778899, "lightsteelblue": 0xB0C4DE, "lightyellow": 0xFFFFE0,
    "lime": 0x00FF00, "limegreen": 0x32CD32, "linen": 0xFAF0E6, "magenta": 0xFF00FF, "maroon": 0x800000,
    "mediumaquamarine": 0x66CDAA, "mediumblue": 0x0000CD, "mediumorchid": 0xBA55D3, "mediumpurple": 0x9370DB,
    "mediumseagreen": 0x3CB371, "mediumslateblue": 0x7B68EE, "mediumspringgreen": 0x00FA9A,
    "mediumturquoise": 0x48D1CC, "mediumvioletred": 0xC71585, "midnightblue": 0x191970, "mintcream": 0xF5FFFA,
    "mistyrose": 0xFFE4E1, "moccasin": 0xFFE4B5, "navajowhite": 0xFFDEAD, "navy": 0x000080, "oldlace": 0xFDF5E6,
    "olive": 0x808000, "olivedrab": 0x6B8E23, "orange": 0xFFA500, "orangered": 0xFF4500, "orchid": 0xDA70D6,
    "palegoldenrod": 0xEEE8AA, "palegreen": 0x98FB98, "paleturquoise": 0xAFEEEE, "palevioletred": 0xDB7093,
    "papayawhip": 0xFFEFD5, "peachpuff": 0xFFDAB9, "peru": 0xCD853F, "pink": 0xFFC0CB, "plum": 0xDDA0DD,
    "powderblue": 0xB0E0E6, "purple": 0x800080, "red": 0xFF0000, "rosybrown": 0xBC8F8F, "royalblue": 0x4169E1,
    "saddlebrown": 0x8B4513, "salmon": 0xFA8072, "sandybrown": 0xF4A460, "seagreen": 0x2E8B57, "seashell": 0xFFF5EE,
    "sienna": 0xA0522D, "silver": 0xC0C0C0, "skyblue": 0x87CEEB, "slateblue": 0x6A5ACD, "slategray": 0x708090,
    "slategrey": 0x708090, "snow": 0xFFFAFA, "springgreen": 0x00FF7F, "steelblue": 0x4682B4, "tan": 0xD2B48C,
    "teal": 0x008080, "thistle": 0xD8BFD8, "tomato": 0xFF6347, "turquoise": 0x40E0D0, "violet": 0xEE82EE,
    "wheat": 0xF5DEB3, "white": 0xFFFFFF, "whitesmoke": 0xF5F5F5, "yellow": 0xFFFF00, "yellowgreen": 0x9ACD32,
}


def parse_color(text: str) -> Optional[Tuple[int, int, int]]:
    """sRGB bytes of an SVG colour, or None when it does not parse ("none", "url(...)", "currentColor", ...)."""
    s = text.strip()
    if s.startswith("#"):
        h = s[1:]
        if not re.fullmatch(r"[0-9a-fA-F]+", h):
            return None
        if len(h) == 3:
            return tuple(int(c * 2, 16) for c in h)
        if len(h) == 6:
            return int(h[0:2], 16), int(h[2:4], 16), int(h[4:6], 16)
        return None
    m = re.fullmatch(r"rgb\(\s*([^,\s]+)\s*,\s*([^,\s]+)\s*,\s*([^,\s)]+)\s*\)", s, re.I)
    if m:
        out = []
        for part in m.groups():
            try:
                if part.endswith("%"):
                    v = float(part[:-1]) / 100.0 * 255.0
                else:
                    v = float(part)
            except ValueError:
                return None
            out.append(int(min(max(v, 0.0), 255.0)))
        return tuple(out)
    v = _NAMED.get(s.lower())
    return None if v is None else ((v >> 16) & 255, (v >> 8) & 255, v & 255)


def to_linear(rgb: Sequence[int]) -> api.Color:
    """demo/src/main.rs:134-151: sRGB byte -> linear f32 (x/255; <= 0.04045 ? /12.92 : ((x+0.055)/1.055)^2.4)."""
    def conv(b: int) -> float:
        l = f32(b) * (f32(1.0) / f32(255.0))
        if l <= f32(0.04045):
            return float(l * (f32(1.0) / f32(12.92)))
        return float(np.power((l + f32(0.055)) * (f32(1.0) / f32(1.055)), f32(2.4), dtype=f32))
    return api.Color(conv(rgb[0]), conv(rgb[1]), conv(rgb[2]), 1.0)


# ---------------------------------------------------------------------------------------------------------------------
# transform="..." (svgtypes::Transform: a 2x3 matrix a b c d e f; list items multiply left to right)
@dataclass(frozen=True)
class Transform:
    a: float = 1.0
    b: float = 0.0
    c: float = 0.0
    d: float = 1.0
    e: float = 0.0
    f: float = 0.0

    def then(self, o: "Transform") -> "Transform":            # self * o  (o is applied to the point first)
        return Transform(self.a * o.a + self.c * o.b, self.b * o.a + self.d * o.b,
                         self.a * o.c + self.c * o.d, self.b * o.c + self.d * o.d,
                         self.a * o.e + self.c * o.f + self.e, self.b * o.e + self.d * o.f + self.f)

    def apply(self, x: float, y: float) -> Tuple[float, float]:      # f64, like Transform::apply_to
        return x * self.a + y * self.c + self.e, x * self.b + y * self.d + self.f


_NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"
_TF_ITEM = re.compile(r"\s*([A-Za-z]+)\s*\(([^)]*)\)\s*,?")


def parse_transform(text: str) -> Optional[Transform]:
    out = Transform()
    pos, text = 0, text.strip()
    while pos < len(text):
        m = _TF_ITEM.match(text, pos)
        if not m:
            return None
        pos = m.end()
        name = m.group(1)
        try:
            v = [float(t) for t in re.findall(_NUM, m.group(2))]
        except ValueError:
            return None
        if name == "matrix" and len(v) == 6:
            t = Transform(*v)
        elif name == "translate" and len(v) in (1, 2):
            t = Transform(e=v[0], f=v[1] if len(v) == 2 else 0.0)
        elif name == "scale" and len(v) in (1, 2):
            t = Transform(a=v[0], d=v[1] if len(v) == 2 else v[0])
        elif name == "rotate" and len(v) in (1, 3):
            r = math.radians(v[0])
            t = Transform(math.cos(r), math.sin(r), -math.sin(r), math.cos(r))
            if len(v) == 3:
                t = Transform(e=v[1], f=v[2]).then(t).then(Transform(e=-v[1], f=-v[2]))
        elif name == "skewX" and len(v) == 1:
            t = Transform(c=math.tan(math.radians(v[0])))
        elif name == "skewY" and len(v) == 1:
            t = Transform(b=math.tan(math.radians(v[0])))
        else:
            return None
        out = out.then(t)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# path data (svgtypes::PathParser): commands with implicit repetition, compact numbers, single-character arc flags
_ARGS = {"M": 2, "L": 2, "H": 1, "V": 1, "C": 6, "S": 4, "Q": 4, "T": 2, "A": 7, "Z": 0}
_NUM_RE = re.compile(_NUM)
_WSP = " \t\r\n,"


def path_segments(d: str):
    """Yield (command letter upper-case, is_absolute, args tuple of python floats).  Stops at the first syntax error, as
    an iterator over PathParser results would before the reference's `unwrap` — here the remainder is dropped."""
    i, n = 0, len(d)
    cmd = None
    first = True
    while True:
        while i < n and d[i] in _WSP:
            i += 1
        if i >= n:
            return
        ch = d[i]
        if ch.isalpha():
            if ch.upper() not in _ARGS:
                return
            cmd = ch
            i += 1
            if first and cmd.upper() != "M":
                return                                         # path data must start with a moveto
            if cmd.upper() == "Z":
                yield "Z", cmd == "Z", ()
                first = False
                continue
        elif cmd is None or cmd.upper() == "Z":
            return
        first = False
        up = cmd.upper()
        args = []
        for k in range(_ARGS[up]):
            while i < n and d[i] in _WSP:
                i += 1
            if up == "A" and k in (3, 4):                      # flags are single characters: "a1 1 0 01 10 10"
                if i < n and d[i] in "01":
                    args.append(float(d[i])); i += 1
                    continue
                return
            m = _NUM_RE.match(d, i)
            if not m:
                return
            args.append(float(m.group(0))); i = m.end()
        yield up, cmd == up, tuple(args)
        if up == "M":                                          # further pairs after a moveto are linetos
            cmd = "L" if cmd == "M" else "l"


@dataclass
class _Arc:
    cx: np.float32
    cy: np.float32
    rx: np.float32
    ry: np.float32
    phi: np.float32
    angle: np.float32
    delta: np.float32


def _arc_center(rx, ry, phi, large_arc: bool, sweep: bool, x0, y0, x1, y1) -> Optional[_Arc]:
    """Endpoint -> centre parametrisation in f32 (svg.rs:41-116)."""
    rx, ry, phi, x0, y0, x1, y1 = (f32(v) for v in (rx, ry, phi, x0, y0, x1, y1))
    eps = np.finfo(np.float32).eps
    if abs(x0 - x1) < eps and abs(y0 - y1) < eps:
        return None
    rx, ry = abs(rx), abs(ry)
    if rx == 0 or ry == 0:
        return None
    c, s = np.cos(phi), np.sin(phi)
    x0 = (x0 * c + y0 * s) / rx
    y0 = (-x0 * s + y0 * c) / ry                              # uses the new x0 (reference shadowing, svg.rs:65-66)
    x1 = (x1 * c + y1 * s) / rx
    y1 = (-x1 * s + y1 * c) / ry
    lx, ly = (x0 - x1) * f32(0.5), (y0 - y1) * f32(0.5)
    cx, cy = (x0 + x1) * f32(0.5), (y0 + y1) * f32(0.5)
    len2 = lx * lx + ly * ly
    if len2 < f32(1.0):
        rad = np.sqrt((f32(1.0) - len2) / len2)
        if large_arc != sweep:
            rad = -rad
        cx = cx + (-ly * rad)
        cy = cy + lx * rad
    theta = np.arctan2(y0 - cy, x0 - cx)
    delta = np.arctan2(y1 - cy, x1 - cx) - theta
    cxs, cys = cx * rx, cy * ry
    cx, cy = cxs * c - cys * s, cxs * s + cys * c
    two_pi = _PI * f32(2.0)
    if sweep:
        if delta < 0:
            delta = delta + two_pi
    elif delta > 0:
        delta = delta - two_pi
    return _Arc(f32(cx), f32(cy), rx, ry, phi, f32(theta), f32(delta))


@dataclass
class _Group:
    transform: Optional[Transform]
    fill: Optional[Tuple[int, int, int]]
    opacity: Optional[float]


_BLEND = {"normal": "Over", "multiply": "Multiply", "screen": "Screen", "overlay": "Overlay", "darken": "Darken",
          "lighten": "Lighten", "color-dodge": "ColorDodge", "color-burn": "ColorBurn", "hard-light": "HardLight",
          "soft-light": "SoftLight", "difference": "Difference", "exclusion": "Exclusion", "hue": "Hue",
          "saturation": "Saturation", "color": "Color", "luminosity": "Luminosity"}


def _float(text: Optional[str]) -> Optional[float]:
    if text is None:
        return None
    try:
        return float(f32(float(text.strip())))
    except ValueError:
        return None


def _attr_color(attrs: Dict[str, str]) -> Optional[Tuple[int, int, int]]:
    v = attrs.get("fill", attrs.get("stop-color"))
    return None if v is None else parse_color(v)


def _attr_opacity(attrs: Dict[str, str]) -> Optional[float]:
    v = attrs.get("opacity", attrs.get("stop-opacity"))
    o = _float(v)
    return o if o is not None else _float(attrs.get("fill-opacity"))


def _blend_mode(attrs: Dict[str, str]) -> str:
    for decl in attrs.get("style", "").split(";"):
        k, _, v = decl.partition(":")
        if k.strip() == "mix-blend-mode":
            return _BLEND.get(v.strip(), "Over")
    return "Over"


class Svg:
    """`Svg(path_or_text, scale)` parses; `.paths` = [(api.Path, fill_rule, fill, blend_mode)]; `.compose(composition)`
    inserts layer i at Order(i) like the reference's `App::compose` (svg.rs:895-922)."""

    def __init__(self, source: str, scale: float = 1.0, *, is_text: bool = False):
        self.groups: List[_Group] = []
        self.paths: List[Tuple[api.Path, str, tuple, str]] = []
        self.gradients: Dict[str, api.Gradient] = {}
        self._gradient: Optional[Tuple[str, api.GradientBuilder]] = None
        self.x = 0.0
        self.y = 0.0
        data = source.encode() if is_text else open(source, "rb").read()
        p = xml.parsers.expat.ParserCreate()
        p.StartElementHandler = self._start
        p.EndElementHandler = self._end
        p.Parse(data, True)
        s = float(scale)
        t9 = [s, 0.0, 0.0, 0.0, s, 0.0, 0.0, 0.0, 1.0]
        self.paths = [(path.transform(t9), fr, fill, bm) for path, fr, fill, bm in self.paths]

    # -- group state -------------------------------------------------------------------------------------------------
    def _t(self, x, y) -> api.Point:
        for g in reversed(self.groups):
            if g.transform is not None:
                tx, ty = g.transform.apply(float(x), float(y))
                return api.Point(float(f32(tx)), float(f32(ty)))
        return api.Point(float(x), float(y))

    def _group_fill(self):
        for g in reversed(self.groups):
            if g.fill is not None:
                return g.fill
        return None

    def _groups_opacity(self) -> float:
        o = f32(1.0)
        for g in self.groups:
            if g.opacity is not None:
                o = o * f32(g.opacity)
        return float(o)

    def _fill(self, attrs) -> tuple:
        ref = attrs.get("fill", "")
        if ref.startswith("url(#") and ref.endswith(")") and ref[5:-1] in self.gradients:
            return api.Fill.Gradient(self.gradients[ref[5:-1]])
        rgb = _attr_color(attrs) or self._group_fill()
        if rgb is None:
            return api.Fill.Solid(api.Color(0.0, 0.0, 0.0, 1.0))
        op = _attr_opacity(attrs)
        if op is None:
            op = self._groups_opacity()
        c = to_linear(rgb)
        return api.Fill.Solid(api.Color(c.r, c.g, c.b, op))

    @staticmethod
    def _stroked(attrs) -> bool:
        return attrs.get("stroke", "none") != "none"

    # -- XML events --------------------------------------------------------------------------------------------------
    def _start(self, tag: str, attrs: Dict[str, str]):
        tag = tag.rsplit(":", 1)[-1]
        if tag == "g":
            tf = attrs.get("transform")
            self.groups.append(_Group(parse_transform(tf) if tf is not None else None, _attr_color(attrs), _attr_opacity(attrs)))
        elif tag == "path":
            self._path(attrs)
        elif tag == "rect":
            self._rect(attrs)
        elif tag in ("linearGradient", "radialGradient"):
            if attrs.get("gradientUnits") != "userSpaceOnUse":
                return
            gid = attrs["id"]
            if tag == "linearGradient":
                v = [_float(attrs.get(k)) for k in ("x1", "y1", "x2", "y2")]
                if any(c is None for c in v):
                    raise ValueError("linearGradient missing x1/y1/x2/y2")
                gb = api.GradientBuilder(api.Point(v[0], v[1]), api.Point(v[2], v[3])).type(api.GradientType.Linear)
            else:
                v = [_float(attrs.get(k)) for k in ("cx", "cy", "r")]
                if any(c is None for c in v):
                    raise ValueError("radialGradient missing cx/cy/r")
                gb = api.GradientBuilder(api.Point(v[0], v[1]), api.Point(float(f32(v[0]) + f32(v[2])), v[1])).type(api.GradientType.Radial)
            self._gradient = (gid, gb)
        elif tag == "stop":
            if self._gradient is None:
                raise ValueError("stop missing gradient start tag")
            rgb = _attr_color(attrs) or (0, 0, 0)
            op = _attr_opacity(attrs)
            c = to_linear(rgb)
            off = attrs.get("offset")
            stop = _float(off[:-1]) if off else None           # "NN%": the last character is dropped unseen (svg.rs:863-866)
            if stop is None:
                raise ValueError("stop missing offset")
            self._gradient[1].color_with_stop(api.Color(c.r, c.g, c.b, 1.0 if op is None else op), float(f32(stop) / f32(100.0)))

    def _end(self, tag: str):
        tag = tag.rsplit(":", 1)[-1]
        if tag == "g":
            if self.groups:
                self.groups.pop()
        elif tag in ("linearGradient", "radialGradient") and self._gradient is not None:
            gid, gb = self._gradient
            self._gradient = None
            g = gb.build()
            if g is None:
                raise ValueError(f"{tag} requires at least 2 stops")
            self.gradients[gid] = g

    # -- shapes ------------------------------------------------------------------------------------------------------
    def _push(self, builder: api.PathBuilder, attrs):
        rule = api.FillRule.EvenOdd if attrs.get("fill-rule") == "evenodd" else api.FillRule.NonZero
        self.paths.append((builder.build(), rule, self._fill(attrs), _blend_mode(attrs)))

    def _rect(self, attrs):
        if self._stroked(attrs):
            return
        x, y = _float(attrs.get("x")) or 0.0, _float(attrs.get("y")) or 0.0
        w, h = _float(attrs.get("width")), _float(attrs.get("height"))
        if w is None or h is None:
            raise ValueError("rect missing width/height")
        x1, y1 = float(f32(x) + f32(w)), float(f32(y) + f32(h))
        b = api.PathBuilder()
        b.move_to(api.Point(x, y)).line_to(api.Point(x, y1)).line_to(api.Point(x1, y1)).line_to(api.Point(x1, y)).line_to(api.Point(x, y))
        self._push(b, attrs)

    def _arc_to(self, b: api.PathBuilder, arc: _Arc, end):
        """<= quarter-turn rational quadratics, weight cos(sweep/2) (svg.rs:276-335)."""
        angle, left = arc.angle, arc.delta
        c, s = np.cos(arc.phi), np.sin(arc.phi)
        quarter = _PI / f32(2.0)
        incr = quarter if left > 0 else -quarter
        for _ in range(8):                                     # |delta| <= 2 pi: at most 4 pieces (+ rounding leftovers)
            if not left != 0:
                break
            theta = angle
            sweep = left if abs(left) <= quarter else incr
            angle = angle + sweep
            left = left - sweep
            half = sweep * f32(0.5)
            w = np.cos(half)
            p1x, p1y = np.cos(theta + half) / w * arc.rx, np.sin(theta + half) / w * arc.ry
            p2x, p2y = np.cos(theta + sweep) * arc.rx, np.sin(theta + sweep) * arc.ry
            q1 = (arc.cx + p1x * c - p1y * s, arc.cy + p1x * s + p1y * c)
            q2 = (arc.cx + p2x * c - p2y * s, arc.cy + p2x * s + p2y * c)
            b.rat_quad_to(self._t(*q1), self._t(*q2), float(w))
            end = (f32(q2[0]), f32(q2[1]))
        return end

    def _path(self, attrs):
        if self._stroked(attrs) or "d" not in attrs:
            return
        b = api.PathBuilder()
        start = None                                           # first point of the current sub-path, set by the first draw
        end = (f32(0.0), f32(0.0))
        qc = cc = None                                         # last quadratic / cubic control point (for T / S)
        T = self._t
        for cmd, absolute, a in path_segments(attrs["d"]):
            a = [f32(v) for v in a]
            pt = (lambda i: (a[i], a[i + 1])) if absolute else (lambda i: (end[0] + a[i], end[1] + a[i + 1]))
            new_qc = new_cc = None
            if cmd == "M":
                p = pt(0)
                b.move_to(T(*p))
                start, end = None, p
                qc = cc = None
                continue
            if cmd == "Z":
                if start is not None:
                    end, start = start, None
                    qc = cc = None
                continue
            if cmd == "A":
                p = pt(5)
                arc = _arc_center(a[0], a[1], a[2], a[3] != 0, a[4] != 0, end[0], end[1], p[0], p[1])
                if arc is not None:
                    prev = end
                    end = self._arc_to(b, arc, end)
                    if start is None:
                        start = prev
                qc = cc = None
                continue
            if cmd == "L":
                p = pt(0)
                b.line_to(T(*p))
            elif cmd == "H":
                p = (a[0], end[1]) if absolute else (end[0] + a[0], end[1] + f32(0.0))
                b.line_to(T(*p))
            elif cmd == "V":
                p = (end[0], a[0]) if absolute else (end[0] + f32(0.0), end[1] + a[0])
                b.line_to(T(*p))
            elif cmd == "Q":
                new_qc, p = pt(0), pt(2)
                b.quad_to(T(*new_qc), T(*p))
            elif cmd == "T":
                ref = qc if qc is not None else end
                new_qc, p = (end[0] * f32(2.0) - ref[0], end[1] * f32(2.0) - ref[1]), pt(0)
                b.quad_to(T(*new_qc), T(*p))
            elif cmd == "C":
                c1, new_cc, p = pt(0), pt(2), pt(4)
                b.cubic_to(T(*c1), T(*new_cc), T(*p))
            else:                                              # "S": the reflected point is ALSO what a following S reflects
                ref = cc if cc is not None else end           # (svg.rs:596-612 stores the reflection, not x2/y2)
                new_cc = (end[0] * f32(2.0) - ref[0], end[1] * f32(2.0) - ref[1])
                c2, p = pt(0), pt(2)
                b.cubic_to(T(*new_cc), T(*c2), T(*p))
            if start is None:
                start = end
            end, qc, cc = p, new_qc, new_cc
        self._push(b, attrs)

    # -- App::compose ------------------------------------------------------------------------------------------------
    def compose(self, composition: api.Composition) -> api.Composition:
        tf = api.GeomPresTransform.try_from([1.0, 0.0, 0.0, 1.0, -self.x, self.y])
        for order, (path, rule, fill, blend) in enumerate(self.paths):
            layer = composition.create_layer()
            layer.insert(path).set_transform(tf).set_props(
                api.Props(fill_rule=rule, func=api.Func.Draw(api.Style(fill=fill, blend_mode=blend))))
            composition.insert(api.Order(order), layer)
        return composition


def write_svg(shapes, width: int, height: int) -> str:
    """Serialise [(path-data string, fill attribute, extra attribute dict)] into an SVG document — used by the scene
    generators to hand the paris-like stand-in to the loader the way the real `paris-30k.svg` would arrive."""
    out = [f'<svg xmlns="http://www.w3.org/2000/svg" width="{width}" height="{height}" viewBox="0 0 {width} {height}">']
    for d, fill, extra in shapes:
        more = "".join(f' {k}="{v}"' for k, v in extra.items())
        out.append(f'<path d="{d}" fill="{fill}"{more}/>')
    out.append("</svg>")
    return "\n".join(out)
