#!/usr/bin/env python3
"""The paris-like stand-in as an SVG file (scenes.paris_like_svg: 30 000 paths, gradients, blend modes), for the loader route:

    python tools/write_standin_svg.py /tmp/paris_like.svg && python bench.py --svg /tmp/paris_like.svg"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from forma_amd import scenes                               # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/paris_like.svg"
text = scenes.paris_like_svg()
with open(out, "w") as f:
    f.write(text)
print(out, len(text), "bytes")
