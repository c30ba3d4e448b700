#!/usr/bin/env python3
"""Where a frame of bench.py's animated leg (the spaceship at 4K with a buffer-layer cache) spends its time on the host: seconds
inside every Context method (ctypes call included), the rest = Python of forma_amd.api.   python tools/animated_prof.py"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import forma_amd
from forma_amd import api, context
from forma_amd.spaceship import Spaceship

acc = collections.defaultdict(float); cnt = collections.Counter()
for name in ("set_geometry", "set_geoms", "set_styles", "set_images", "render", "tiles_written"):
    f = getattr(context.Context, name)
    def wrap(f=f, name=name):
        def g(self, *a, **k):
            t0 = time.perf_counter(); r = f(self, *a, **k); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
        return g
    setattr(context.Context, name, wrap())
W, H = 3840, 2160
for cached in (True, False):
    acc.clear(); cnt.clear()
    comp, r = api.Composition(), api.Renderer(device=0)
    cache = r.create_buffer_layer_cache() if cached else None
    game = Spaceship(api, W, H)
    buf = np.zeros(W * H * 4, np.uint8); lay = api.LinearLayout(W, W * 4, H)
    spent = 0.0; frames = 300
    for f in range(frames):
        game.compose(comp)
        b = api.BufferBuilder(buf, lay)
        if cached: b = b.layer_cache(cache)
        if f == 20: acc.clear(); cnt.clear(); spent = 0.0
        t0 = time.perf_counter()
        r.render(comp, b.build(), api.BGR1, api.Color(1, 1, 1, 0), None)
        spent += time.perf_counter() - t0
    n = frames - 20
    print("cached" if cached else "no cache", "us per frame %.1f" % (spent / n * 1e6), {k: (round(v / n * 1e6, 1), round(cnt[k] / n, 2)) for k, v in acc.items()},
          "python rest %.1f" % ((spent - sum(acc.values())) / n * 1e6))
    r._ctx.close()
