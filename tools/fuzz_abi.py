"""No-crash fuzz of the C ABI (product only, nothing is compared with the oracle — it is not hardened against such inputs):
random scenes with wild transforms (NaN, +-inf, 1e20), orders without a style, flags, arbitrary float bits in colours / stops /
texture transforms, NaN clear colours, any channel codes, degenerate crops, 1-pixel canvases.  Every call must return (a frame
or an error code) and two renders of the same input must be the same bytes.
    python tools/fuzz_abi.py 0 1500        (on a GPU box; ~3 s per 1000 scenes)"""
import os, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import scene as S
from oracle import oracle as orc
import forma_amd
from forma_amd._lib import FormaError
o = orc.Oracle(); c = forma_amd.Context(0)
specials = np.array([np.nan, np.inf, -np.inf, 3e38, -3e38, 1e20, 0.0, -0.0, 1e-30, 1e9, -1e9], np.float32)
stats = {}
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(120000 + seed)
    w, h = int(rng.integers(1, 400)), int(rng.integers(1, 300))
    t = dict(S.random_mixed(n=int(rng.integers(1, 40)), width=max(w, 16), height=max(h, 16), seed=121000 + seed).tables(o))
    g = t["geoms"].copy()
    for _ in range(int(rng.integers(0, 4))):                 # wild transforms / orders / flags
        i = int(rng.integers(0, len(g)))
        k = int(rng.integers(0, 4))
        if k == 0: g["xf"][i][int(rng.integers(0, 6))] = specials[int(rng.integers(0, len(specials)))]
        elif k == 1: g["flags"][i] = int(rng.integers(0, 4))
        elif k == 2: g["order"][i] = int(rng.integers(0, len(t["style_offsets"]) + 3))
        else: g["xf"][i] = rng.normal(size=6).astype(np.float32) * float(np.exp(rng.uniform(-5, 12)))
    t["geoms"] = g
    sw = t["style_words"].copy()
    for _ in range(int(rng.integers(0, 4))):                 # colours / stops / transforms of styles: any float bits
        i = int(rng.integers(0, len(sw)))
        # never a header word: those are validated structure
        if i not in set(int(v) for v in t["style_offsets"] if v != 0xFFFFFFFF) and (i - 1) not in set(int(v) for v in t["style_offsets"] if v != 0xFFFFFFFF):
            sw[i] = specials[int(rng.integers(0, len(specials)))].view(np.uint32) if rng.random() < 0.7 else np.uint32(rng.integers(0, 2**32))
    t["style_words"] = sw
    clear = tuple(float(specials[int(rng.integers(0, len(specials)))]) if rng.random() < 0.2 else float(rng.random()) for _ in range(4))
    ch = tuple(int(v) for v in rng.integers(0, 6, 4))
    crop = None
    if rng.random() < 0.3:
        crop = tuple(int(v) for v in rng.integers(0, 500, 4))
    try:
        S.load(c, t)
        a = c.render(w, h, clear=clear, channels=ch, crop=crop)
        b = c.render(w, h, clear=clear, channels=ch, crop=crop)
        c.render(w, h, clear=clear, channels=ch, crop=crop, device_only=True)
        d = c.read_image(w, h)
        r = "ok" if (np.array_equal(a, b) and (crop is not None or np.array_equal(a, d))) else "NONDETERMINISTIC"
        if r != "ok": print("seed", seed, r, w, h, ch, crop, flush=True)
    except FormaError as e:
        r = "err %d" % e.code
    stats[r] = stats.get(r, 0) + 1
print(stats)
