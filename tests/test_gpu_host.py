"""Product host mirror (forma_amd.api: C++ path preparation + HIP flatten kernel + composition
bookkeeping) against the oracle's flattener and the test-side scene builder."""
import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def both_paths(cmds):
    """Replay one command list into the oracle builder and the product builder."""
    from forma_amd import api
    po = orc.Path(); pb = api.PathBuilder()
    for c in cmds:
        k, a = c[0], c[1:]
        if k == "M": po.move_to(*a); pb.move_to(api.Point(*a))
        elif k == "L": po.line_to(*a); pb.line_to(api.Point(*a))
        elif k == "Q": po.quad_to(*a); pb.quad_to(api.Point(a[0], a[1]), api.Point(a[2], a[3]))
        elif k == "C": po.cubic_to(*a); pb.cubic_to(api.Point(a[0], a[1]), api.Point(a[2], a[3]), api.Point(a[4], a[5]))
        elif k == "RQ": po.rat_quad_to(*a); pb.rat_quad_to(api.Point(a[0], a[1]), api.Point(a[2], a[3]), a[4])
        elif k == "RC": po.rat_cubic_to(*a); pb.rat_cubic_to(api.Point(a[0], a[1]), api.Point(a[2], a[3]), api.Point(a[4], a[5]), a[6], a[7])
    return po.build(), pb.build()


def rand_cmds(rng, n):
    cmds = [("M", *map(float, rng.uniform(-50, 600, 2).astype(np.float32)))]
    for _ in range(n):
        k = rng.integers(0, 6)
        p = [float(v) for v in rng.uniform(-50, 600, 6).astype(np.float32)]
        if k == 0: cmds.append(("L", p[0], p[1]))
        elif k == 1: cmds.append(("Q", *p[:4]))
        elif k == 2: cmds.append(("C", *p))
        elif k == 3: cmds.append(("RQ", *p[:4], float(np.float32(rng.uniform(0.2, 3.0)))))
        elif k == 4: cmds.append(("RC", *p, float(np.float32(rng.uniform(0.3, 2.0))), float(np.float32(rng.uniform(0.3, 2.0)))))
        else: cmds.append(("M", p[0], p[1]))
    return cmds


def test_flatten_matches_oracle_bit_exact():
    from forma_amd import api
    rng = np.random.default_rng(3)
    o = orc.Oracle()
    comp = api.Composition()
    r = api.Renderer(0)
    xs, ys, ls = [], [], []
    for i in range(200):
        po, pb = both_paths(rand_cmds(rng, int(rng.integers(1, 12))))
        if i % 5 == 0:   # GeomPresTransform (cheap per-point transform) and a non-affine 3x3
            t9 = [0.8, 0.1, 5.0, -0.1, 0.8, 7.0, 0.0, 0.0, 1.0] if i % 10 == 0 else [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0005, 0.0, 1.0]
            pb = pb.transform(t9)
            if i % 10 == 0: po.affine = (0.8, -0.1, 0.1, 0.8, 5.0, 7.0)
            else: po.transform9(t9)
        comp.get_mut_or_insert_default(api.Order(i)).insert(pb)
        x, y, nc = o.flatten(po)
        if len(x):
            pid = np.where(nc != 0, S.NONE, i).astype(np.uint32); pid[-1] = S.NONE
            xs.append(x); ys.append(y); ls.append(pid)
    r._upload_geometry(comp)
    t = r.host_tables
    # slots are assigned in first-use order == i for non-empty paths: compare through the order mapping
    x = np.concatenate(xs); y = np.concatenate(ys)
    assert len(t["x"]) == len(x)
    assert np.array_equal(t["x"].view(np.uint32), x.view(np.uint32))
    assert np.array_equal(t["y"].view(np.uint32), y.view(np.uint32))
    none_o = np.concatenate(ls)[:-1] == S.NONE
    assert np.array_equal(t["line_slot"] == S.NONE, none_o)


def test_renderer_matches_oracle_on_e2e_scene():
    """The same scene built through the product API and through the test-side builder renders identically."""
    from forma_amd import api
    W, H, PAD = 64.0, 64.0, 8.0
    comp = api.Composition()
    sq = (api.PathBuilder().move_to(api.Point(PAD, PAD)).line_to(api.Point(PAD, H - PAD)).line_to(api.Point(W - PAD, H - PAD))
          .line_to(api.Point(W - PAD, PAD)).build())
    w = float(np.sqrt(np.float32(2.0)) / np.float32(2.0))
    cx, cy, rad = W * 0.5, H * 0.5, W * 0.5 - PAD
    circ = (api.PathBuilder().move_to(api.Point(cx + rad, cy)).rat_quad_to(api.Point(cx + rad, cy - rad), api.Point(cx, cy - rad), w)
            .rat_quad_to(api.Point(cx - rad, cy - rad), api.Point(cx - rad, cy), w).rat_quad_to(api.Point(cx - rad, cy + rad), api.Point(cx, cy + rad), w)
            .rat_quad_to(api.Point(cx + rad, cy + rad), api.Point(cx + rad, cy), w).build())
    comp.get_mut_or_insert_default(api.Order(0)).insert(sq).set_props(api.Props(func=api.Func.Draw(api.Style(fill=api.Fill.Solid(api.Color(0, 0, 0, 0.7))))))
    gb = api.GradientBuilder(api.Point(PAD, 0.0), api.Point(W - PAD, 0.0))
    gb.color(api.Color(0, 0, 1, 1)).color(api.Color(1, 1, 1, 1)).color(api.Color(1, 0, 0, 1))
    comp.get_mut_or_insert_default(api.Order(4)).insert(circ).set_props(
        api.Props(fill_rule=api.FillRule.EvenOdd, func=api.Func.Draw(api.Style(fill=api.Fill.Gradient(gb.build()), blend_mode=api.BlendMode.Multiply))))
    r = api.Renderer(0)
    img = np.zeros(64 * 64 * 4, np.uint8)
    lay = api.LinearLayout(64, 256, 64)
    r.render(comp, api.BufferBuilder(img, lay).build(), api.RGBA, api.Color(1, 1, 1, 0), None)

    c2 = S.Composition()
    c2.get_mut_or_insert_default(0).insert(S.square()).set_props(S.solid((0, 0, 0, 0.7)))
    c2.get_mut_or_insert_default(4).insert(S.circle()).set_props(
        S.Props(fill_rule="EvenOdd", fill=S.gradient((PAD, 0.0), (W - PAD, 0.0), [(0, 0, 1, 1), (1, 1, 1, 1), (1, 0, 0, 1)]), blend_mode="Multiply"))
    o = orc.Oracle()
    S.load(o, c2.tables(o))
    want = o.render(64, 64)
    assert np.array_equal(want.reshape(-1), img)


def test_layer_bookkeeping_like_reference():
    """composition/mod.rs tests: layer replace / disable / clear / transform, through the product renderer."""
    from forma_amd import api
    def pixel(x, y):
        return (api.PathBuilder().move_to(api.Point(x, y)).line_to(api.Point(x, y + 1)).line_to(api.Point(x + 1, y + 1))
                .line_to(api.Point(x + 1, y)).line_to(api.Point(x, y)).build())
    def solid(c):
        return api.Props(func=api.Func.Draw(api.Style(fill=api.Fill.Solid(c))))
    RED, GREEN, BLACK = api.Color(1, 0, 0, 1), api.Color(0, 1, 0, 1), api.Color(0, 0, 0, 1)
    r = api.Renderer(0)
    def render(comp, w=3, h=1, clear=api.Color(1, 1, 1, 0)):
        img = np.zeros(w * h * 4, np.uint8)
        r.render(comp, api.BufferBuilder(img, api.LinearLayout(w, w * 4, h)).build(), api.RGBA, clear, None)
        return img.reshape(h, w, 4)
    # one_pixel / two_pixels_same_layer (composition/mod.rs:568-611)
    comp = api.Composition()
    comp.get_mut_or_insert_default(api.Order(0)).insert(pixel(1, 0)).set_props(solid(RED))
    assert render(comp).tolist() == [[[255, 255, 255, 0], [255, 0, 0, 255], [255, 255, 255, 0]]]
    comp.get_mut(api.Order(0)).insert(pixel(2, 0))
    assert render(comp).tolist() == [[[255, 255, 255, 0], [255, 0, 0, 255], [255, 0, 0, 255]]]
    # one_pixel_translated (:613-641): half-pixel translation -> 0xBB coverage of black over white alpha 0
    comp = api.Composition()
    comp.get_mut_or_insert_default(api.Order(0)).insert(pixel(1, 0)).set_props(solid(BLACK)).set_transform(
        api.GeomPresTransform.try_from([1.0, 0.0, 0.0, 1.0, 0.5, 0.0]))
    out = render(comp)
    assert out[0, 1].tolist() == [0xBB, 0xBB, 0xBB, 0x80] and out[0, 2].tolist() == [0xBB, 0xBB, 0xBB, 0x80]
    # insert_over_layer / layer_replace_remove (:703-789)
    comp = api.Composition()
    comp.get_mut_or_insert_default(api.Order(0)).insert(pixel(0, 0)).set_props(solid(RED))
    l1 = comp.create_layer(); l1.insert(pixel(1, 0)).set_props(solid(GREEN))
    old = comp.insert(api.Order(0), l1)
    assert old is not None
    assert render(comp).tolist() == [[[255, 255, 255, 0], [0, 255, 0, 255], [255, 255, 255, 0]]]
    comp.remove(api.Order(0))
    assert render(comp).tolist() == [[[255, 255, 255, 0]] * 3]
    # layer_clear (:791-830) and disable
    comp = api.Composition()
    comp.get_mut_or_insert_default(api.Order(0)).insert(pixel(0, 0)).set_props(solid(RED))
    comp.get_mut_or_insert_default(api.Order(1)).insert(pixel(1, 0)).set_props(solid(GREEN))
    comp.get_mut(api.Order(0)).clear()
    assert render(comp).tolist() == [[[255, 255, 255, 0], [0, 255, 0, 255], [255, 255, 255, 0]]]
    comp.get_mut(api.Order(1)).disable()
    assert render(comp).tolist() == [[[255, 255, 255, 0]] * 3]
    with pytest.raises(api.OrderError):
        api.Order((1 << 21))
    with pytest.raises(api.GeomPresTransformError):
        api.GeomPresTransform.try_from([2.0, 0.0, 0.0, 1.0, 0.0, 0.0])


def test_buffer_layer_cache_through_the_product_api():
    """Reference composition/mod.rs `render_changed_layers_only` and the BufferLayerCache doc test (buffer/mod.rs:125-165):
    with a cache attached, a frame in which nothing changed leaves the caller's buffer untouched; changing one layer
    repaints only the tiles it touches."""
    from forma_amd import api
    def rect(x0, y0, x1, y1):
        return (api.PathBuilder().move_to(api.Point(x0, y0)).line_to(api.Point(x0, y1)).line_to(api.Point(x1, y1))
                .line_to(api.Point(x1, y0)).line_to(api.Point(x0, y0)).build())
    def solid(c):
        return api.Props(func=api.Func.Draw(api.Style(fill=api.Fill.Solid(c))))
    W, H = 48, 32                                                         # 3 x 2 tiles
    comp = api.Composition()
    comp.get_mut_or_insert_default(api.Order(0)).insert(rect(2, 2, 10, 10)).set_props(solid(api.Color(1, 0, 0, 1)))     # tile (0,0)
    comp.get_mut_or_insert_default(api.Order(1)).insert(rect(34, 18, 44, 28)).set_props(solid(api.Color(0, 1, 0, 1)))   # tile (2,1)
    r = api.Renderer(0)
    cache = r.create_buffer_layer_cache()
    assert cache is not None
    lay = api.LinearLayout(W, W * 4, H)
    img = np.zeros(W * H * 4, np.uint8)
    white = api.Color(1, 1, 1, 1)
    r.render(comp, api.BufferBuilder(img, lay).layer_cache(cache).build(), api.RGBA, white, None)
    px = img.reshape(H, W, 4)
    assert tuple(px[5, 5]) == (255, 0, 0, 255) and tuple(px[20, 40]) == (0, 255, 0, 255) and tuple(px[20, 5]) == (255, 255, 255, 255)
    # frame 2: nothing changed, buffer wiped by the caller -> stays wiped
    img[:] = 0
    r.render(comp, api.BufferBuilder(img, lay).layer_cache(cache).build(), api.RGBA, white, None)
    assert not img.any()
    # frame 3: only layer 1 changes colour -> only tile (2,1) is rewritten
    comp.get_mut_or_insert_default(api.Order(1)).set_props(solid(api.Color(0, 0, 1, 1)))
    r.render(comp, api.BufferBuilder(img, lay).layer_cache(cache).build(), api.RGBA, white, None)
    px = img.reshape(H, W, 4)
    assert tuple(px[20, 40]) == (0, 0, 255, 255)
    assert not px[:16].any() and not px[16:, :32].any()                   # the other five tiles were left alone
    # cache.clear(): everything is drawn again
    cache.clear()
    r.render(comp, api.BufferBuilder(img, lay).layer_cache(cache).build(), api.RGBA, white, None)
    assert tuple(px[5, 5]) == (255, 0, 0, 255) and tuple(px[20, 5]) == (255, 255, 255, 255)


def test_animated_scene_cached_frames_equal_full_repaints():
    """BASELINE config 5 (spaceship demo with the per-tile damage optimizer) at a reduced size, through the product API:
    a buffer carried from frame to frame with a BufferLayerCache must hold, after every frame, exactly the pixels a full
    repaint of that frame produces (reference demo/src/demos/spaceship.rs + cpu/buffer/layer_cache semantics)."""
    from forma_amd import api, scenes
    W, H = 960, 544
    comp, moving, state = scenes.spaceship(W, H, enemies=24, stars=60, seed=5)
    r = api.Renderer(0)
    cache = r.create_buffer_layer_cache()
    lay = api.LinearLayout(W, W * 4, H)
    carried = np.zeros(W * H * 4, np.uint8)
    black = api.Color(0, 0, 0, 1)
    for f in range(5):
        xf = scenes.spaceship_transforms(state, f / 60.0)
        for o, t in zip(moving, xf):
            comp.get_mut(api.Order(o)).set_transform(api.GeomPresTransform.try_from([float(v) for v in t]))
        r.render(comp, api.BufferBuilder(carried, lay).layer_cache(cache).build(), api.RGBA, black, None)
        fresh = np.zeros(W * H * 4, np.uint8)
        r.render(comp, api.BufferBuilder(fresh, lay).build(), api.RGBA, black, None)
        assert np.array_equal(carried, fresh), f"frame {f}"
        assert fresh.reshape(H, W, 4)[..., :3].any()


def _random_svg(n=160, w=512, h=384, seed=17):
    """An SVG document exercising every construct the loader knows: groups with transforms / fills / opacities, all path
    commands incl. arcs, rects, both gradient kinds, blend modes, even-odd."""
    rng = np.random.default_rng(seed)
    out = [f'<svg xmlns="http://www.w3.org/2000/svg" width="{w}" height="{h}">',
           '<linearGradient id="lg" gradientUnits="userSpaceOnUse" x1="40" y1="30" x2="400" y2="300">'
           '<stop offset="0%" stop-color="#ff8000"/><stop offset="50%" stop-color="rgb(0,128,255)" stop-opacity="0.7"/>'
           '<stop offset="100%" stop-color="seagreen"/></linearGradient>',
           '<radialGradient id="rg" gradientUnits="userSpaceOnUse" cx="256" cy="192" r="180">'
           '<stop offset="10%" stop-color="white"/><stop offset="90%" stop-color="#203040" stop-opacity="0.5"/></radialGradient>']
    modes = ["normal", "multiply", "screen", "overlay", "darken", "lighten", "color-dodge", "color-burn", "hard-light",
             "soft-light", "difference", "exclusion", "hue", "saturation", "color", "luminosity"]
    for i in range(n):
        x, y = rng.uniform(0, w - 60), rng.uniform(0, h - 60)
        s = rng.uniform(10, 90)
        kind = i % 8
        fill = "#%02x%02x%02x" % tuple(int(v) for v in rng.integers(0, 256, 3))
        if i % 11 == 0:
            fill = "url(#lg)"
        elif i % 13 == 0:
            fill = "url(#rg)"
        extra = ""
        if i % 7 == 0:
            extra += f' style="mix-blend-mode:{modes[(i // 7) % 16]}"'
        if i % 5 == 0:
            extra += ' fill-opacity="0.6"'
        if i % 9 == 0:
            extra += ' fill-rule="evenodd"'
        if kind == 0:
            d = f"M{x:.2f} {y:.2f}l{s:.2f} 0 0 {s:.2f}-{s:.2f} 0z m{s/4:.2f} {s/4:.2f}h{s/2:.2f}v{s/2:.2f}h-{s/2:.2f}z"
        elif kind == 1:
            d = f"M{x:.2f},{y:.2f}C{x+s:.2f},{y:.2f} {x+s:.2f},{y+s:.2f} {x:.2f},{y+s:.2f}S{x-s/2:.2f},{y+s/2:.2f} {x:.2f},{y:.2f}"
        elif kind == 2:
            d = f"M{x:.2f} {y:.2f}q{s:.2f} {s/3:.2f} {s/2:.2f} {s:.2f}t-{s/2:.2f} -{s/4:.2f}T{x:.2f} {y:.2f}"
        elif kind == 3:
            d = f"M{x:.2f} {y+s/2:.2f}a{s/2:.2f} {s/3:.2f} 0 1 0 {s:.2f} 0a{s/2:.2f} {s/3:.2f} 0 1 0 -{s:.2f} 0"
        elif kind == 4:
            d = f"M{x:.2f} {y:.2f}A{s:.2f} {s/2:.2f} 30 0 1 {x+s:.2f} {y+s/2:.2f}L{x:.2f} {y+s:.2f}Z"
        elif kind == 5:
            out.append(f'<rect x="{x:.2f}" y="{y:.2f}" width="{s:.2f}" height="{s*0.6:.2f}" fill="{fill}"{extra}/>')
            continue
        elif kind == 6:
            out.append(f'<g transform="translate({x:.2f} {y:.2f}) rotate({rng.uniform(0, 90):.1f}) scale(0.8)" opacity="0.8" fill="{fill}">'
                       f'<g opacity="0.9"><path d="M0 0L{s:.2f} 0L{s/2:.2f} {s:.2f}z"{extra}/></g></g>')
            continue
        else:
            d = f"M{x:.2f} {y:.2f}c{s:.2f} 0 {s:.2f} {s:.2f} 0 {s:.2f}s-{s:.2f} -{s/2:.2f} 0 -{s:.2f}"
        out.append(f'<path d="{d}" fill="{fill}"{extra}/>')
    out.append('<path d="M0 0L10 10" stroke="#000"/></svg>')
    return "\n".join(out)


def test_svg_loader_route_matches_oracle():
    """SVG text -> forma_amd.svg (mirror of demo/src/demos/svg.rs) -> Composition -> hip Renderer.  The
    tables the renderer uploaded are replayed through the CPU oracle: streams bit-exact, image within 1 code value."""
    from forma_amd import api, svg
    W, H = 512, 384
    doc = svg.Svg(_random_svg(), 1.0, is_text=True)
    assert len(doc.paths) == 160
    comp = doc.compose(api.Composition())
    r = api.Renderer(0)
    img = np.zeros(W * H * 4, np.uint8)
    r.render(comp, api.BufferBuilder(img, api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    o = orc.Oracle()
    S.load(o, r.host_tables)
    want = o.render(W, H, clear=(1.0, 1.0, 1.0, 1.0))
    got = img.reshape(want.shape)
    assert np.array_equal(o.segments(0), r._ctx.segments(0))          # unsorted stream, reference order
    assert np.array_equal(o.segments(1), r._ctx.segments(1))          # sorted stream
    diff = np.abs(want.astype(np.int16) - got.astype(np.int16))
    assert diff.max() <= 1, f"max diff {diff.max()}"
    assert (got != 255).any()
    # scale = 2 renders the same picture twice as large (Path::transform after parsing, svg.rs:217-220)
    doc2 = svg.Svg(_random_svg(), 2.0, is_text=True)
    r2 = api.Renderer(0)
    img2 = np.zeros(4 * W * H * 4, np.uint8)
    r2.render(doc2.compose(api.Composition()), api.BufferBuilder(img2, api.LinearLayout(2 * W, 2 * W * 4, 2 * H)).build(),
              api.RGBA, api.Color(1, 1, 1, 1), None)
    o2 = orc.Oracle()
    S.load(o2, r2.host_tables)
    want2 = o2.render(2 * W, 2 * H, clear=(1.0, 1.0, 1.0, 1.0))
    assert np.abs(want2.astype(np.int16) - img2.reshape(want2.shape).astype(np.int16)).max() <= 1


def test_paris_like_30k_4k_full_size():
    """BASELINE.json configs[2] — the workload bench.py reports — at full size through the product API: 30 000 layers
    (solid / linear / radial fills, 16 blend modes, translucent) on 3840 x 2160, ~13.8 M pixel segments.  Both segment
    streams bit-exact against the oracle, image within 1 code value, on the synchronous first frame and on a
    read-back-free second frame."""
    from forma_amd import api, scenes
    fn, W, H = scenes.WORKLOADS["paris-like-30k-4k"]
    comp = fn()
    r = api.Renderer(0)
    img = np.zeros(W * H * 4, np.uint8)
    lay = api.LinearLayout(W, W * 4, H)
    r.render(comp, api.BufferBuilder(img, lay).build(), api.RGBA, api.Color(1, 1, 1, 1), None, timings=True)
    assert r.last_timings["n_segments"] > 13_000_000
    o = orc.Oracle()
    S.load(o, r.host_tables)
    want = o.render(W, H, clear=(1.0, 1.0, 1.0, 1.0))
    assert np.array_equal(o.segments(0), r._ctx.segments(0))
    assert np.array_equal(o.segments(1), r._ctx.segments(1))
    d = np.abs(want.astype(np.int16) - img.reshape(want.shape).astype(np.int16))
    assert d.max() <= 1, (int(d.max()), int((d > 0).sum()))
    img2 = np.zeros(W * H * 4, np.uint8)
    r.render(comp, api.BufferBuilder(img2, lay).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    assert np.array_equal(img, img2)


def test_paris_like_30k_4k_with_three_frames_in_flight_matches_the_oracle():
    """BASELINE.json configs[2] through the path `bench.py` reports as `value`: ONE context with three frame slots, device-resident
    frames enqueued round-robin.  The most recent frame's sorted stream and image are compared with the oracle after the slots
    have each run their synchronous and their read-back-free frames, and again after a change of the clear colour in flight."""
    import forma_amd
    from forma_amd import api, scenes
    fn, W, H = scenes.WORKLOADS["paris-like-30k-4k"]
    r = api.Renderer(0)
    r.render(fn(), api.BufferBuilder(np.zeros(W * H * 4, np.uint8), api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    t = r.host_tables
    o = orc.Oracle()
    S.load(o, t)
    c = forma_amd.Context(0, frames_in_flight=3)
    S.load(c, t)
    clears = [(1.0, 1.0, 1.0, 1.0), (0.1, 0.2, 0.3, 1.0)]
    want = {cl: o.render(W, H, clear=cl) for cl in clears}
    sorted_ref = o.segments(1)
    for k in range(14):
        cl = clears[(k // 5) % 2]
        assert c.render(W, H, clear=cl, device_only=True) is None
        if k in (8, 13):
            d = np.abs(c.read_image(W, H).astype(np.int16) - want[cl].astype(np.int16))
            assert d.max() <= 1, (k, int(d.max()), int((d > 0).sum()))
            assert np.array_equal(c.segments(1), sorted_ref), k
    c.sync()
    c.close()


def test_svg_loader_route_at_full_size_matches_the_api_built_stand_in():
    """SURVEY §8 f1 at BASELINE configs[2] scale: the stand-in serialised as SVG TEXT (30 000 <path>s, linear / radial
    gradients, mix-blend-mode, fill-opacity — `scenes.paris_like_svg`) goes through the loader (`forma_amd.svg`, after
    demo/src/demos/svg.rs:337-863,904-920) and the product API.  Geometry round-trips to the bit, so both pixel-segment
    streams must equal the API-built stand-in's; the loaded scene's image (its colours went through 8-bit sRGB) is checked
    against the oracle on the loaded scene's own tables; a read-back-free second frame repeats the first."""
    import time
    from forma_amd import api, scenes, svg
    _, W, H = scenes.WORKLOADS["paris-like-30k-4k"]
    text = scenes.paris_like_svg()
    t0 = time.perf_counter()
    loaded = svg.Svg(text, 1.0, is_text=True)
    comp = loaded.compose(api.Composition())
    load_s = time.perf_counter() - t0
    assert len(comp) == 30000 and load_s < 60.0
    kinds = {}
    for _, layer in comp.layers_iter():
        st = layer.props().func[1]
        kinds[(st.fill[0], st.blend_mode != "Over")] = kinds.get((st.fill[0], st.blend_mode != "Over"), 0) + 1
    assert kinds.get(("gradient", False), 0) > 2000 and sum(v for (k, b), v in kinds.items() if b) > 1000
    lay = api.LinearLayout(W, W * 4, H)
    r = api.Renderer(0)
    img = np.zeros(W * H * 4, np.uint8)
    r.render(comp, api.BufferBuilder(img, lay).build(), api.RGBA, api.Color(1, 1, 1, 1), None, timings=True)
    n = r.last_timings["n_segments"]
    u_svg, s_svg = r._ctx.segments(0), r._ctx.segments(1)
    o = orc.Oracle()
    S.load(o, r.host_tables)
    want = o.render(W, H, clear=(1.0, 1.0, 1.0, 1.0))
    d = np.abs(want.astype(np.int16) - img.reshape(want.shape).astype(np.int16))
    assert d.max() <= 1, (int(d.max()), int((d > 0).sum()))
    img2 = np.zeros(W * H * 4, np.uint8)
    r.render(comp, api.BufferBuilder(img2, lay).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    assert np.array_equal(img, img2)
    del o, want
    r2 = api.Renderer(0)                                       # the API-built stand-in: same geometry, so the same streams
    r2.render(scenes.paris_like(), api.BufferBuilder(np.zeros(W * H * 4, np.uint8), lay).build(), api.RGBA, api.Color(1, 1, 1, 1), None,
              timings=True)
    assert r2.last_timings["n_segments"] == n > 13_000_000
    assert np.array_equal(u_svg, r2._ctx.segments(0))
    assert np.array_equal(s_svg, r2._ctx.segments(1))


def test_circles_demo_scene_matches_oracle():
    """The reference demo's `circles` mode (demo/src/demos/circles.rs) through the product API: 3 000 translucent discs
    on 1000 x 1000 — tiles ~20 layers deep, nothing opaque, so every layer is blended."""
    from forma_amd import api, scenes
    comp = scenes.circles(3000)
    r = api.Renderer(0)
    W = H = 1000
    img = np.zeros(W * H * 4, np.uint8)
    r.render(comp, api.BufferBuilder(img, api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    o = orc.Oracle()
    S.load(o, r.host_tables)
    want = o.render(W, H, clear=(1.0, 1.0, 1.0, 1.0))
    assert np.array_equal(o.segments(1), r._ctx.segments(1))
    assert np.abs(want.astype(np.int16) - img.reshape(want.shape).astype(np.int16)).max() <= 1
