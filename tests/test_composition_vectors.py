"""The reference's composition-level tests (forma/src/composition/mod.rs:482-1428 and the BufferLayerCache doc test,
cpu/buffer/mod.rs:125-165), written ONCE against forma's public API shape and run against two backends:

* `oracle` (CPU, `-m "not gpu"`): tests/ref_api.py = the reference's Composition / Layer / Renderer bookkeeping over
  the oracle.  This is what pins the oracle's frame path — including the buffer-layer cache, `is_unchanged` bits,
  clear-colour caching, emptied tiles, per-cache state and size changes — to the reference's expected buffers.
* `hip` (`-m gpu`): the product mirror `forma_amd.api` over libforma_hip.so.  Same test bodies, same expected bytes.

Every test is the reference test of the same name; expected values are the reference's literals."""
import numpy as np
import pytest

TILE_WIDTH = TILE_HEIGHT = 16
BLACK_SRGB = [0x00, 0x00, 0x00, 0xFF]
GRAY_SRGB = [0xBB, 0xBB, 0xBB, 0xFF]
GRAY_ALPHA_50_SRGB = [0xBB, 0xBB, 0xBB, 0x80]
WHITE_ALPHA_0_SRGB = [0xFF, 0xFF, 0xFF, 0x00]
RED_SRGB = [0xFF, 0x00, 0x00, 0xFF]
GREEN_SRGB = [0x00, 0xFF, 0x00, 0xFF]
RED_50_GREEN_50_SRGB = [0xBB, 0xBB, 0x00, 0xFF]


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def api(request):
    if request.param == "oracle":
        import ref_api
        return ref_api
    from forma_amd import api as product
    return product


class K:
    """colours / helpers of the reference test module (composition/mod.rs:413-480), bound to one backend"""

    def __init__(self, api):
        self.api = api
        C = api.Color
        self.BLACK, self.BLACK_ALPHA_50, self.GRAY = C(0, 0, 0, 1), C(0, 0, 0, 0.5), C(0.5, 0.5, 0.5, 1)
        self.WHITE_TRANSPARENT, self.RED, self.GREEN = C(1, 1, 1, 0), C(1, 0, 0, 1), C(0, 1, 0, 1)

    def pixel_path(self, x, y):
        a = self.api; P = a.Point
        return (a.PathBuilder().move_to(P(x, y)).line_to(P(x, y + 1)).line_to(P(x + 1, y + 1)).line_to(P(x + 1, y))
                .line_to(P(x, y)).build())

    def solid(self, color):
        a = self.api
        return a.Props(func=a.Func.Draw(a.Style(fill=a.Fill.Solid(color))))

    def buf(self, *pixels):
        return np.asarray(pixels, np.uint8).reshape(-1).copy()

    def render(self, renderer, comp, buffer, layout, clear, cache=None, crop=None, flusher=None):
        a = self.api
        b = a.BufferBuilder(buffer, layout)
        if cache is not None:
            b = b.layer_cache(cache)
        if flusher is not None:
            b = b.flusher(flusher)
        renderer.render(comp, b.build(), a.RGBA, clear, crop)

    def order(self, v):
        return self.api.Order.new(v)

    def translate(self, tx, ty):
        return self.api.GeomPresTransform.try_from([1.0, 0.0, 0.0, 1.0, tx, ty])


@pytest.fixture
def k(api):
    return K(api)


def px(buffer, i):
    return buffer[4 * i: 4 * i + 4].tolist()


def test_composition_len(api, k):                                   # :482-493
    comp = api.Composition()
    assert comp.is_empty() and len(comp) == 0
    comp.get_mut_or_insert_default(k.order(0))
    assert not comp.is_empty() and len(comp) == 1


def test_background_color_clear(api, k):                            # :495-512
    buffer = k.buf(GREEN_SRGB)
    k.render(api.Renderer(), api.Composition(), buffer, api.LinearLayout(1, 4, 1), k.RED)
    assert buffer.tolist() == RED_SRGB


def test_background_color_clear_when_changed(api, k):               # :514-561
    buffer = k.buf(GREEN_SRGB)
    layout = api.LinearLayout(1, 4, 1)
    comp, r = api.Composition(), api.Renderer()
    cache = r.create_buffer_layer_cache()
    k.render(r, comp, buffer, layout, k.RED, cache)
    assert buffer.tolist() == RED_SRGB
    buffer[:] = GREEN_SRGB
    k.render(r, comp, buffer, layout, k.RED, cache)
    assert buffer.tolist() == GREEN_SRGB                            # Skip clearing if the color is the same.
    k.render(r, comp, buffer, layout, k.BLACK, cache)
    assert buffer.tolist() == BLACK_SRGB


def test_one_pixel(api, k):                                         # :563-585
    buffer = k.buf(*[GREEN_SRGB] * 3)
    comp, r = api.Composition(), api.Renderer()
    layer = comp.create_layer()
    layer.insert(k.pixel_path(1, 0)).set_props(k.solid(k.RED))
    comp.insert(k.order(0), layer)
    k.render(r, comp, buffer, api.LinearLayout(3, 12, 1), k.GREEN)
    assert buffer.tolist() == GREEN_SRGB + RED_SRGB + GREEN_SRGB


def test_two_pixels_same_layer(api, k):                             # :587-611
    buffer = k.buf(*[GREEN_SRGB] * 3)
    comp, r = api.Composition(), api.Renderer()
    layer = comp.create_layer()
    layer.insert(k.pixel_path(1, 0)).insert(k.pixel_path(2, 0)).set_props(k.solid(k.RED))
    comp.insert(k.order(0), layer)
    k.render(r, comp, buffer, api.LinearLayout(3, 12, 1), k.GREEN)
    assert buffer.tolist() == GREEN_SRGB + RED_SRGB + RED_SRGB


def test_one_pixel_translated(api, k):                              # :613-641
    buffer = k.buf(*[GREEN_SRGB] * 3)
    comp, r = api.Composition(), api.Renderer()
    layer = comp.create_layer()
    layer.insert(k.pixel_path(1, 0)).set_props(k.solid(k.RED)).set_transform(k.translate(0.5, 0.0))
    comp.insert(k.order(0), layer)
    k.render(r, comp, buffer, api.LinearLayout(3, 12, 1), k.GREEN)
    assert buffer.tolist() == GREEN_SRGB + RED_50_GREEN_50_SRGB + RED_50_GREEN_50_SRGB


def test_one_pixel_rotated(api, k):                                 # :643-680
    buffer = k.buf(*[GREEN_SRGB] * 3)
    comp, r = api.Composition(), api.Renderer()
    angle = np.float32(-np.pi / 2.0)
    c, s = float(np.cos(angle, dtype=np.float32)), float(np.sin(angle, dtype=np.float32))
    layer = comp.create_layer()
    layer.insert(k.pixel_path(-1, 1)).set_props(k.solid(k.RED)).set_transform(api.GeomPresTransform.try_from([c, -s, s, c, 0.0, 0.0]))
    comp.insert(k.order(0), layer)
    k.render(r, comp, buffer, api.LinearLayout(3, 12, 1), k.GREEN)
    assert buffer.tolist() == GREEN_SRGB + RED_SRGB + GREEN_SRGB


def test_clear_and_resize(api, k):                                  # :682-762
    buffer = k.buf(*[GREEN_SRGB] * 4)
    comp, r = api.Composition(), api.Renderer()
    o0, o1, o2 = k.order(0), k.order(1), k.order(2)
    l0 = comp.create_layer(); l0.insert(k.pixel_path(0, 0)).set_props(k.solid(k.RED)); comp.insert(o0, l0)
    l1 = comp.create_layer(); l1.insert(k.pixel_path(1, 0)).set_props(k.solid(k.RED)); comp.insert(o1, l1)
    l2 = comp.create_layer(); l2.insert(k.pixel_path(2, 0)).insert(k.pixel_path(3, 0)).set_props(k.solid(k.RED)); comp.insert(o2, l2)
    k.render(r, comp, buffer, api.LinearLayout(4, 16, 1), k.GREEN)
    assert buffer.tolist() == RED_SRGB * 4
    assert comp.builder_len() == 16 and comp.actual_len() == 16
    buffer[:] = GREEN_SRGB * 4
    comp.get_mut(o0).clear()
    k.render(r, comp, buffer, api.LinearLayout(4, 16, 1), k.GREEN)
    assert buffer.tolist() == GREEN_SRGB + RED_SRGB * 3
    assert comp.builder_len() == 16 and comp.actual_len() == 12
    buffer[:] = GREEN_SRGB * 4
    comp.get_mut(o2).clear()
    k.render(r, comp, buffer, api.LinearLayout(4, 16, 1), k.GREEN)
    assert buffer.tolist() == GREEN_SRGB + RED_SRGB + GREEN_SRGB * 2
    assert comp.builder_len() == 4 and comp.actual_len() == 4


def test_clear_twice(api, k):                                       # :764-784
    comp = api.Composition()
    order = k.order(0)
    layer = comp.create_layer(); layer.insert(k.pixel_path(0, 0)).set_props(k.solid(k.RED))
    comp.insert(order, layer)
    assert comp.actual_len() == 4
    comp.get_mut(order).clear()
    assert comp.actual_len() == 0
    comp.get_mut(order).clear()
    assert comp.actual_len() == 0


def test_insert_over_layer(api, k):                                 # :786-837
    buffer = k.buf(*[BLACK_SRGB] * 3)
    layout = api.LinearLayout(3, 12, 1)
    comp, r = api.Composition(), api.Renderer()
    layer = comp.create_layer(); layer.insert(k.pixel_path(0, 0)).set_props(k.solid(k.RED))
    comp.insert(k.order(0), layer)
    k.render(r, comp, buffer, layout, k.BLACK)
    assert buffer.tolist() == RED_SRGB + BLACK_SRGB * 2
    layer = comp.create_layer(); layer.insert(k.pixel_path(1, 0)).set_props(k.solid(k.GREEN))
    buffer[:] = BLACK_SRGB * 3
    k.render(r, comp, buffer, layout, k.BLACK)                      # a layer that is not in the composition draws nothing
    assert buffer.tolist() == RED_SRGB + BLACK_SRGB * 2
    comp.insert(k.order(0), layer)
    buffer[:] = BLACK_SRGB * 3
    k.render(r, comp, buffer, layout, k.BLACK)
    assert buffer.tolist() == BLACK_SRGB + GREEN_SRGB + BLACK_SRGB


def test_layer_replace_remove(api, k):                              # :839-892
    buffer = k.buf(*[BLACK_SRGB] * 3)
    layout = api.LinearLayout(3, 12, 1)
    comp, r = api.Composition(), api.Renderer()
    layer = comp.create_layer(); layer.insert(k.pixel_path(0, 0)).set_props(k.solid(k.RED))
    comp.insert(k.order(0), layer)
    k.render(r, comp, buffer, layout, k.BLACK)
    assert buffer.tolist() == RED_SRGB + BLACK_SRGB * 2
    layer = comp.create_layer(); layer.insert(k.pixel_path(1, 0)).set_props(k.solid(k.GREEN))
    _old = comp.insert(k.order(0), layer)
    buffer[:] = BLACK_SRGB * 3
    k.render(r, comp, buffer, layout, k.BLACK)
    assert buffer.tolist() == BLACK_SRGB + GREEN_SRGB + BLACK_SRGB
    _old = comp.remove(k.order(0))
    buffer[:] = BLACK_SRGB * 3
    k.render(r, comp, buffer, layout, k.BLACK)
    assert buffer.tolist() == BLACK_SRGB * 3


def test_layer_clear(api, k):                                       # :894-966
    buffer = k.buf(*[BLACK_SRGB] * 3)
    layout = api.LinearLayout(3, 12, 1)
    comp, r = api.Composition(), api.Renderer()
    order = k.order(0)
    layer = comp.create_layer(); layer.insert(k.pixel_path(0, 0)).set_props(k.solid(k.RED))
    comp.insert(order, layer)
    k.render(r, comp, buffer, layout, k.BLACK)
    assert buffer.tolist() == RED_SRGB + BLACK_SRGB * 2
    comp.get_mut(order).insert(k.pixel_path(1, 0))
    buffer[:] = BLACK_SRGB * 3
    k.render(r, comp, buffer, layout, k.BLACK)
    assert buffer.tolist() == RED_SRGB * 2 + BLACK_SRGB
    comp.get_mut(order).clear()
    buffer[:] = BLACK_SRGB * 3
    k.render(r, comp, buffer, layout, k.BLACK)
    assert buffer.tolist() == BLACK_SRGB * 3
    comp.get_mut(order).insert(k.pixel_path(2, 0))
    buffer[:] = BLACK_SRGB * 3
    k.render(r, comp, buffer, layout, k.BLACK)
    assert buffer.tolist() == BLACK_SRGB * 2 + RED_SRGB


def test_geom_id(api, k):                                           # :968-1000
    comp = api.Composition()
    layer = comp.create_layer()
    layer.insert(api.PathBuilder().build()); g0 = layer.geom_id()
    layer.insert(api.PathBuilder().build()); g1 = layer.geom_id()
    assert g0 == g1
    layer.clear()
    assert layer.geom_id() != g0
    layer.insert(api.PathBuilder().build()); g2 = layer.geom_id()
    assert g0 != g2
    order = k.order(0)
    comp.insert(order, layer)
    assert comp.get_order_if_stored(g2) == order
    comp.insert(order, comp.create_layer())
    assert comp.get_order_if_stored(g2) is None


def test_srgb_alpha_blending(api, k):                               # :1002-1035
    buffer = k.buf(*[BLACK_SRGB] * 3)
    comp, r = api.Composition(), api.Renderer()
    layer = comp.create_layer(); layer.insert(k.pixel_path(0, 0)).set_props(k.solid(k.BLACK_ALPHA_50))
    comp.insert(k.order(0), layer)
    layer = comp.create_layer(); layer.insert(k.pixel_path(1, 0)).set_props(k.solid(k.GRAY))
    comp.insert(k.order(1), layer)
    k.render(r, comp, buffer, api.LinearLayout(3, 12, 1), k.WHITE_TRANSPARENT)
    assert buffer.tolist() == GRAY_ALPHA_50_SRGB + GRAY_SRGB + WHITE_ALPHA_0_SRGB


def test_render_changed_layers_only(api, k):                        # :1037-1105
    W = 3 * TILE_WIDTH
    buffer = k.buf(*[BLACK_SRGB] * (W * TILE_HEIGHT))
    layout = api.LinearLayout(W, W * 4, TILE_HEIGHT)
    comp, r = api.Composition(), api.Renderer()
    cache = r.create_buffer_layer_cache()
    layer = comp.create_layer()
    layer.insert(k.pixel_path(0, 0)).insert(k.pixel_path(TILE_WIDTH, 0)).set_props(k.solid(k.RED))
    comp.insert(k.order(0), layer)
    order = k.order(1)
    layer = comp.create_layer()
    layer.insert(k.pixel_path(TILE_WIDTH + 1, 0)).insert(k.pixel_path(2 * TILE_WIDTH, 0)).set_props(k.solid(k.GREEN))
    comp.insert(order, layer)
    k.render(r, comp, buffer, layout, k.BLACK, cache)
    assert px(buffer, 0) == RED_SRGB and px(buffer, TILE_WIDTH) == RED_SRGB
    assert px(buffer, TILE_WIDTH + 1) == GREEN_SRGB and px(buffer, 2 * TILE_WIDTH) == GREEN_SRGB
    buffer = k.buf(*[BLACK_SRGB] * (W * TILE_HEIGHT))
    comp.get_mut(order).set_props(k.solid(k.RED))
    k.render(r, comp, buffer, layout, k.BLACK, cache)
    assert px(buffer, 0) == BLACK_SRGB                              # the first tile holds only the unchanged layer
    assert px(buffer, TILE_WIDTH) == RED_SRGB and px(buffer, TILE_WIDTH + 1) == RED_SRGB and px(buffer, 2 * TILE_WIDTH) == RED_SRGB


def test_insert_remove_same_order_will_not_render_again(api, k):    # :1107-1149
    buffer = k.buf(*[BLACK_SRGB] * 3)
    layout = api.LinearLayout(3, 12, 1)
    comp, r = api.Composition(), api.Renderer()
    cache = r.create_buffer_layer_cache()
    layer = comp.create_layer(); layer.insert(k.pixel_path(0, 0)).set_props(k.solid(k.RED))
    comp.insert(k.order(0), layer)
    k.render(r, comp, buffer, layout, k.BLACK, cache)
    assert buffer.tolist() == RED_SRGB + BLACK_SRGB * 2
    layer = comp.remove(k.order(0))
    comp.insert(k.order(0), layer)
    buffer[:] = BLACK_SRGB * 3
    k.render(r, comp, buffer, layout, k.BLACK, cache)
    assert buffer.tolist() == BLACK_SRGB * 3


def test_clear_emptied_tiles(api, k):                               # :1151-1228
    W = 2 * TILE_WIDTH
    buffer = k.buf(*[BLACK_SRGB] * (W * TILE_HEIGHT))
    layout = api.LinearLayout(W, W * 4, TILE_HEIGHT)
    comp, r = api.Composition(), api.Renderer()
    cache = r.create_buffer_layer_cache()
    order = k.order(0)
    layer = comp.create_layer()
    layer.insert(k.pixel_path(0, 0)).set_props(k.solid(k.RED)).insert(k.pixel_path(TILE_WIDTH, 0))
    comp.insert(order, layer)
    k.render(r, comp, buffer, layout, k.BLACK, cache)
    assert px(buffer, 0) == RED_SRGB
    comp.get_mut(order).set_transform(k.translate(float(TILE_WIDTH), 0.0))
    k.render(r, comp, buffer, layout, k.BLACK, cache)
    assert px(buffer, 0) == BLACK_SRGB
    comp.get_mut(order).set_transform(k.translate(-float(TILE_WIDTH), 0.0))
    k.render(r, comp, buffer, layout, k.BLACK, cache)
    assert px(buffer, 0) == RED_SRGB
    comp.get_mut(order).set_transform(k.translate(0.0, float(TILE_HEIGHT)))
    k.render(r, comp, buffer, layout, k.BLACK, cache)
    assert px(buffer, 0) == BLACK_SRGB


def test_separate_layer_caches(api, k):                             # :1230-1316
    buffer = k.buf(*[BLACK_SRGB] * (TILE_WIDTH * TILE_HEIGHT))
    layout = api.LinearLayout(TILE_WIDTH, TILE_WIDTH * 4, TILE_HEIGHT)
    comp, r = api.Composition(), api.Renderer()
    cache0, cache1 = r.create_buffer_layer_cache(), r.create_buffer_layer_cache()
    order = k.order(0)
    layer = comp.create_layer(); layer.insert(k.pixel_path(0, 0)).set_props(k.solid(k.RED))
    comp.insert(order, layer)
    k.render(r, comp, buffer, layout, k.BLACK, cache0)
    assert px(buffer, 0) == RED_SRGB
    buffer = k.buf(*[BLACK_SRGB] * (TILE_WIDTH * TILE_HEIGHT))
    k.render(r, comp, buffer, layout, k.BLACK, cache0)
    assert px(buffer, 0) == BLACK_SRGB
    k.render(r, comp, buffer, layout, k.BLACK, cache1)
    assert px(buffer, 0) == RED_SRGB
    comp.get_mut(order).set_transform(k.translate(1.0, 0.0))
    k.render(r, comp, buffer, layout, k.BLACK, cache0)
    assert px(buffer, 0) == BLACK_SRGB and px(buffer, 1) == RED_SRGB
    buffer = k.buf(*[BLACK_SRGB] * (TILE_WIDTH * TILE_HEIGHT))
    k.render(r, comp, buffer, layout, k.BLACK, cache1)
    assert px(buffer, 0) == BLACK_SRGB and px(buffer, 1) == RED_SRGB


def test_draw_if_width_or_height_change(api, k):                    # :1318-1382
    buffer = k.buf(BLACK_SRGB)
    comp, r = api.Composition(), api.Renderer()
    cache = r.create_buffer_layer_cache()
    k.render(r, comp, buffer, api.LinearLayout(1, 4, 1), k.RED, cache)
    assert px(buffer, 0) == RED_SRGB
    buffer = k.buf(BLACK_SRGB)
    k.render(r, comp, buffer, api.LinearLayout(1, 4, 1), k.RED, cache)
    assert px(buffer, 0) == BLACK_SRGB
    buffer = k.buf(*[BLACK_SRGB] * 2)
    k.render(r, comp, buffer, api.LinearLayout(2, 8, 1), k.RED, cache)
    assert buffer.tolist() == RED_SRGB * 2
    buffer = k.buf(*[BLACK_SRGB] * 2)
    k.render(r, comp, buffer, api.LinearLayout(1, 4, 2), k.RED, cache)
    assert buffer.tolist() == RED_SRGB * 2


def test_even_odd(api, k):                                          # :1384-1427
    P = api.Point
    b = api.PathBuilder()
    b.move_to(P(0.0, 0.0)); b.line_to(P(0.0, TILE_HEIGHT)); b.line_to(P(3.0 * TILE_WIDTH, TILE_HEIGHT))
    b.line_to(P(3.0 * TILE_WIDTH, 0.0)); b.line_to(P(TILE_WIDTH, 0.0)); b.line_to(P(TILE_WIDTH, TILE_HEIGHT))
    b.line_to(P(2.0 * TILE_WIDTH, TILE_HEIGHT)); b.line_to(P(2.0 * TILE_WIDTH, 0.0)); b.line_to(P(0.0, 0.0))
    path = b.build()
    W = 3 * TILE_WIDTH
    buffer = k.buf(*[BLACK_SRGB] * (W * TILE_HEIGHT))
    comp, r = api.Composition(), api.Renderer()
    layer = comp.create_layer()
    layer.insert(path).set_props(api.Props(fill_rule=api.FillRule.EvenOdd, func=api.Func.Draw(api.Style(fill=api.Fill.Solid(k.RED)))))
    comp.insert(k.order(0), layer)
    k.render(r, comp, buffer, api.LinearLayout(W, W * 4, TILE_HEIGHT), k.BLACK)
    assert px(buffer, 0) == RED_SRGB and px(buffer, TILE_WIDTH) == BLACK_SRGB and px(buffer, 2 * TILE_WIDTH) == RED_SRGB


def test_buffer_layer_cache_doc_test(api, k):                       # cpu/buffer/mod.rs:125-165
    buffer = np.zeros(4, np.uint8)
    comp, r = api.Composition(), api.Renderer()
    cache = r.create_buffer_layer_cache()
    white = api.Color(1.0, 1.0, 1.0, 1.0)
    k.render(r, comp, buffer, api.LinearLayout(1, 4, 1), white, cache)
    assert buffer.tolist() == [255] * 4                             # Rendered white on first frame.
    buffer[:] = 0
    k.render(r, comp, buffer, api.LinearLayout(1, 4, 1), white, cache)
    assert buffer.tolist() == [0] * 4                               # Skipped rendering on second frame since nothing changed.


def test_at_most_32_layer_caches_and_ids_are_released(api, k):      # renderer.rs:68-73, small_bit_set.rs:52-56, buffer/mod.rs:98-111
    r = api.Renderer()
    caches = [r.create_buffer_layer_cache() for _ in range(32)]
    assert all(c is not None for c in caches) and sorted(c.id for c in caches) == list(range(32))
    assert r.create_buffer_layer_cache() is None
    del caches[5]                                                   # IdDropper: the id returns to the pool
    again = r.create_buffer_layer_cache()
    assert again is not None and again.id == 5


class CountingFlusher:
    def __init__(self):
        self.slices = []

    def flush(self, s):
        self.slices.append(len(s))
        s[:] = 255


def test_flusher_through_the_renderer(api, k):                      # painter/mod.rs `flusher` :1533-1572 at the Renderer level
    width = TILE_WIDTH + TILE_WIDTH // 2
    buffer = np.zeros(width * TILE_HEIGHT * 4, np.uint8)
    comp, r = api.Composition(), api.Renderer()
    fl = CountingFlusher()
    k.render(r, comp, buffer, api.LinearLayout(width, width * 4, TILE_HEIGHT), api.Color(0, 0, 0, 0), flusher=fl)
    assert (buffer == 255).all()
    assert sorted(fl.slices) == sorted([TILE_WIDTH * 4] * TILE_HEIGHT + [TILE_WIDTH * 2] * TILE_HEIGHT)
    # with a cache: an unchanged second frame writes no tile, so nothing is flushed (TileWriteOp::None, :538-539)
    cache = r.create_buffer_layer_cache()
    fl = CountingFlusher()
    k.render(r, comp, buffer, api.LinearLayout(width, width * 4, TILE_HEIGHT), k.BLACK, cache, flusher=fl)
    assert len(fl.slices) == 2 * TILE_HEIGHT
    fl = CountingFlusher()
    k.render(r, comp, buffer, api.LinearLayout(width, width * 4, TILE_HEIGHT), k.BLACK, cache, flusher=fl)
    assert fl.slices == []
