"""The reference's host-side unit tests, replayed literally against the product's mirror of forma's API (`forma_amd.api`):
math/transform.rs:226-300 (GeomPresTransform), utils/order.rs tests (Order limits), styling.rs tests (the bias-shifted half
float of image texels), cpu/buffer/mod.rs `clone_and_drop` (cache ids return to the renderer's pool; GPU: needs a Renderer).
Everything except the last runs without a GPU: these classes are host logic."""
import gc

import numpy as np
import pytest

from forma_amd import api

f32 = np.float32
MAX_SCALING_FACTOR_X = f32(1.0) + f32(1.0 / 16.0) / f32(api.MAX_WIDTH)      # transform.rs:19-20, MAX_ERROR = 1 / 16
MAX_SCALING_FACTOR_Y = f32(1.0) + f32(1.0 / 16.0) / f32(api.MAX_HEIGHT)


# ---- math/transform.rs -----------------------------------------------------------------------------------------------------
def test_default_identity():               # :229-237
    p = api.GeomPresTransform().transform(api.Point(2.0, 3.0))
    assert (p.x, p.y) == (2.0, 3.0)


def test_as_slice():                       # :239-247
    s = [float(f32(v)) for v in (0.1, 0.5, 0.4, 0.3, 0.7, 0.9)]
    assert api.GeomPresTransform.try_from(s).to_array() == s


def test_scale_translate():                # :249-257
    p = api.GeomPresTransform.try_from([0.1, 0.5, 0.4, 0.3, 0.5, 0.6]).transform(api.Point(2.0, 3.0))
    assert (f32(p.x), f32(p.y)) == (f32(2.2), f32(2.3))


@pytest.mark.parametrize("t,x,y", [
    ([0.1, float(np.sqrt(MAX_SCALING_FACTOR_Y)), float(np.sqrt(MAX_SCALING_FACTOR_X)), 0.1, 0.5, 0.0], True, True),   # :259-273
    ([0.1, 0.0, float(np.sqrt(MAX_SCALING_FACTOR_X)), 0.0, 0.5, 0.0], True, False),                                    # :275-283
    ([0.0, float(np.sqrt(MAX_SCALING_FACTOR_Y)), 0.0, 0.1, 0.5, 0.0], False, True),                                    # :285-293
])
def test_wrong_scaling_factor(t, x, y):
    with pytest.raises(api.GeomPresTransformError) as e:
        api.GeomPresTransform.try_from(t)
    assert f"x: {x}" in str(e.value) and f"y: {y}" in str(e.value)


def test_correct_scaling_factor():         # :295-303
    t = [1.0, float(np.sqrt(MAX_SCALING_FACTOR_Y)), 0.0, 0.0, 0.5, 0.0]
    assert api.GeomPresTransform.try_from(t).to_array() == [float(f32(v)) for v in t]


# ---- utils/order.rs --------------------------------------------------------------------------------------------------------
def test_wrong_order_values():             # wrong_u32_order_value, wrong_usize_order_values
    for v in (api.Order.MAX + 1, 2 ** 64 - 1):
        with pytest.raises(api.OrderError):
            api.Order(v)


def test_correct_order_value():
    assert api.Order(api.Order.MAX).as_u32() == api.Order.MAX == api.LAYER_LIMIT == (1 << 21) - 1


# ---- styling.rs: f16 -------------------------------------------------------------------------------------------------------
def _f16_to_f32(h):                        # styling.rs:230-238
    h = np.asarray(h, np.uint32)
    return np.where(h != 0, (np.uint32(0x38000000) + (h << np.uint32(13))).view(np.float32), f32(0.0)).astype(np.float32)


def test_f16_error():                      # styling.rs tests::f16_error
    alpha = np.arange(256, dtype=np.float32) / f32(255.0)
    a16 = api._to_f16(alpha)
    assert float(np.mean((alpha - _f16_to_f32(a16)) ** 2)) < 5e-8
    assert len(set(a16.tolist())) == 256
    comp = api._to_linear(np.arange(256, dtype=np.uint8))
    c16 = api._to_f16(comp)
    assert float(np.mean((comp - _f16_to_f32(c16)) ** 2)) < 3e-8
    assert len(set(c16.tolist())) == 256
    assert _f16_to_f32(api._to_f16(np.array([0.0], np.float32)))[0] == 0.0
    assert _f16_to_f32(api._to_f16(np.array([1.0], np.float32)))[0] == 1.0


def test_f16_conversion():                 # styling.rs tests::f16_conversion (numpy's float16 = IEEE binary16 = the `half` crate)
    v = np.arange(255, dtype=np.float32) / f32(255.0)
    ours = api._to_f16(v)
    ieee = v.astype(np.float16).view(np.uint16)
    assert np.all(np.abs(ieee.astype(np.int32) - ours.astype(np.int32)) <= 1)
    assert np.array_equal(ours.view(np.float16).astype(np.float32), _f16_to_f32(ours))


# ---- cpu/buffer/mod.rs: clone_and_drop --------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_cache_ids_return_to_the_pool():
    r = api.Renderer(device=0)
    c0, c1, c2 = (r.create_buffer_layer_cache() for _ in range(3))
    assert (c0.id, c1.id, c2.id) == (0, 1, 2)
    a0, a1, a2 = c0, c1, c2                # "clones" share the id; dropping one keeps it
    del a0, a1, a2
    gc.collect()
    assert {0, 1, 2} <= r._caches
    del c1
    gc.collect()
    assert 0 in r._caches and 1 not in r._caches and 2 in r._caches
    c1 = r.create_buffer_layer_cache()
    assert c1.id == 1 and {0, 1, 2} <= r._caches          # first empty slot
    del c0, c1, c2
    gc.collect()
    assert not ({0, 1, 2} & r._caches)
