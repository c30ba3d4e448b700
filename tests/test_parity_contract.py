"""Two items of SURVEY.md's parity contract that had no test of their own until round 4.

* Contract 3: "every RGBA8 channel within 1 code value of the oracle run with the optimizer passes both enabled and
  disabled" — the passes of layer_workbench/mod.rs:236-248 are optimisations, a tile painted layer by layer must show the
  same pixels.  `Oracle.set_optimizer(False)` skips them (a switch of the test oracle only).
* The reference's own GPU-vs-CPU stress on sub-tile geometry around the origin, gpu/rasterizer/mod.rs:358-422
  `rasterize_random_quad`: 4096 closed quads with integer coordinates in [-8, 16), one layer each, canvas unbounded
  (`fill_gpu_view(usize::MAX, usize::MAX)`).  The reference compares the *significant* pixel segments of its two backends
  after sorting; here the HIP rasterizer's stream must equal the oracle's segment for segment, IN ORDER, and the sorted
  streams must be bit-identical.  The reference draws its points from `SmallRng::seed_from_u64(0)` (rand's xoshiro256++,
  not reproducible without the crate): the generator here is numpy's PCG64 with the seeds below, documented instead.
"""
import numpy as np
import pytest

import scene as S
from oracle import oracle as orc


def random_quads(seed, n=4096):
    rng = np.random.default_rng(seed)                                  # PCG64(seed): integers in [-8, 16) like Uniform::new(-8i32, 16)
    comp = S.Composition()
    pts = rng.integers(-8, 16, size=(n, 4, 2))
    for i in range(n):
        p = pts[i]
        path = (S.P().move_to(float(p[0, 0]), float(p[0, 1])).line_to(float(p[1, 0]), float(p[1, 1]))
                .line_to(float(p[2, 0]), float(p[2, 1])).line_to(float(p[3, 0]), float(p[3, 1])).build())
        comp.get_mut_or_insert_default(i).insert(path).set_props(S.solid((0.3, 0.5, 0.7, 0.5)))
    return comp


# ---- CPU: the oracle against itself ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [7, 8, 9])
def test_oracle_image_is_the_same_with_the_optimizer_passes_off(seed):
    W, H = 320, 208
    o = orc.Oracle()
    t = S.random_mixed(n=200, width=W, height=H, seed=seed).tables(o)
    S.load(o, t)
    on = o.render(W, H, clear=(0.3, 0.2, 0.1, 1.0))
    o.set_optimizer(False)
    off = o.render(W, H, clear=(0.3, 0.2, 0.1, 1.0))
    assert np.abs(on.astype(int) - off.astype(int)).max() <= 1


def test_oracle_e2e_scenes_with_the_optimizer_passes_off():
    for name, comp in S.e2e_scenes().items():
        o = orc.Oracle()
        t = comp.tables(o)
        S.load(o, t)
        on = o.render(64, 64)
        o.set_optimizer(False)
        assert np.abs(on.astype(int) - o.render(64, 64).astype(int)).max() <= 1, name


def test_oracle_random_quads_stream_is_sorted_and_complete():
    """the restated scene itself: every quad yields segments, tile coordinates stay within [-1, 0], the sort is a permutation"""
    o = orc.Oracle()
    t = random_quads(0).tables(o)
    S.load(o, t)
    o.prepare_lines(65536, 32768)
    u = o.rasterize()
    s = o.sort()
    assert len(u) == len(s) > 4096 and np.array_equal(np.sort(u), np.sort(s))
    k = s >> np.uint64(20)
    assert (k[1:] >= k[:-1]).all()
    ty = (s >> np.uint64(53)).astype(np.int64) - 1
    tx = ((s >> np.uint64(41)) & np.uint64(0xFFF)).astype(np.int64) - 1
    assert set(np.unique(ty)) <= {-1, 0} and set(np.unique(tx)) <= {-1, 0}


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.fixture()
def ctx():
    import forma_amd
    c = forma_amd.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_rasterize_random_quad(ctx, seed):
    """gpu/rasterizer/mod.rs:358-422 restated: line parameters, unsorted stream (in order) and sorted stream bit-identical"""
    o = orc.Oracle()
    t = random_quads(seed).tables(o)
    S.load(o, t); S.load(ctx, t)
    W, H = 65536, 32768                                                # the largest canvas: nothing is culled on the right / below
    lo = o.prepare_lines(W, H); lg = ctx.prepare_lines(W, H)
    for k in lo:
        assert np.array_equal(np.asarray(lo[k]).view(np.uint32), np.asarray(lg[k]).view(np.uint32)), k
    so = o.rasterize()
    assert np.array_equal(so, ctx.rasterize_lines(lo))
    assert np.array_equal(o.sort(), ctx.sort_array(so))
    # and through the frame path on a canvas that holds the visible part (tile 0, 0 and the clamped -1 buckets)
    want = o.render(32, 32)
    got = ctx.render(32, 32)
    assert np.array_equal(ctx.segments(0), o.segments(0)) and np.array_equal(ctx.segments(1), o.segments(1))
    assert np.abs(want.astype(int) - got.astype(int)).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7, 8])
def test_hip_image_within_one_code_value_of_the_oracle_with_passes_on_and_off(ctx, seed):
    W, H = 500, 333
    o = orc.Oracle()
    t = S.random_mixed(n=260, width=W, height=H, seed=seed).tables(o)
    S.load(o, t); S.load(ctx, t)
    clear = (0.3, 0.2, 0.1, 0.6)
    got = ctx.render(W, H, clear=clear).astype(int)
    assert np.abs(got - o.render(W, H, clear=clear).astype(int)).max() <= 1
    o.set_optimizer(False)
    assert np.abs(got - o.render(W, H, clear=clear).astype(int)).max() <= 1
    for name, comp in S.e2e_scenes().items():
        o2 = orc.Oracle()
        t2 = comp.tables(o2)
        S.load(o2, t2); S.load(ctx, t2)
        g = ctx.render(64, 64).astype(int)
        o2.set_optimizer(False)
        assert np.abs(g - o2.render(64, 64).astype(int)).max() <= 1, name


# ---- stage 1 through the C ABI, with tables that do NOT come from the product's host code ---------------------------------
def _random_primitives(seed, n=300):
    rng = np.random.default_rng(seed)
    prim = orc.Primitives()
    p = rng.uniform(0, 400, 2)
    for k in range(n):
        kind = int(rng.integers(0, 4))
        if kind == 0 or k == 0:
            prim.push_contour(); p = rng.uniform(0, 400, 2)
        q = [p] + [p + rng.normal(0, 60, 2) for _ in range(3)]
        if kind == 1:
            prim.push_line(tuple(q[0]), tuple(q[1])); p = q[1]
        elif kind == 2:
            w = float(rng.uniform(0.4, 1.6)) if rng.random() < 0.3 else 1.0
            prim.push_quad(tuple(q[0]), (q[1][0] * w, q[1][1] * w, w), tuple(q[2])); p = q[2]
        else:
            prim.push_cubic(tuple(q[0]), tuple(q[1]), tuple(q[2]), tuple(q[3])); p = q[3]
    return prim


def test_oracle_work_items_reproduce_its_own_flattener():
    """populate_buffers restated on its own (the oracle's flattener fuses it with the map): same number of output points, the
    Start / End commands carry the spline end points the flattener emits"""
    o = orc.Oracle()
    for seed in range(4):
        prim = _random_primitives(seed)
        t = prim.tables()
        x, y, nc = o.flatten_primitives(prim)
        assert t["n_points"] == len(x)
        cmd = t["point_commands"][:t["n_points"]]
        boxed = (cmd & 0x7F800000) == 0x7F800000
        start = boxed & ((cmd & 0x80000000) == 0); end = boxed & ((cmd & 0x80000000) != 0)
        si = (cmd & 0x3FFFFF).astype(np.int64)
        assert np.array_equal(x[start], t["sp0x"][si[start]]) and np.array_equal(y[start], t["sp0y"][si[start]])
        assert np.array_equal(x[end], t["sp2x"][si[end]]) and np.array_equal(y[end], t["sp2y"][si[end]])
        assert np.array_equal(nc[end] != 0, (cmd[end] & 0x400000) != 0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_forma_hip_flatten_on_tables_from_the_oracles_populate_buffers(ctx, seed):
    """the drop-in's fourth hook (rust/forma_hip: PathData::segments hands populate_buffers' ScratchBuffers to
    forma_hip_flatten): k_flatten on work items produced by the ORACLE's restatement of populate_buffers — not by the product's
    host_path.cpp — returns the oracle's points bit for bit"""
    o = orc.Oracle()
    prim = _random_primitives(seed, n=1200)
    t = prim.tables()
    x, y, _ = o.flatten_primitives(prim)
    gx, gy = ctx.flatten_tables(t)
    assert np.array_equal(gx.view(np.uint32), x.view(np.uint32)) and np.array_equal(gy.view(np.uint32), y.view(np.uint32))
