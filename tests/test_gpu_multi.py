"""SURVEY.md §8(e) behind the C ABI: `forma_hip_create_multi` — ONE context over several devices whose `forma_hip_render` is
the whole multi-GPU frame, in both layouts (`forma_hip_multi_layout`): EXCHANGE (line-sharded rasterization -> HIP bucketing by
tile-row owner -> one all-to-all -> band-local sort + paint) and BANDS (no exchange: every device culls the whole scene to its
band of tile rows and renders it like a single device); every device copies its rows into the ONE caller buffer — and frames in
flight inside one context (`forma_hip_set_frames_in_flight`).

This pool hands out single-GPU boxes, so the devices of most tests are the same GPU listed several times: every device still
has its own context, stream, buffers and host thread, the planner / bucket / gather / chunk-mapped sort run exactly as on G
GPUs, and the all-to-all is done with device copies (the library's rehearsal transport).  The RCCL transport is exercised
with a world of one (FORMA_HIP_DEBUG=force_exchange: ncclCommInitAll over one device, grouped ncclAllToAll on the library's own
buffers and stream) and, when two GPUs are visible, for real."""
import os
import subprocess
import sys

import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_for(t):
    o = orc.Oracle()
    S.load(o, t)
    return o


def painted_rows(sorted_full, tiles_h):
    ty = (sorted_full >> np.uint64(53)).astype(np.int64) - 1
    return sorted_full[(ty >= 0) & (ty < tiles_h)]


LAYOUTS = ["exchange", "bands"]


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0] * 8])
def test_multi_device_context_matches_the_oracle(devices, layout):
    import forma_amd
    W, H = 512, 384
    clear = (0.2, 0.3, 0.4, 1.0)
    o = orc.Oracle()
    t = S.random_mixed().tables(o)
    S.load(o, t)
    want = o.render(W, H, clear=clear)
    c = forma_amd.Context(devices=devices, layout=layout)
    S.load(c, t)
    for frame in range(4):                                             # plan + synchronous, then read-back-free
        img = np.full((H, W * 4), 7, np.uint8)
        c.render(W, H, clear=clear, dst=img)
        assert np.array_equal(img, want), frame
        assert np.array_equal(c.segments(1), painted_rows(o.segments(1), (H + 15) // 16)), frame
        assert np.array_equal(c.read_image(W, H), want)
        assert c.tiles_written(W, H).all()
    img, tm = c.render(W, H, clear=clear, timings=True)
    assert np.array_equal(img, want) and tm["n_segments"] > 0 and (tm["exchange_us"] > 0) == (layout == "exchange")
    assert c.info()["layout"] == layout and c.info()["transport"] == ("copy" if layout == "exchange" else "none")
    c.close()


def test_multi_device_e2e_scenes_and_tiny_canvases():
    """the reference's e2e scenes (64 x 64: four tile rows) on 2 and 8 devices — more devices than tile rows leaves bands empty"""
    import forma_amd
    for devices, layout in (([0, 0], "exchange"), ([0] * 8, "exchange"), ([0, 0], "bands"), ([0] * 8, "bands")):
        c = forma_amd.Context(devices=devices, layout=layout)
        for name, comp in S.e2e_scenes().items():
            o = orc.Oracle()
            t = comp.tables(o)
            S.load(o, t); S.load(c, t)
            want = o.render(64, 64)
            for frame in range(2):
                got = c.render(64, 64)
                assert np.abs(want.astype(int) - got.astype(int)).max() <= 1, (name, len(devices), layout, frame)
        c.close()


@pytest.mark.parametrize("layout", LAYOUTS)
def test_multi_device_crop_channels_and_device_resident_frames(layout):
    import forma_amd
    W, H = 500, 333                                                     # odd canvas: partial last tile row and column
    o = orc.Oracle()
    t = S.random_mixed(width=W, height=H, seed=5).tables(o)
    S.load(o, t)
    c = forma_amd.Context(devices=[0, 0, 0], layout=layout)
    S.load(c, t)
    for ch in (S.RGBA, S.BGR1, S.RGB0):
        want = o.render(W, H, channels=ch, clear=(0.1, 0.2, 0.3, 0.5))
        got = c.render(W, H, channels=ch, clear=(0.1, 0.2, 0.3, 0.5))
        assert np.array_equal(got, want), ch
    crop = (40, 300, 50, 250)
    want = o.render(W, H, crop=crop, dst=np.full((H, W * 4), 9, np.uint8))
    got = c.render(W, H, crop=crop, dst=np.full((H, W * 4), 9, np.uint8))
    assert np.array_equal(got, want)
    c.render(W, H, device_only=True)                                    # image stays on the devices, assembled on request
    assert np.array_equal(c.read_image(W, H), o.render(W, H))
    c.close()


@pytest.mark.parametrize("layout", LAYOUTS)
def test_multi_device_replans_when_the_scene_outgrows_its_buckets(layout):
    """a transform that makes the scene grow overflows what the plan provisioned (local segment counts, bucket capacity):
    the frame fails ON THE DEVICE, the context re-plans (new line shares, bands, capacity) and re-runs it — the caller only
    ever sees the right image"""
    import forma_amd
    W, H = 512, 384
    o = orc.Oracle()
    t = S.random_mixed().tables(o)
    S.load(o, t)
    c = forma_amd.Context(devices=[0, 0], layout=layout)        # (BANDS has no buckets to outgrow: the same frames, unbalanced bands)
    S.load(c, t)

    def scaled(k, tx, ty):
        g = t["geoms"].copy()
        g["flags"] = 1
        g["xf"] = np.array([k, 0.0, 0.0, k, tx, ty], np.float32)
        return g

    for k, tx, ty in ((0.4, 150.0, 100.0), (0.95, 10.0, 5.0), (0.95, 14.0, 8.0), (0.3, 10.0, 250.0), (1.0, 0.0, 0.0)):
        g = scaled(k, tx, ty)
        o.set_geoms(g); c.set_geoms(g)
        want = o.render(W, H)
        for _ in range(3):
            assert np.array_equal(c.render(W, H), want), (k, tx, ty)
        assert np.array_equal(c.segments(1), painted_rows(o.segments(1), (H + 15) // 16))
    c.close()


@pytest.mark.parametrize("layout", LAYOUTS)
def test_multi_device_buffer_layer_cache(layout):
    """damage tracking across devices: every device keeps the cache state of its band; an unchanged frame writes nothing,
    a moved layer rewrites the tiles the oracle-backed reference rewrites"""
    import forma_amd
    W, H = 256, 256
    comp = S.Composition()
    comp.get_mut_or_insert_default(0).insert(S.custom_square(0, 0, W, H)).set_props(S.solid((0.9, 0.9, 0.8, 1.0)))
    comp.get_mut_or_insert_default(1).insert(S.custom_circle(80, 90, 40)).set_props(S.solid((0.8, 0.1, 0.1, 1.0)))
    comp.get_mut_or_insert_default(2).insert(S.custom_circle(170, 180, 30)).set_props(S.solid((0.1, 0.2, 0.9, 0.6)))
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(o, t)
    c = forma_amd.Context(devices=[0, 0, 0], layout=layout)
    S.load(c, t)
    n_orders = len(t["style_offsets"])

    def frame(unchanged):
        """both backends render into sentinel-filled buffers: a tile the optimizer skips (TileWriteOp::None) keeps the sentinel"""
        u = np.asarray(unchanged, np.uint8)
        o.set_styles(t["style_offsets"], t["style_words"], u); c.set_styles(t["style_offsets"], t["style_words"], u)
        so, sc = np.full((H, W * 4), 201, np.uint8), np.full((H, W * 4), 201, np.uint8)
        o.render(W, H, cache_id=0, dst=so)
        c.render(W, H, cache_id=0, dst=sc)
        assert np.array_equal(so, sc)
        tiles = (sc != 201).reshape(H // 16, 16, W // 16, 16, 4).any(axis=(1, 3, 4))
        assert np.array_equal(tiles.reshape(-1), c.tiles_written(W, H) != 0)
        return tiles

    assert frame([0] * n_orders).all()
    assert not frame([1] * n_orders).any()                              # nothing changed: nothing written, on any device
    g = t["geoms"].copy()                                               # layer 2 moves
    g[2]["flags"] = 1
    g[2]["xf"] = np.array([1, 0, 0, 1, -30.0, -20.0], np.float32)
    o.set_geoms(g); c.set_geoms(g)
    w = frame([1, 1, 0])
    assert w.any() and not w.all()
    assert not frame([1, 1, 1]).any()
    c.close()


def test_multi_layout_auto_and_switching_on_a_live_context():
    """FORMA_LAYOUT_AUTO (the default) decides per plan — BANDS for every scene with more pixel segments than lines; the layout
    can be changed on a live context with frames in flight: in flight frames are settled, the next frame plans anew, every image
    and sorted stream stays the oracle's"""
    import forma_amd
    from forma_amd import FormaError
    W, H = 512, 384
    o = orc.Oracle()
    t = S.random_mixed().tables(o)
    S.load(o, t)
    want = o.render(W, H)
    c = forma_amd.Context(devices=[0, 0, 0], frames_in_flight=2)
    S.load(c, t)
    assert np.array_equal(c.render(W, H), want)
    assert c.info()["layout"] == "bands" and c.info()["transport"] == "none"
    for layout in ("exchange", "bands", "exchange", "auto", "exchange"):
        for _ in range(3):
            c.render(W, H, device_only=True)                           # (frames in flight under the old layout)
        c.set_layout(layout)
        for _ in range(5):
            c.render(W, H, device_only=True)
        assert np.array_equal(c.read_image(W, H), want), layout
        assert np.array_equal(c.segments(1), painted_rows(o.segments(1), (H + 15) // 16)), layout
        assert c.info()["layout"] == ("bands" if layout == "auto" else layout)
        assert np.array_equal(c.render(W, H), want), layout
    with pytest.raises(FormaError):
        c.set_layout(7)
    c.close()
    single = forma_amd.Context(0)
    with pytest.raises(FormaError) as e:
        single.set_layout("bands")
    assert e.value.code == -5
    single.close()


def test_multi_device_context_refuses_the_single_device_plumbing():
    import forma_amd
    from forma_amd import FormaError
    c = forma_amd.Context(devices=[0, 0])
    for call in (lambda: c.set_band(0, 2), lambda: c.rasterize_frame(64, 64), lambda: c.stream_handle(), lambda: c.segments(0)):
        with pytest.raises(FormaError) as e:
            call()
        assert e.value.code == -5
    with pytest.raises(FormaError):
        forma_amd.Context(devices=[0, 99])
    with pytest.raises(FormaError):
        forma_amd.Context(devices=[0] * 9)
    c.close()


def _full_size(workload):
    from forma_amd import api, scenes
    fn, W, H = scenes.WORKLOADS[workload]
    r = api.Renderer(0)
    r.render(fn(), api.BufferBuilder(np.zeros(W * H * 4, np.uint8), api.LinearLayout(W, W * 4, H)).build(), api.RGBA,
             api.Color(1, 1, 1, 1), None)
    t = dict(r.host_tables)
    r._ctx.close()
    o = oracle_for(t)
    want = o.render(W, H, clear=(1.0, 1.0, 1.0, 1.0))
    return t, W, H, want, o.segments(1)


@pytest.mark.parametrize("workload,G,layout", [("triangles-10m-8k", 8, "exchange"), ("paris-like-30k-4k", 4, "exchange"),
                                               ("triangles-10m-8k", 8, "bands"), ("paris-like-30k-4k", 8, "bands")])
def test_multi_device_full_size(workload, G, layout):
    """BASELINE configs 3 (stand-in) and 4 through forma_hip_render on a multi-device context: image within one code value,
    sorted stream of the painted rows bit-identical, on the planning frame and on the read-back-free frames after it"""
    import forma_amd
    t, W, H, want, sorted_full = _full_size(workload)
    c = forma_amd.Context(devices=[0] * G, layout=layout)
    S.load(c, t)
    img = np.zeros((H, W * 4), np.uint8)
    for frame in range(3):
        img[:] = 0
        c.render(W, H, clear=(1.0, 1.0, 1.0, 1.0), dst=img)
        d = np.abs(img.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1, (frame, int(d.max()))
        assert np.array_equal(c.segments(1), painted_rows(sorted_full, (H + 15) // 16)), frame
    c.close()


def test_product_api_renderer_over_several_devices():
    """`Renderer::with_devices` through the product API: the reference's composition flow (compose -> render into a caller
    buffer) on a multi-device renderer"""
    from forma_amd import api, scenes
    comp = scenes.random_cubics(200, 640, 480)
    r1, rm = api.Renderer(0), api.Renderer(devices=[0, 0, 0, 0])
    W, H = 640, 480
    a, b = np.zeros(W * H * 4, np.uint8), np.zeros(W * H * 4, np.uint8)
    lay = api.LinearLayout(W, W * 4, H)
    for _ in range(3):
        r1.render(comp, api.BufferBuilder(a, lay).build(), api.BGRA, api.Color(0.9, 0.9, 0.9, 1.0), None)
        rm.render(comp, api.BufferBuilder(b, lay).build(), api.BGRA, api.Color(0.9, 0.9, 0.9, 1.0), None)
        assert np.array_equal(a, b)
    o = oracle_for(rm.host_tables)
    assert np.array_equal(o.render(W, H, channels=S.BGRA, clear=(0.9, 0.9, 0.9, 1.0)).reshape(-1), b)


_RCCL_WORLD1 = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch                                    # (first: one HIP runtime in the process, tests/conftest.py)
import scene as S
from oracle import oracle as orc
import forma_amd
o = orc.Oracle()
t = S.random_mixed().tables(o)
S.load(o, t)
W, H = 512, 384
want = o.render(W, H)
c = forma_amd.Context(devices=DEVICES)
S.load(c, t)
for frame in range(4):
    img = np.full((H, W * 4), 7, np.uint8)
    c.render(W, H, dst=img)
    assert np.array_equal(img, want), frame
ty = (o.segments(1) >> np.uint64(53)).astype(np.int64) - 1
assert np.array_equal(c.segments(1), o.segments(1)[(ty >= 0) & (ty < (H + 15) // 16)])
c.close()
print("RCCL-OK")
'''


def _run_rccl_script(devices, env_extra):
    env = dict(os.environ)
    env.pop("FORMA_HIP_DEBUG", None)                       # (no xchg=copy from the outside: this is the RCCL path)
    env.update(env_extra)
    code = _RCCL_WORLD1.replace("ROOT", repr(ROOT)).replace("DEVICES", repr(devices))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "RCCL-OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


@pytest.mark.timeout(300)
def test_rccl_inside_the_library_with_a_world_of_one():
    """everything the multi-device frame asks of RCCL, inside libforma_hip.so, on one GPU: librccl found with dlopen,
    ncclCommInitAll over [0], grouped ncclAllToAll of the counts and of the padded buckets on the context's stream with
    the library's own buffers, the received bucket sorted and painted (FORMA_HIP_DEBUG=force_exchange)"""
    _run_rccl_script([0], {"FORMA_HIP_DEBUG": "force_exchange"})


@pytest.mark.timeout(300)
def test_two_devices_over_rccl_in_one_process():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_rccl_script([0, 1], {})


# ---- frames in flight inside ONE context ------------------------------------------------------------------------------------------
def test_frames_in_flight_in_one_context():
    import forma_amd
    W, H = 512, 384
    o = orc.Oracle()
    t = S.random_mixed().tables(o)
    S.load(o, t)
    c = forma_amd.Context(0, frames_in_flight=3)
    S.load(c, t)
    clears = [(1, 1, 1, 1), (0.2, 0.3, 0.4, 1.0), (0, 0, 0, 0)]
    for k in range(14):                                                # enqueue, enqueue, ...: results are read when asked for
        clear = clears[k % 3]
        assert c.render(W, H, clear=clear, device_only=True) is None
        if k % 5 == 4:
            assert np.array_equal(c.read_image(W, H), o.render(W, H, clear=clear)), k   # the MOST RECENT frame
            assert np.array_equal(c.segments(1), o.segments(1))
    c.sync()
    # a frame into caller memory keeps the reference's contract and sees everything enqueued before it
    img = c.render(W, H, clear=clears[1], dst=np.zeros((H, W * 4), np.uint8))
    assert np.array_equal(img, o.render(W, H, clear=clears[1]))
    # scene changes wait for the frames in flight, then every slot sees the new scene
    g = t["geoms"].copy()
    g["flags"] = 1
    g["xf"] = np.array([0.8, 0.0, 0.0, 0.8, 30.0, 20.0], np.float32)
    for k in range(4):
        c.render(W, H, device_only=True)
    o.set_geoms(g); c.set_geoms(g)
    for k in range(7):
        c.render(W, H, device_only=True)
    assert np.array_equal(c.read_image(W, H), o.render(W, H))
    c.set_frames_in_flight(1)
    assert np.array_equal(c.render(W, H), o.render(W, H))
    c.close()


def test_frames_in_flight_reports_a_deferred_error():
    """a frame that fails on the device after the call returned (a tile deeper than the painter's list) surfaces at the call
    that completes it"""
    import forma_amd
    from forma_amd import FormaError
    comp = S.Composition()
    n = 4200
    for i in range(n):
        comp.get_mut_or_insert_default(i).insert(S.custom_square(2, 2, 12, 12)).set_props(S.solid((0.5, 0.5, 0.5, 0.5)))
    o = orc.Oracle()
    t = comp.tables(o)
    c = forma_amd.Context(0, frames_in_flight=2)
    S.load(c, t)
    try:
        for _ in range(4):
            c.render(32, 32, device_only=True)
        c.sync()
        deep_ok = True
    except FormaError as e:
        deep_ok = False
        assert e.code == -4
    if deep_ok:                                                        # (builds without the layer cap render it)
        S.load(o, t)
        assert np.abs(c.read_image(32, 32).astype(int) - o.render(32, 32).astype(int)).max() <= 1
    c.close()


# ---- frames in flight on a MULTI-DEVICE context (round 4): F frame slots x G devices, one set of communicators per slot --------
@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("devices,slots", [([0, 0], 2), ([0, 0], 3), ([0] * 8, 2), ([0] * 8, 3)])
def test_multi_device_frames_in_flight(devices, slots, layout):
    """device-resident, cache-less frames are ENQUEUED on the next frame slot of every device (buckets, the slot's exchange,
    the owner's sort + paint, all stream-ordered) and verified when the slot comes round; image and sorted stream of the most
    recent frame equal the oracle's; frames into caller memory, scene changes and `sync` see everything enqueued before them"""
    import forma_amd
    W, H = 512, 384
    o = orc.Oracle()
    t = S.random_mixed().tables(o)
    S.load(o, t)
    c = forma_amd.Context(devices=devices, frames_in_flight=slots, layout=layout)
    info = c.info()
    assert info["n_devices"] == len(devices) and info["frames_in_flight"] == slots
    assert info["transport"] == ("copy" if layout == "exchange" else "none") and info["layout"] == layout
    S.load(c, t)
    clears = [(1, 1, 1, 1), (0.2, 0.3, 0.4, 1.0), (0, 0, 0, 0)]
    for k in range(17):
        clear = clears[k % 3]
        assert c.render(W, H, clear=clear, device_only=True) is None
        if k % 5 == 4 or k < 2:
            assert np.array_equal(c.read_image(W, H), o.render(W, H, clear=clear)), k          # the MOST RECENT frame
            assert np.array_equal(c.segments(1), painted_rows(o.segments(1), (H + 15) // 16)), k
            assert c.tiles_written(W, H).all()
    c.sync()
    img = c.render(W, H, clear=clears[1], dst=np.zeros((H, W * 4), np.uint8))                  # synchronous contract kept
    assert np.array_equal(img, o.render(W, H, clear=clears[1]))
    # a scene change waits for the frames in flight; every slot of every device then sees the new scene
    g = t["geoms"].copy()
    g["flags"] = 1
    g["xf"] = np.array([0.8, 0.0, 0.0, 0.8, 30.0, 20.0], np.float32)
    for k in range(slots + 1):
        c.render(W, H, device_only=True)
    o.set_geoms(g); c.set_geoms(g)
    for k in range(2 * slots + 1):
        c.render(W, H, device_only=True)
    assert np.array_equal(c.read_image(W, H), o.render(W, H))
    assert np.array_equal(c.segments(1), painted_rows(o.segments(1), (H + 15) // 16))
    # a crop, and another canvas size (a new plan while frames are in flight)
    crop = (40, 300, 50, 250)
    for k in range(slots + 1):
        c.render(W, H, crop=crop, device_only=True)
    want = o.render(W, H, crop=crop, dst=np.zeros((H, W * 4), np.uint8))
    got = c.read_image(W, H)
    y0, y1, x0, x1 = crop[2] // 16 * 16, min(H, (crop[3] + 15) // 16 * 16), crop[0] // 16 * 16, min(W, (crop[1] + 15) // 16 * 16)
    assert np.array_equal(got.reshape(H, W, 4)[y0:y1, x0:x1], want.reshape(H, W, 4)[y0:y1, x0:x1])
    for k in range(slots + 2):
        c.render(300, 200, device_only=True)
    assert np.array_equal(c.read_image(300, 200), o.render(300, 200))
    c.set_frames_in_flight(1)
    assert np.array_equal(c.render(W, H), o.render(W, H))
    c.close()


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("slots", [2, 3])
def test_multi_device_frames_in_flight_replan_while_frames_are_deferred(slots, layout):
    """the scene grows while frames are in flight: deferred frames whose buckets outgrew the plan are void on every device; the
    context settles every slot, re-plans and runs them again — the caller only ever reads the right image"""
    import forma_amd
    W, H = 512, 384
    o = orc.Oracle()
    t = S.random_mixed().tables(o)
    S.load(o, t)
    c = forma_amd.Context(devices=[0, 0, 0], frames_in_flight=slots, layout=layout)
    S.load(c, t)

    def scaled(k, tx, ty):
        g = t["geoms"].copy()
        g["flags"] = 1
        g["xf"] = np.array([k, 0.0, 0.0, k, tx, ty], np.float32)
        return g

    for k, tx, ty in ((0.4, 150.0, 100.0), (0.95, 10.0, 5.0), (0.3, 10.0, 250.0), (1.0, 0.0, 0.0), (0.5, 200.0, 0.0)):
        g = scaled(k, tx, ty)
        o.set_geoms(g); c.set_geoms(g)
        for _ in range(2 * slots + 1):
            c.render(W, H, device_only=True)
        assert np.array_equal(c.read_image(W, H), o.render(W, H)), (k, tx, ty)
        assert np.array_equal(c.segments(1), painted_rows(o.segments(1), (H + 15) // 16))
    c.close()


@pytest.mark.parametrize("layout", LAYOUTS)
def test_multi_device_frames_in_flight_report_a_deferred_error(layout):
    """a deferred frame that fails on a device after the call returned surfaces at the call that completes it, and the context
    keeps working"""
    import forma_amd
    from forma_amd import FormaError
    W, H = 256, 128
    o = orc.Oracle()
    t = S.random_mixed(width=W, height=H, seed=3).tables(o)
    S.load(o, t)
    c = forma_amd.Context(devices=[0, 0], frames_in_flight=2, layout=layout)
    S.load(c, t)
    for _ in range(4):
        c.render(W, H, device_only=True)
    c.sync()
    bad = t["style_offsets"].copy()
    bad[:] = 0xFFFFFFFF                                                # every order loses its style: k_carry_rows reports it (FORMA_E_ARG)
    c.set_styles(bad, t["style_words"], t["unchanged"])
    with pytest.raises(FormaError):
        for _ in range(6):
            c.render(W, H, device_only=True)
        c.sync()
    try:
        c.sync()                                                       # (the other slots' frames failed the same way: flush them)
    except FormaError:
        pass
    c.set_styles(t["style_offsets"], t["style_words"], t["unchanged"])
    for _ in range(5):
        c.render(W, H, device_only=True)
    assert np.array_equal(c.read_image(W, H), o.render(W, H))
    c.close()


def _device_free_bytes():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_trim_gives_per_frame_memory_back(devices):
    """forma_hip_trim: per-frame device memory (grown to the largest frame seen) is released, the scene and the buffer-layer
    caches stay; the next frames — synchronous first, then read-back-free, in flight, with the cache — are the oracle's."""
    import forma_amd
    from forma_amd._lib import FormaError
    W, H = 2048, 1536
    comp = S.random_mixed(n=1500, width=W, height=H, seed=77)
    o = orc.Oracle()
    t = comp.tables(o)
    S.load(o, t)
    want = o.render(W, H, clear=(0.1, 0.2, 0.3, 1.0))
    c = forma_amd.Context(0, devices=devices) if devices else forma_amd.Context(0, frames_in_flight=3)
    S.load(c, t)
    for k in range(6):                                                 # (the HIP runtime's own first-launch allocations — code objects,
        c.render(48, 48, device_only=True)                             #  scratch, per-stream pools — are not ours to give back)
    c.render(48, 48, cache_id=3, dst=np.zeros((48, 48 * 4), np.uint8))
    c.trim()
    base = _device_free_bytes()
    bo, bc = np.zeros((H, W * 4), np.uint8), np.zeros((H, W * 4), np.uint8)
    for k in range(5):
        c.render(W, H, clear=(0.1, 0.2, 0.3, 1.0), device_only=True)
    assert np.array_equal(c.read_image(W, H), want)
    for k in range(2):
        o.render(W, H, clear=(0.1, 0.2, 0.3, 1.0), cache_id=3, dst=bo); c.render(W, H, clear=(0.1, 0.2, 0.3, 1.0), cache_id=3, dst=bc)
    assert np.array_equal(bo, bc)
    grown = _device_free_bytes()
    c.trim()
    trimmed = _device_free_bytes()
    assert base - grown > 64 << 20, (base, grown)                      # the frames did allocate
    assert trimmed - grown > 0.4 * (base - grown), (base, grown, trimmed)   # ... and gave it back (what stays: the cache's image and
                                                                            # tiles, and whatever the HIP runtime grew for itself)
    if not devices:
        with pytest.raises(FormaError):
            c.read_image(W, H)                                         # no image on the device until the next render
    c.trim()                                                           # (idempotent)
    # the cache survived: with every layer flagged unchanged no tile is written, the caller's buffer keeps what it holds
    unch = np.ones_like(t["unchanged"])
    o.set_styles(t["style_offsets"], t["style_words"], unch); c.set_styles(t["style_offsets"], t["style_words"], unch)
    bo2, bc2 = np.full_like(bo, 201), np.full_like(bc, 201)
    o.render(W, H, clear=(0.1, 0.2, 0.3, 1.0), cache_id=3, dst=bo2); c.render(W, H, clear=(0.1, 0.2, 0.3, 1.0), cache_id=3, dst=bc2)
    assert np.array_equal(bo2, bc2) and (bc2 == 201).all()
    assert int(np.count_nonzero(c.tiles_written(W, H))) == 0
    o.set_styles(t["style_offsets"], t["style_words"], t["unchanged"]); c.set_styles(t["style_offsets"], t["style_words"], t["unchanged"])
    for k in range(4):
        c.render(W, H, clear=(0.1, 0.2, 0.3, 1.0), device_only=True)
    assert np.array_equal(c.read_image(W, H), want)
    c.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("FORMA_TEST_FUZZ_SEEDS", "6")))))
def test_multi_device_random_scenes_crops_and_caches(seed):
    """ONE context over 2-5 emulated devices under the same randomised regime as the single-device cache tests: all-features
    scenes on canvases off the tile grid, random crops and channel orders, layers moving / toggling between frames, a
    buffer-layer cache — plain frames equal the oracle's image, cache frames the buffer it carries AND the set of tiles it
    rewrote (the bands of the devices stitch the damage set of the whole canvas)."""
    import forma_amd
    rng = np.random.default_rng(52000 + seed)
    G = int(rng.integers(2, 6))
    w, h = int(rng.integers(60, 800)), int(rng.integers(60, 520))
    comp = S.random_mixed(n=int(rng.integers(5, 200)), width=w, height=h, seed=53000 + seed)
    orders = sorted(comp.layers.keys())
    crop = None
    if seed % 3 == 2:
        x0, y0 = int(rng.integers(0, w // 2)), int(rng.integers(0, h // 2))
        crop = (x0, int(rng.integers(x0 + 1, w + 1)), y0, int(rng.integers(y0 + 1, h + 1)))
    ch = [(0, 1, 2, 3), (2, 1, 0, 3), (3, 2, 1, 0)][seed % 3]
    clear = tuple(float(v) for v in rng.random(4))
    o = orc.Oracle(); c = forma_amd.Context(0, devices=[0] * G)
    try:
        for frame_no in range(6):
            if frame_no:
                k = int(rng.integers(0, max(2, len(orders) // 5)))
                moved = set(int(v) for v in rng.choice(orders, size=min(k, len(orders)), replace=False))
                for order, layer in comp.layers.items():
                    layer.unchanged = order not in moved
                for m in moved:
                    if rng.random() < 0.7:
                        comp.layers[m].set_transform([1.0, 0.0, 0.0, 1.0, float(rng.uniform(-60, 60)), float(rng.uniform(-40, 40))])
                    else:
                        comp.layers[m].enabled = not comp.layers[m].enabled
            t = comp.tables(o)
            S.load(o, t)
            if frame_no == 0:
                S.load(c, t)
            else:
                c.set_geoms(t["geoms"]); c.set_styles(t["style_offsets"], t["style_words"], t["unchanged"])
            want = o.render(w, h, clear=clear, crop=crop, channels=ch).reshape(h, w, 4).astype(int)
            got = c.render(w, h, clear=clear, crop=crop, channels=ch).reshape(h, w, 4).astype(int)
            m = np.ones((h, w), bool)
            if crop is not None:
                m[:] = False
                m[crop[2] // 16 * 16: min(h, (crop[3] + 15) // 16 * 16), crop[0] // 16 * 16: min(w, (crop[1] + 15) // 16 * 16)] = True
            assert np.abs(want - got)[m].max(initial=0) <= 1, (seed, G, frame_no, "plain")
            sent = [np.full((h, w * 4), 201, np.uint8), np.full((h, w * 4), 201, np.uint8)]
            o.render(w, h, clear=clear, crop=crop, channels=ch, cache_id=4, dst=sent[0])
            c.render(w, h, clear=clear, crop=crop, channels=ch, cache_id=4, dst=sent[1])
            assert np.abs(sent[0].astype(int) - sent[1].astype(int)).max() <= 1, (seed, G, frame_no, "cache")
            wr = (sent[1].reshape(h, w, 4) != 201).any(axis=2)
            tw, th = (w + 15) // 16, (h + 15) // 16
            tiles = np.array([[wr[ty * 16: ty * 16 + 16, tx * 16: tx * 16 + 16].any() for tx in range(tw)] for ty in range(th)])
            assert np.array_equal(tiles.reshape(-1), c.tiles_written(w, h) != 0), (seed, G, frame_no)
    finally:
        c.close()
