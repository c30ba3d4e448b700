"""Every result-neutral schedule of the library, forced BOTH ways, on the driver's box (VERDICT r5 "What's weak" #2: the
driver's `-m gpu` run exercised the default policies only; the non-default variants had one forcing test each on one scene
family, and "the whole suite under each switch" existed only as a builder-side log).

`FORMA_HIP_DEBUG` (csrc/debug.h) is parsed when a context is created, so each case below creates its own context under one
switch string and renders three mid-size scene families — a mixed scene (gradients, textures, blend modes, clips: the general
painter), opaque cubics (the all-solid painter, occlusion culling, deep tiles) and translucent cubics (every carry visible) —
as a synchronous frame, read-back-free frames with one frame in flight and frames on three frame slots, full canvas and a crop
that starts and ends inside tiles.  Every image is compared with the oracle's (<= 1 code value; the u64 streams bit-exact).
The oracle images are computed once per module."""
import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

W, H = 1280, 720
CLEAR = (0.9, 0.95, 1.0, 1.0)
CROP = (16 * 3 + 5, W - 37, 16 * 2 + 9, H - 21)

# one switch string per case: each switch of csrc/debug.h that selects a schedule or a kernel variant, at every value it takes
SWITCHES = [
    "",                                            # the policies themselves
    "strip_tiles=100000000", "strip_tiles=0",      # k_paint_wave<.., NPX = 1> everywhere / never
    "paint_quad=2", "paint_quad=0",                # k_paint_quad for every all-solid scene / never
    "no_order", "order_thr=1",                     # heavy-first order never / every tile filed as heavy
    "runs_chain=1", "runs_chain=0",                # k_runs_wave<1> alone / k_runs_count + k_runs_wave<0>
    "runs_blk=1,runs_chain=0", "runs_blk=0,runs_chain=0", "runs_blk=1,runs_chain=0,carry_slices=1",   # k_runs_wave<2> (runs numbered per tile of the run kernel) / never
    "sort_cus=0", "sort_cus=64", "sort_cus=128",   # persistent workgroups of a digit pass
    "carry_half=0", "carry_half=2", "carry_half=4", "carry_covl=0", "carry_covl=0,carry_slices=1",
    "carry_slices=1", "carry_slices=3", "carry_slices=8", "no_small_carry",
    "force_cull", "no_cull",
    "digit_bits=4", "digit_bits=8", "digit_bits=9", "no_bias", "no_ras_hist", "no_prezero",
    "global_runsort", "span_groups", "no_span_groups",
    "sync",                                        # no read-back-free frames at all
    "tail_poll=0",                                 # the host waits for the stream instead of polling k_frame_tail's word
    "paint_split=3", "paint_split=8", "paint_split=0",   # frames into caller memory: the painter in bands whose copies leave early / never
    "paint_split=5,strip_tiles=100000000", "paint_split=4,paint_quad=2", "paint_split=2,poison_frame=255,force_cull",
    "paint_split=2,split_first=10", "paint_split=2,split_first=70",   # the policy's two bands (first band: split_first percent of the rows)
    "no_simple_paint",                             # (process-wide, read once: effective only if this is the process's first context)
    # combinations that meet in the bench configurations
    "strip_tiles=100000000,force_cull,runs_chain=1", "paint_quad=2,runs_chain=1,carry_half=2", "order_thr=1,force_cull,sort_cus=128",
    "poison_frame=255,runs_chain=1,strip_tiles=100000000", "poison_frame=0,order_thr=1,paint_quad=2",
    "poison_frame=255,runs_blk=1,runs_chain=0,carry_slices=1,force_cull", "poison_frame=0,runs_blk=1,runs_chain=0,blk_round=2,strip_tiles=100000000",
]


@pytest.fixture(scope="module")
def cases():
    """[(name, tables, {crop: oracle image}, unsorted stream, sorted stream)]"""
    out = []
    for name, comp in (("mixed", S.random_mixed(n=450, width=W, height=H, seed=61)),
                       ("opaque-cubics", S.random_cubics(n=500, width=W, height=H, seed=62)),
                       ("translucent-cubics", S.random_cubics(n=260, width=W, height=H, seed=63, alpha=0.55))):
        o = orc.Oracle()
        t = comp.tables(o)
        S.load(o, t)
        imgs = {None: o.render(W, H, clear=CLEAR), CROP: o.render(W, H, clear=CLEAR, crop=CROP)}
        out.append((name, t, imgs, o.segments(0).copy(), o.segments(1).copy()))
    return out


def _same(img, ref, crop, what):
    y0, y1, x0, x1 = (crop[2], crop[3], crop[0], crop[1]) if crop else (0, H, 0, W)
    a = img.reshape(H, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16)
    b = ref.reshape(H, -1)[y0:y1, 4 * x0:4 * x1].astype(np.int16)
    d = np.abs(a - b)
    assert d.max() <= 1, (what, int(d.max()), int((d > 1).sum()))


@pytest.mark.parametrize("switch", SWITCHES)
def test_every_schedule_forced_both_ways_paints_the_oracles_image(monkeypatch, cases, switch):
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", switch)
    c = forma_amd.Context(0)
    try:
        for name, t, imgs, unsorted, sorted_ in cases:
            S.load(c, t)
            for crop in (None, CROP):
                for k in range(4):                           # frame 0: synchronous; 1..3: read-back-free (order lists, culling, chain)
                    img = c.render(W, H, clear=CLEAR, crop=crop)
                    _same(img, imgs[crop], crop, (switch, name, crop, "frame %d" % k))
                if crop is None:
                    assert np.array_equal(c.segments(0), unsorted), (switch, name, "unsorted stream")
                    assert np.array_equal(c.segments(1), sorted_), (switch, name, "sorted stream")
            # three frame slots: frames enqueued round-robin, every slot's frames checked (the last three frames = one per slot)
            c.set_frames_in_flight(3)
            for k in range(9):
                c.render(W, H, clear=CLEAR, device_only=True)
                if k >= 6:
                    _same(c.read_image(W, H), imgs[None], None, (switch, name, "slot frame %d" % k))
            assert np.array_equal(c.segments(1), sorted_), (switch, name, "sorted stream, three slots")
            c.set_frames_in_flight(1)
    finally:
        c.close()


@pytest.mark.parametrize("slots", [2, 4])
def test_two_and_four_frame_slots(cases, slots):
    """The slot counts between and beyond the recommended three (the digit passes' CU share follows the slot count)."""
    import forma_amd
    c = forma_amd.Context(0)
    try:
        name, t, imgs, _u, sorted_ = cases[0]
        S.load(c, t)
        c.set_frames_in_flight(slots)
        for k in range(4 * slots):
            c.render(W, H, clear=CLEAR, device_only=True)
            if k >= 3 * slots:
                _same(c.read_image(W, H), imgs[None], None, (slots, name, k))
        assert np.array_equal(c.segments(1), sorted_)
    finally:
        c.close()


@pytest.mark.parametrize("split", ["paint_split=0", "paint_split=2", "paint_split=3", "paint_split=8"])
def test_split_frames_into_a_padded_buffer_leave_the_rest_alone(monkeypatch, cases, split):
    """A frame into caller memory whose painter runs in bands (api.cpp split_plan / send_split_bands: every band's copy leaves behind
    its launch's event on a second stream): pitched destination, a crop inside tiles — what lies outside the crop's tiles and
    in the padding keeps the caller's bytes, what lies inside equals the one-launch frame's."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", split)
    name, t, imgs, _, _ = cases[0]
    stride = W * 4 + 192
    c = forma_amd.Context(0)
    try:
        S.load(c, t)
        for crop in (None, CROP):
            for k in range(4):
                dst = np.full((H, stride), 0xA5, np.uint8)
                out = c.render(W, H, clear=CLEAR, crop=crop, dst=dst, stride=stride)
                got = np.asarray(out).reshape(H, stride)
                assert (got[:, W * 4:] == 0xA5).all(), (split, crop, k, "padding overwritten")
                _same(got[:, :W * 4].reshape(H, W, 4), imgs[crop], crop, (split, name, crop, "padded frame %d" % k))
                if crop is not None:
                    x0, x1, y0, y1 = crop
                    ty0, ty1 = (y0 // 16) * 16, min(-(-y1 // 16) * 16, H)
                    assert (got[:ty0] == 0xA5).all() and (got[ty1:] == 0xA5).all(), (split, k, "rows outside the crop's tiles overwritten")
    finally:
        c.close()


@pytest.mark.parametrize("split", ["paint_split=3", "paint_split=2,split_first=40"])
def test_a_split_frame_whose_prediction_fails_is_rendered_again(monkeypatch, split):
    """The bands of a split frame leave before the frame is verified (send_split_bands): when frame k + 1 has far more pixel
    segments than frame k predicted, the copies on their way are waited for, the frame runs again synchronously and `dst`
    holds the right image when the call returns (complete_async_frame: settle_split before FORMA_RETRY)."""
    import forma_amd
    monkeypatch.setenv("FORMA_HIP_DEBUG", split)
    rng = np.random.default_rng(5)
    comp = S.Composition()
    for i in range(150):
        x0, y0 = rng.uniform(0, 300, 2)
        comp.get_mut_or_insert_default(i).insert(S.custom_circle(float(x0), float(y0), float(rng.uniform(5, 30)))) \
            .set_props(S.solid((float(rng.random()), float(rng.random()), float(rng.random()), 0.7)))
    o = orc.Oracle()
    c = forma_amd.Context(0)
    try:
        for shift in (-5000.0, -5000.0, 0.0, 0.0, -120.0, 40.0, -5000.0, 0.0):
            for order, layer in comp.layers.items():
                layer.set_transform([1.0, 0.0, 0.0, 1.0, shift, 0.0])
            t = comp.tables(o)
            S.load(o, t)
            c.set_geoms(t["geoms"])
            if shift == -5000.0 and not getattr(c, "_loaded", False):
                S.load(c, t)
                c._loaded = True
                c.render(320, 320, clear=(1, 1, 1, 1))                  # first frame of the scene: synchronous
            want = o.render(320, 320, clear=(1, 1, 1, 1))
            dst = np.full((320, 320 * 4), 0x5A, np.uint8)
            got = np.asarray(c.render(320, 320, clear=(1, 1, 1, 1), dst=dst, stride=320 * 4)).reshape(want.shape)
            assert np.array_equal(c.segments(1), o.segments(1)), (split, shift)
            assert np.abs(want.astype(int) - got.astype(int)).max() <= 1, (split, shift)
    finally:
        c.close()
