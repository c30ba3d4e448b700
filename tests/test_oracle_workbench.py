"""Pins the oracle's LayerWorkbench / optimizer-pass restatement (oracle/forma_oracle.cpp: Workbench,
tile_unchanged_pass, skip_trivial_clips_pass, skip_fully_covered_layers_pass, drive_tile_painting) to the
reference's own unit tests of `forma/src/cpu/painter/layer_workbench/mod.rs:346-1307` — every test of that
module except `masked_vec` (a container test; its behaviour is covered through the ids the passes leave).
Each test below is the reference test of the same name, replayed through oracle.Workbench: same carries,
same segments, same props, same cached tile state, same expected ids / ControlFlow / TileWriteOp."""
import numpy as np
import pytest

import scene as S
from oracle import oracle as orc

WHITEF = (1.0, 1.0, 1.0, 1.0)
BLACKF = (0.0, 0.0, 0.0, 0.0)          # sic: the reference test module's BLACKF has alpha 0 (mod.rs:364-369)
REDF = (1.0, 0.0, 0.0, 1.0)
RED = [255, 0, 0, 255]
WHITE = [255, 255, 255, 255]
PARTIAL = [1] * 16                      # cover(id, CoverType::Partial), mod.rs:452-468
FULL = [16] * 16                        # consts::PIXEL_WIDTH

CONT, BRK_NONE, BRK_SOLID = orc.Workbench.CONTINUE, orc.Workbench.BREAK_NONE, orc.Workbench.BREAK_SOLID


def segment(layer_id):                  # mod.rs:470-472
    return orc.pixel_segment(layer_id, 0, 0, 0, 0, 0, 0)


def bench(props_by_layer, unchanged=None):
    """An oracle whose style table holds `props_by_layer` (dict id -> scene.Props) + a workbench on top of it."""
    n = max(props_by_layer) + 1
    offsets = np.full(n, S.NONE, np.uint32)
    words = []
    for lid, p in sorted(props_by_layer.items()):
        offsets[lid] = len(words)
        words += S.encode_props(p, [])
    un = np.zeros(n, np.uint8)
    for lid in range(n):
        un[lid] = 1 if (unchanged and unchanged(lid)) else 0
    o = orc.Oracle()
    o.set_styles(offsets, np.asarray(words, np.uint32), un)
    return o, orc.Workbench(o)


DEFAULT = S.Props()                     # Props::default(): NonZero, Draw(solid black alpha 1, Over, not clipped)


def test_populate_layers():             # mod.rs:474-537
    o, wb = bench({i: DEFAULT for i in range(6)})
    wb.init([(0, PARTIAL), (3, PARTIAL), (4, PARTIAL)])
    wb.context([segment(0), segment(1), segment(1), segment(2), segment(5), segment(5), segment(5)],
               cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    assert wb.ids() == [0, 1, 2, 3, 4, 5]
    assert [wb.segment_range(i) for i in range(6)] == [(0, 0), (1, 2), (3, 3), None, None, (4, 6)]
    assert [wb.queue_index(i) for i in range(6)] == [0, None, None, 1, 2, None]


def test_skip_unchanged():              # mod.rs:539-653
    o, wb = bench({i: DEFAULT for i in range(6)}, unchanged=lambda lid: lid < 5)
    wb.cached_tile(layer_count=4)
    segs = [segment(i) for i in range(5)]
    wb.context(segs, cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    # Optimization should fail because the number of layers changed.
    assert wb.tile_unchanged_pass()[0] == CONT
    assert wb.cached_tile_state()[0] == 5
    # Skip should occur because the previous pass updated the number of layers.
    wb.context(segs, cached_clear_color=BLACKF, clear_color=BLACKF)
    assert wb.tile_unchanged_pass()[0] == BRK_NONE
    assert wb.cached_tile_state()[0] == 5
    # Optimization should fail because at least one layer changed.
    wb.context([segment(i) for i in range(1, 6)], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.next_tile(); wb.populate_layers()
    assert wb.tile_unchanged_pass()[0] == CONT
    assert wb.cached_tile_state()[0] == 5
    # Optimization should fail because the clear color changed.
    wb.context(segs, cached_clear_color=BLACKF, clear_color=WHITEF)
    wb.next_tile(); wb.populate_layers()
    assert wb.tile_unchanged_pass()[0] == CONT
    assert wb.cached_tile_state()[0] == 5


def test_skip_full_clip():              # mod.rs:655-714
    props = {0: DEFAULT, 1: S.Props(clip=1), 2: S.Props(is_clipped=True), 3: S.Props(clip=1)}
    o, wb = bench(props)
    wb.init([(0, PARTIAL), (1, FULL), (2, PARTIAL), (3, FULL)])
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    wb.skip_trivial_clips_pass()
    assert wb.ids() == [0, 2]
    assert not wb.skip_clipping_contains(0)
    assert wb.skip_clipping_contains(2)


def test_skip_layer_outside_of_clip():  # mod.rs:716-760
    o, wb = bench({0: S.Props(is_clipped=True), 1: S.Props(is_clipped=True)})
    wb.init([(0, PARTIAL), (1, PARTIAL)])
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    wb.skip_trivial_clips_pass()
    assert wb.ids() == []


def test_skip_without_layer_usage():    # mod.rs:762-811
    props = {0: DEFAULT, 1: S.Props(clip=1), 3: DEFAULT, 4: S.Props(clip=1)}
    o, wb = bench(props)
    wb.init([(0, PARTIAL), (1, PARTIAL), (3, PARTIAL), (4, PARTIAL)])
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    wb.skip_trivial_clips_pass()
    assert wb.ids() == [0, 3]


def test_skip_everything_below_opaque():    # mod.rs:813-858
    o, wb = bench({i: DEFAULT for i in range(4)})
    wb.init([(0, PARTIAL), (1, PARTIAL), (2, FULL)])
    wb.context([segment(3)], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    assert wb.skip_fully_covered_layers_pass()[0] == CONT
    assert wb.ids() == [2, 3]


def gray50(blend):
    return S.Props(fill=(0.5, 0.5, 0.5, 0.5), blend_mode=blend)


def test_blend_top_full_layers():       # mod.rs:860-920
    o, wb = bench({0: gray50("Over"), 1: gray50("Multiply")})
    wb.init([(0, FULL), (1, FULL)])
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    assert wb.skip_fully_covered_layers_pass() == (BRK_SOLID, (0.28125, 0.28125, 0.28125, 0.75))


def test_blend_top_full_layers_with_clear_color():   # mod.rs:922-978
    o, wb = bench({0: gray50("Multiply"), 1: gray50("Multiply")})
    wb.init([(0, FULL), (1, FULL)])
    wb.context([], cached_clear_color=WHITEF, clear_color=WHITEF)
    wb.populate_layers()
    assert wb.skip_fully_covered_layers_pass() == (BRK_SOLID, (0.5625, 0.5625, 0.5625, 1.0))


def test_skip_fully_covered_layers_clip():           # mod.rs:980-1029
    o, wb = bench({0: S.Props(clip=1), 1: S.Props(blend_mode="Multiply")})
    wb.init([(0, PARTIAL), (1, FULL)])
    wb.context([], cached_clear_color=WHITEF, clear_color=WHITEF)
    wb.populate_layers()
    assert wb.skip_fully_covered_layers_pass()[0] == CONT


def test_skip_clip_then_blend():        # mod.rs:1031-1080
    o, wb = bench({0: S.Props(clip=1), 1: gray50("Multiply")})
    wb.init([(0, PARTIAL), (1, FULL)])
    wb.context([], cached_clear_color=WHITEF, clear_color=WHITEF)
    assert wb.drive_tile_painting() == (orc.Workbench.OP_SOLID, [224, 224, 224, 255])


def test_skip_visible_is_unchanged():   # mod.rs:1082-1217
    props = {0: DEFAULT, 1: DEFAULT, 2: S.Props(fill=REDF)}
    o, wb = bench(props, unchanged=lambda lid: lid != 0)
    wb.init([(0, PARTIAL), (1, PARTIAL), (2, FULL)])
    wb.cached_tile(layer_count=3)
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    # Tile has changed because layer 0 changed.
    assert wb.tile_unchanged_pass()[0] == CONT
    # However, we can still skip drawing because everything visible is unchanged.
    assert wb.skip_fully_covered_layers_pass()[0] == BRK_NONE

    wb.cached_tile(layer_count=2)
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()                # (the reference re-populates without next_tile here)
    # Tile has changed because layer 0 changed and number of layers has changed.
    assert wb.tile_unchanged_pass()[0] == CONT
    # We can still skip the tile because any newly added layer is covered by an opaque layer.
    assert wb.skip_fully_covered_layers_pass()[0] == BRK_NONE

    wb.cached_tile(layer_count=4)
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    assert wb.tile_unchanged_pass()[0] == CONT
    # This time we cannot skip because there might have been a visible layer last frame that is now removed.
    assert wb.skip_fully_covered_layers_pass() == (BRK_SOLID, REDF)


def test_skip_solid_color_is_unchanged():   # mod.rs:1219-1305
    o, wb = bench({0: S.Props(fill=REDF)})
    wb.init([(0, FULL)])
    wb.cached_tile(use=False)
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    # We can't skip drawing because we don't have any cached tile.
    assert wb.drive_tile_painting() == (orc.Workbench.OP_SOLID, RED)

    # (drive_tile_painting ends with next_tile(): the carry of the full layer 0 is the next tile's queue)
    assert wb.queue() == [(0, FULL)]
    wb.cached_tile(layer_count=0, solid_color=WHITE)
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    # We can't skip drawing because the tile solid color (RED) is different from the previous one (WHITE).
    assert wb.drive_tile_painting() == (orc.Workbench.OP_SOLID, RED)

    wb.cached_tile(layer_count=0, solid_color=RED)
    wb.context([], cached_clear_color=BLACKF, clear_color=BLACKF)
    wb.populate_layers()
    # We can skip drawing because the tile solid color is unchanged.
    assert wb.drive_tile_painting() == (orc.Workbench.OP_NONE, None)


def test_masked_ids_after_mask_and_skip():
    """MaskedVec::{set_mask, skip_until, iter_masked} (mod.rs:64-118, test `masked_vec` :417-450) as the passes use it:
    a full opaque cover in the middle hides everything below (skip_until), a full clip is masked out (set_mask)."""
    props = {0: DEFAULT, 1: DEFAULT, 2: S.Props(fill=REDF), 3: S.Props(clip=2), 4: S.Props(is_clipped=True), 6: DEFAULT}
    o, wb = bench(props)
    wb.init([(0, PARTIAL), (1, PARTIAL), (2, FULL), (3, FULL), (4, PARTIAL), (6, PARTIAL)])
    wb.context([], clear_color=BLACKF)
    wb.populate_layers()
    wb.skip_trivial_clips_pass()
    assert wb.ids() == [0, 1, 2, 4, 6] and wb.skip_clipping_contains(4)
    assert wb.skip_fully_covered_layers_pass()[0] == CONT
    assert wb.ids() == [2, 4, 6]
    assert wb.ids(masked=False) == [0, 1, 2, 3, 4, 6]
