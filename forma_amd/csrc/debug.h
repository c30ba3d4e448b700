// debug.h — the ONE place libforma_hip.so looks at the environment.
//
// FORMA_HIP_DEBUG = "name[=value],name,..." holds the switches tests and tools use to force or forbid one path each; a
// deployment never sets it.  Parsed when a context is created (tests flip switches between contexts of one process), except
// the two process-wide ones noted below.
//   sync                 no read-back-free frames (every frame reads N, the key masks and J back)
//   global_runsort       the carry pre-pass never orders a row's runs in LDS
//   xgather              multi-GPU: always materialise the received stream before sorting it
//   no_small_carry       never pick the small-LDS variant of k_carry_rows
//   span_groups / no_span_groups   span group lists on every frame / never
//   carry_slices=N       N workgroups per tile row in the carry pre-pass
//   digit_bits=4|8|9     radix digit width (default: 8, or 9 where that saves a pass)
//   no_packed_copy       cache frames copy the whole crop out instead of the packed written tiles
//   no_simple_paint / force_simple_paint   the all-solid painter kernel never / always (process-wide, read once)
//   poison=BYTE          every fresh device allocation is filled with BYTE (process-wide, read once)
//   poison_frame=BYTE    every per-frame buffer is refilled with BYTE when a frame starts
//   no_bias              sort digits are never taken relative to a tile field's minimum (SortPlan::bias)
//   no_prezero           the frame's first kernel clears nothing for later stages (they use memsets: A/B of the folding)
//   no_ras_hist          the sort's digit histograms are always taken by k_sort_hist, never by the rasterizer
//   carry_half=0|1|N     never / by policy / with N slices per row: the 512-lane carry kernel (three workgroups per CU)
//   carry_covl=0|1       never / by policy (default): the carry kernel that stages a row's cover sums and style summaries in LDS
//   paint_quad=0|1|2     the four-tiles-per-wavefront painter of all-solid scenes: never / by policy / always
//   order_thr=N          the painters file a tile as heavy from N shader clocks on (no steering, never switched off): tests
//   sort_cus=N           persistent workgroups of a digit pass (0: one per CU; default: all, or half the CUs in a context with three frame slots — with two from 4 M keys on)
//   no_order             the painters always take their tiles in index order (PaintParams::order_*)
//   no_cull / force_cull the painters never / always drop the entries below a tile's topmost occluder (PaintParams::cull; default:
//                        once the geometry has had tiles beyond the wave painter's lists)
//   runs_chain=0|1       read-back-free frames never / always find their runs without the counting pass (k_runs_count): the chained
//                        k_runs_wave, runs numbered per tile row (default: streams of <= RUNS_CHAIN_MAX_TILES x 2 048 segments, one frame in flight)
//   runs_blk=0|1         read-back-free frames whose tile rows take one carry workgroup each never / always (default) number their runs per
//                        2 048-segment tile (launch_runs' BLOCKS: no counting pass, no look-back; k_carry_rows compacts the rows' records)
//   blk_round=N          (tests) tiles per round of k_carry_rows' table of a row's tiles under the BLOCKS numbering: a power of two <= 256 (default 256)
//   strip_tiles=N        the painter runs four strip wavefronts per tile on frames of <= N painted tiles (0: never)
//   tail_poll=0|1        the host waits for a read-back-free frame by hipStreamSynchronize / by polling the pinned word k_frame_tail writes last (default)
//   paint_split=0|1|N    a synchronous frame into caller memory paints in bands whose copies leave while the rest is painted: never / by
//                        policy (default: two bands) / always, in N equal bands (tests: small canvases too)
//   split_first=P        ... the policy's first band: P percent of the painted tile rows (default 25)
//   trim_debug           forma_hip_trim prints what it releases
//   force_exchange       forma_hip_create_multi with ONE device still builds the multi-device context (RCCL world of one)
//   xchg=copy            multi-device contexts exchange with device copies instead of RCCL
//   multi_layout=exchange|bands   the layout a multi-device context starts with (forma_hip_multi_layout; default: auto)
#pragma once
#include <cstdlib>
#include <cstring>
#include <algorithm>

struct ForMaDebug {
    bool sync = false, global_runsort = false, xgather = false, no_small_carry = false, span_groups = false, no_span_groups = false;
    bool no_packed_copy = false, no_simple_paint = false, force_simple_paint = false, trim_debug = false, force_exchange = false;
    bool xchg_copy = false, no_prezero = false, no_bias = false, no_ras_hist = false, no_cull = false, force_cull = false, no_order = false;
    int carry_slices = 0, digit_bits = 0, poison = -1, poison_frame = -1, strip_tiles = -1, carry_half = 1, paint_quad = 1, order_thr = -1, sort_cus = -1, runs_chain = -1, multi_layout = 0, carry_covl = 1, runs_blk = -1, blk_round = 256, tail_poll = 1, paint_split = 1, split_first = 25;
};

inline ForMaDebug forma_debug_parse() {
    ForMaDebug d;
    const char* e = getenv("FORMA_HIP_DEBUG");
    if (!e) return d;
    char buf[512];
    strncpy(buf, e, sizeof buf - 1); buf[sizeof buf - 1] = 0;
    char* save = nullptr;   // strtok_r: contexts may be created from several threads at once
    for (char* tok = strtok_r(buf, ",; ", &save); tok; tok = strtok_r(nullptr, ",; ", &save)) {
        char* val = strchr(tok, '=');
        if (val) *val++ = 0;
        const long v = val ? strtol(val, nullptr, 0) : 0;
#define FD_FLAG(name) if (!strcmp(tok, #name)) { d.name = true; continue; }
        FD_FLAG(sync) FD_FLAG(global_runsort) FD_FLAG(xgather) FD_FLAG(no_small_carry) FD_FLAG(span_groups) FD_FLAG(no_span_groups)
        FD_FLAG(no_cull) FD_FLAG(force_cull) FD_FLAG(no_order) FD_FLAG(no_prezero) FD_FLAG(no_bias) FD_FLAG(no_ras_hist) FD_FLAG(no_packed_copy) FD_FLAG(no_simple_paint) FD_FLAG(force_simple_paint) FD_FLAG(trim_debug) FD_FLAG(force_exchange)
#undef FD_FLAG
        if (!strcmp(tok, "xchg")) { d.xchg_copy = val && !strcmp(val, "copy"); continue; }
        if (!strcmp(tok, "multi_layout")) { d.multi_layout = val && !strcmp(val, "exchange") ? 1 : (val && !strcmp(val, "bands") ? 2 : 0); continue; }
        if (!strcmp(tok, "carry_slices")) { d.carry_slices = (int)v; continue; }
        if (!strcmp(tok, "digit_bits")) { d.digit_bits = (int)v; continue; }
        if (!strcmp(tok, "carry_half")) { d.carry_half = (int)std::max(v, 0L); continue; }
        if (!strcmp(tok, "carry_covl")) { d.carry_covl = (int)std::max(v, 0L); continue; }
        if (!strcmp(tok, "sort_cus")) { d.sort_cus = (int)std::max(v, 0L); continue; }
        if (!strcmp(tok, "order_thr")) { d.order_thr = (int)std::max(v, 0L); continue; }
        if (!strcmp(tok, "paint_quad")) { d.paint_quad = (int)std::max(v, 0L); continue; }
        if (!strcmp(tok, "runs_chain")) { d.runs_chain = (int)std::max(v, 0L); continue; }
        if (!strcmp(tok, "runs_blk")) { d.runs_blk = (int)std::max(v, 0L); continue; }
        if (!strcmp(tok, "blk_round")) { d.blk_round = (int)std::min(std::max(v, 1L), 256L); continue; }
        if (!strcmp(tok, "paint_split")) { d.paint_split = (int)std::min(std::max(v, 0L), 8L); continue; }
        if (!strcmp(tok, "split_first")) { d.split_first = (int)std::min(std::max(v, 1L), 99L); continue; }
        if (!strcmp(tok, "tail_poll")) { d.tail_poll = (int)std::max(v, 0L); continue; }
        if (!strcmp(tok, "strip_tiles")) { d.strip_tiles = (int)std::max(v, 0L); continue; }
        if (!strcmp(tok, "poison")) { d.poison = (int)(v & 0xFF); continue; }
        if (!strcmp(tok, "poison_frame")) { d.poison_frame = (int)(v & 0xFF); continue; }
    }
    return d;
}
