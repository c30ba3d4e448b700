// api.cpp — the C ABI of libforma_hip.so (include/forma_hip.h): context, device buffers, frame
// orchestration on one HIP stream.  Product code: nothing here (or anywhere in forma_amd/) touches
// oracle/.  Errors never unwind across the ABI: every entry point returns a FORMA_E_* code.
#include <new>

#include "ctx.h"

#ifndef FORMA_ARCH
#define FORMA_ARCH "gfx950"
#endif

thread_local KernelTimer* g_ktimer = nullptr;
static inline float* kt_us(forma_hip_ctx* c) { return c->kt_dur_us; }

int fd_fail(forma_hip_ctx* c, int code, const char* what, hipError_t e) {
    if (c) snprintf(c->err, sizeof c->err, "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
    return code;
}

namespace {

inline int fail(forma_hip_ctx* c, int code, const char* what, hipError_t e = hipSuccess) { return fd_fail(c, code, what, e); }
void share_scene(forma_hip_ctx* o);                        // (defined with the frame-slot code below)
void invalidate_counts(forma_hip_ctx* o);
inline forma_hip_ctx* last_slot(forma_hip_ctx* ctx) { return ctx->last ? ctx->last : ctx; }
#define HIPCHECK(expr)                                                        \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return fail(ctx, FORMA_E_HIP, #expr, _e);       \
    } while (0)

// A timed frame: every kernel launched between stage_begin and stage_end carries its own pair of events (FORMA_LAUNCH,
// common.h) and is booked on the stage; only the copy into caller memory — not a kernel — is bracketed by markers.  (Until
// round 4 every stage was bracketed: ~12 us of marker packets per stage sat inside the stage times.)
inline void stage_begin(forma_hip_ctx* c, int st, bool timing) {
    if (!timing) return;
    c->stage_used[st] = true;
    if (st == ST_D2H) { (void)hipEventRecord(c->ev0[st], c->stream); return; }
    c->kt.cur_stage = st; c->kt.stream = c->stream; g_ktimer = &c->kt;
}
inline void stage_end(forma_hip_ctx* c, int st, bool timing) {
    if (!timing) return;
    if (st == ST_D2H) { (void)hipEventRecord(c->ev1[st], c->stream); return; }
    g_ktimer = nullptr;
}

int read_info(forma_hip_ctx* ctx) {
    HIPCHECK(hipMemcpyAsync(ctx->h_info, ctx->info.p, sizeof(FrameInfo), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return FORMA_OK;
}

int reset_info(forma_hip_ctx* ctx) {              // device-to-device from a template: no host round trip
    // (a read-back-free frame ends with k_frame_tail, which leaves the device copy pristine: the next frame's reset is free)
    if (!ctx->info_clean) HIPCHECK(hipMemcpyAsync(ctx->info.p, ctx->info_init.p, sizeof(FrameInfo), hipMemcpyDeviceToDevice, ctx->stream));
    ctx->info_clean = false;
    return FORMA_OK;
}
// the end of a read-back-free frame: FrameInfo to pinned host memory (and / or the segment count to a pinned word), device copy reset
int frame_tail(forma_hip_ctx* ctx, bool to_host_info, uint32_t* host_count) {
    const bool timed = ctx->stage_used[ST_PAINT];         // (a timed frame books the tail on the paint stage)
    stage_begin(ctx, ST_PAINT, timed);
    launch_frame_tail(ctx->stream, ctx->info.as<FrameInfo>(), to_host_info ? ctx->h_info : nullptr, host_count,
                      ctx->order_cnt_dev, ctx->order_keep_dev, ctx->chain_rows, ctx->n_chain_rows,
                      to_host_info ? ctx->h_seq : nullptr, to_host_info ? ++ctx->tail_seq : 0u);
    ctx->order_cnt_dev = nullptr; ctx->order_keep_dev = nullptr; ctx->chain_rows = nullptr; ctx->n_chain_rows = 0;
    stage_end(ctx, ST_PAINT, timed);
    HIPCHECK(hipGetLastError());
    ctx->info_clean = true;
    return FORMA_OK;
}

int check_canvas(forma_hip_ctx* ctx, uint32_t width, uint32_t height) {
    if (width == 0 || height == 0 || width > FORMA_MAX_WIDTH || height > FORMA_MAX_HEIGHT)   // consts.rs:25-26
        return fail(ctx, FORMA_E_ARG, "canvas size out of range");
    return FORMA_OK;
}

// prepare + scan + compact.  Synchronous form (bound_n == 0): (n_segments, n_compact) are read back and block_first is
// valid for N.  Asynchronous form: nothing is read back, block_first is provisioned for bound_n segments.
int run_line_table(forma_hip_ctx* ctx, const LineSource& src, size_t n_lines, bool timing, uint32_t bound_n = 0,
                   const ZeroJobs* zero = nullptr) {
    HIPCHECK(ctx->cl_idx.ensure(n_lines * 4));
    HIPCHECK(ctx->cl_start.ensure(n_lines * 4));
    HIPCHECK(ctx->prep_scratch.ensure(prepare_scratch_words(n_lines) * 4));
    HIPCHECK(ctx->block_first.ensure(std::max<size_t>(4096, ((size_t)bound_n / RAS_TILE + 2) * 4)));
    const uint32_t bf_cap = (uint32_t)std::min<size_t>(ctx->block_first.cap / 4, 0xFFFFFFFFu);
    stage_begin(ctx, ST_PREPARE, timing);
    launch_prepare_compact(ctx->stream, src, (uint32_t)n_lines, ctx->cl_idx.as<uint32_t>(), ctx->cl_start.as<uint32_t>(),
                           ctx->block_first.as<uint32_t>(), bf_cap, ctx->prep_scratch.as<uint32_t>(), ctx->info.as<FrameInfo>(), zero);
    stage_end(ctx, ST_PREPARE, timing);
    HIPCHECK(hipGetLastError());
    if (bound_n) return FORMA_OK;
    int rc = read_info(ctx);
    if (rc) return rc;
    const size_t N = ctx->h_info->n_segments;
    ctx->n_seg = N; ctx->n_compact = ctx->h_info->n_compact;
    const size_t need = (N + RAS_TILE - 1) / RAS_TILE + 1;
    if (need > bf_cap) {                      // first frame at this size: grow, rebuild the block table
        HIPCHECK(ctx->block_first.ensure(need * 4 * 2));
        launch_block_first(ctx->stream, ctx->cl_start.as<uint32_t>(), (uint32_t)ctx->n_compact, (uint32_t)N,
                           ctx->block_first.as<uint32_t>());
    }
    return FORMA_OK;
}

LineSource geometry_source(forma_hip_ctx* ctx, uint32_t width, uint32_t height) {
    LineSource S;
    memset(&S, 0, sizeof S);
    const size_t l0 = ctx->line_ranged ? ctx->line_lo : 0;              // a share of the lines: line i joins points i and i + 1
    S.x = ctx->x.as<float>() + l0; S.y = ctx->y.as<float>() + l0; S.line_slot = ctx->line_slot.as<uint32_t>() + l0;
    S.geoms = ctx->geoms.as<forma_geom_t>(); S.n_geoms = (uint32_t)ctx->n_geoms;
    S.width = (float)width; S.height = (float)height;
    S.band_lo = -3.0e38f; S.band_hi = 3.0e38f;
    // (an eighth of a pixel = two sub-pixel steps wider than the band: a line within half a sub-pixel step of the band's edge can
    //  round INTO the band's first or last tile row (a zero-cover segment: invisible, but part of the sorted stream) — lines are
    //  only culled where no rounding can bring them back; k_rasterize's exact tile-row test flags what a kept line leaves outside)
    if (ctx->band_row1 > 0) { S.band_lo = (float)(ctx->band_row0 * 16u) - 0.125f; S.band_hi = (float)(ctx->band_row1 * 16u) + 0.125f; }
    return S;
}

int finish_rasterize(forma_hip_ctx* ctx) {
    int rc = read_info(ctx);
    if (rc) return rc;
    uint64_t k_or = (uint64_t)ctx->h_info->key_or | ((uint64_t)ctx->h_info->key_or_hi << 32);
    uint64_t k_and = (uint64_t)ctx->h_info->key_and | ((uint64_t)ctx->h_info->key_and_hi << 32);
    ctx->live44 = (k_or ^ k_and) & 0xFFFFFFFFFFFull;
    ctx->layer_sorted = ctx->h_info->layer_unsorted == 0;
    return FORMA_OK;
}

SortPlan frame_sort_plan(forma_hip_ctx* ctx, uint64_t live44, bool layer_sorted, int digit_bits, bool speculated, bool* biased);

// stages 1-2 on the uploaded geometry: line table + rasterize -> seg_u.
// bound_n != 0: fully asynchronous (no read-back): N is only known to the device, buffers / grids are provisioned for
// bound_n segments, the sort plan is the speculated one.
int run_rasterize_frame(forma_hip_ctx* ctx, uint32_t width, uint32_t height, bool timing, bool speculate = false,
                        uint32_t bound_n = 0, const ZeroJobs* zero = nullptr, const forma_hip_ctx::PreZero* cleared = nullptr,
                        bool hist_too = false /* count the speculated sort plan's digits while the keys are made (RasHist) */) {
    const size_t all_lines = ctx->n_points ? ctx->n_points - 1 : 0;
    const size_t n_lines = ctx->line_ranged ? std::min(ctx->line_hi, all_lines) - std::min(ctx->line_lo, all_lines) : all_lines;
    ctx->n_lines = n_lines;
    ctx->n_seg = 0; ctx->n_compact = 0; ctx->have_unsorted = true; ctx->live44 = 0; ctx->layer_sorted = true;
    ctx->speculated = false; ctx->ras_hist_on = false;
    ctx->pz = forma_hip_ctx::PreZero();                   // (nothing of this frame has been cleared ahead of its stage yet)
    int rc = reset_info(ctx);
    if (rc) return rc;
    if (n_lines == 0) return FORMA_OK;
    const LineSource S = geometry_source(ctx, width, height);
    if ((rc = run_line_table(ctx, S, n_lines, timing, bound_n, zero))) return rc;
    if (zero && cleared) ctx->pz = *cleared;              // (k_line_len ran: the words of `zero` are cleared for the later stages)
    FrameInfo* dinfo = ctx->info.as<FrameInfo>();
    DevCount nc_seg, nc_cmp;
    if (bound_n) {
        nc_seg = DevCount{&dinfo->n_segments, bound_n};
        nc_cmp = DevCount{&dinfo->n_compact, (uint32_t)n_lines};
    } else {
        const size_t N = ctx->n_seg;
        if (N == 0) return FORMA_OK;
        if (N >= (1ull << 30)) return fail(ctx, FORMA_E_CAPACITY, "more than 2^30-1 pixel segments on one device");
        nc_seg = DevCount{nullptr, (uint32_t)N};
        nc_cmp = DevCount{nullptr, (uint32_t)ctx->n_compact};
    }
    HIPCHECK(ctx->seg_u.ensure(((size_t)nc_seg.bound + SEG_PAD) * 8));
    HIPCHECK(ctx->ras_masks.ensure(((size_t)nc_seg.bound / RAS_TILE + 2) * 32));
    // the sort's histograms from the rasterizer: a read-back-free frame whose first kernel cleared the sort's scratch, with
    // the plan the sort will use if the speculation holds (run_sort checks that it is the same plan)
    RasHist RH;
    memset(&RH, 0, sizeof RH);
    if (hist_too && bound_n && ctx->pred_valid && ctx->pz.sort_p && ctx->pz.sort_p == ctx->sort_counters.p && !ctx->dbg.no_ras_hist) {
        ctx->ras_plan = frame_sort_plan(ctx, ctx->pred_live44, ctx->pred_layer_sorted, ctx->digit_bits, true, nullptr);
        RH = make_ras_hist(ctx->ras_plan, ctx->sort_counters.as<uint32_t>());
        ctx->ras_hist_on = RH.hist != nullptr;
        // the tile fields' spans are worth measuring only where a biased plan could ever beat the plain one: the plain digits of
        // the tile fields take more than two passes (canvases beyond 255 tiles in a dimension), or this frame's plan is biased
        // already (its digits are checked against the spans).  4K and below: no — 80 VALU instructions per rasterizer lane less.
        {
            const SortPlan plain = make_segment_sort_plan(ctx->pred_live44, ctx->pred_layer_sorted, ctx->digit_bits, nullptr, nullptr);
            const SortPlan lay = ctx->pred_layer_sorted ? SortPlan{} : make_sort_plan((ctx->pred_live44 & 0x1FFFFFull) << 20, 20, 41, ctx->digit_bits);
            bool biased = false;
            for (int p = 0; p < ctx->ras_plan.n_passes; p++) biased |= ctx->ras_plan.fmask[p] != 0u;
            RH.track_range = (biased || plain.n_passes > (ctx->pred_layer_sorted ? 0 : lay.n_passes) + 2) ? 1u : 0u;
        }
    }
    stage_begin(ctx, ST_RASTER, timing);
    launch_rasterize(ctx->stream, S, nc_cmp, nc_seg, ctx->cl_idx.as<uint32_t>(), ctx->cl_start.as<uint32_t>(),
                     ctx->block_first.as<uint32_t>(), ctx->seg_u.as<uint64_t>(), dinfo, (int)ctx->band_row0,
                     (int)ctx->band_row1, ctx->ras_masks.as<uint32_t>(), /*reduce_now=*/bound_n == 0, &RH);
    // read-back-free frame: the masks stay per-workgroup records until k_runs_count combines them (nothing reads them earlier)
    ctx->pending_masks = bound_n ? PendingMasks{ctx->ras_masks.as<uint32_t>(), 0u, ctx->ras_hist_on ? 1u : 0u} : PendingMasks{nullptr, 0u, 0u};
    stage_end(ctx, ST_RASTER, timing);
    HIPCHECK(hipGetLastError());
    ctx->speculated = (speculate || bound_n) && ctx->pred_valid;
    if (ctx->speculated) {
        ctx->live44 = ctx->pred_live44; ctx->layer_sorted = ctx->pred_layer_sorted;
        // (asynchronous frames: k_runs_count verifies the plan on the device before anything relies on the sort order)
        return FORMA_OK;
    }
    return finish_rasterize(ctx);
}

// Strip painters (k_paint_wave<.., NPX = 1>: four wavefronts per tile) for frames whose tiles do not fill the chip's wave slots
// with one wavefront each — 1080p canvases, the band of a multi-device rank, crops: such a launch is as long as its deepest
// tile, and a strip walks that tile's pixels four times as fast.  Bigger frames are bound by throughput, and the strips'
// fourfold list work would cost them.  FORMA_HIP_DEBUG=strip_tiles=N moves the limit (0: never).
// ... and only for scenes with COSTLY layers (gradients, textures, the fifteen other blend modes, clipped layers: ~7 500 clocks per
// layer and tile against ~900 for a solid colour blended Over): a strip does a quarter of a layer's pixel arithmetic but all of its
// fixed work — segment loads, LDS phases — and builds the tile's list itself.  Translucent discs 120 layers deep lose 16 % to
// strips, the 4K scene's band (one layer in seven costly) gains 20 %.
// A digit pass is one persistent 1 024-lane workgroup per CU: 148 KB of LDS and the whole register file, nothing co-resides.  With
// frames in flight the passes of the frames therefore run one after the other on an otherwise idle chip; on half the CUs a pass
// is slower (89 against 60 us) but costs less CU-time, and the other frames' kernels run beside it.  Round 6 swept it
// (tools/policy_sweep.py, profiles/r06_policy_sweep.json: 4 scene families x 5 canvases x 1-4 slots, each setting forced): with one
// slot half the CUs lose 4-16 % everywhere; with THREE they win 0-5 % in 18 of 20 cells (never lose more than 1 %); with TWO they
// win 1-5 % on frames of >= 4 M pixel segments and lose as much on smaller ones; with four it is a wash (+-2 %).  96 and 64 CUs
// with three slots: +1.4 % / -1.7 % on the 4K scene (profiles/r06_experiments.txt) - the rate is set by the CU-time of all the
// frame's kernels, not by the passes alone.  A frame whose image also crosses PCIe behind its kernels (forma_hip_render_enqueue)
// keeps all CUs: there the passes are on the frame's critical path (two slots: 1 453 -> 1 134 frames/s with half of them).
// FORMA_HIP_DEBUG=sort_cus=N sets it (0: all).
static uint32_t sort_workgroups(const forma_hip_ctx* ctx, size_t n_keys) {
    if (ctx->dbg.sort_cus >= 0) return (uint32_t)ctx->dbg.sort_cus;
    const forma_hip_ctx* o = ctx->owner ? ctx->owner : ctx;
    const size_t slots = o->slots.size();
    if (ctx->frame_has_dst) return 0u;
    return slots == 3 || (slots == 2 && n_keys >= ((size_t)1 << 22)) ? o->n_cus / 2u : 0u;
}

// Both schedules that shorten ONE frame's painter launch at the price of more work — strips, and the heavy-first order below —
// are for a context with one frame in flight.  With frame slots (forma_hip_set_frames_in_flight) the tail of a launch is filled by
// the other frames' kernels anyway and the extra work is a loss: measured with three slots, the 4K scene -2.5 % frames/s with the
// order on, its 1/8 band -5 % with strips.
#ifndef CARRY_HALF_SLICED
#define CARRY_HALF_SLICED 1
#endif
static bool one_frame_in_flight(const forma_hip_ctx* ctx) {
    const forma_hip_ctx* o = ctx->owner ? ctx->owner : ctx;
    return o->slots.size() <= 1;
}
static bool paint_by_strips(const forma_hip_ctx* ctx, uint32_t tiles_painted) {
    if (ctx->dbg.strip_tiles >= 0) return tiles_painted <= (uint32_t)ctx->dbg.strip_tiles;
    return tiles_painted <= paint_strip_tiles(ctx) && !ctx->scene_simple && (uint64_t)ctx->costly_layers * 32u >= ctx->n_orders && one_frame_in_flight(ctx);
}

// Quad painters (k_paint_quad: four tiles per wavefront) for all-solid scenes whose tiles are shallow AND many: the list work of
// a tile with a handful of entries is done by a 16-lane group, but a wavefront then paints its four tiles' pixels one after the
// other — a gain where the launch is many rounds of wavefronts (the 8K scene, 262 144 tiles: paint 329 -> 287 us, +13 % frames/s
// pipelined), a loss where it is one or two (a 64-row band of that scene: 56 -> 72 us; 1080p: 23 -> 40 us).  Shallow = few runs
// per tile (a tile with more than 16 entries after culling goes to k_paint_deep).  FORMA_HIP_DEBUG=paint_quad=0|2: never / always.
static bool paint_by_quads(const forma_hip_ctx* ctx, int cache_id, uint32_t runs_bound, uint32_t tiles) {
    if (!ctx->scene_simple || ctx->dbg.no_simple_paint || cache_id >= 0 || ctx->dbg.paint_quad == 0) return false;
    if (ctx->dbg.paint_quad == 2) return true;
    return tiles >= 16u * 8192u && (uint64_t)runs_bound <= 8ull * tiles;     // (>= 4 rounds of quads on the chip's 8 192 wave slots)
}

// A biased plan met a void frame.  plan_bad has other causes too (a slice beyond the small carry variant, a count over its
// bound), so this is a suspicion, not a proof: plain digits for a while, then the cheaper plan is tried again; every repeat
// doubles the ban (64 .. 4096 frames), new geometry lifts it (invalidate_counts).
static void ban_bias(forma_hip_ctx* ctx) {
    ctx->bias_ban_len = ctx->bias_ban_len ? std::min(ctx->bias_ban_len * 2u, 4096u) : 64u;
    ctx->bias_banned = ctx->bias_ban_len;
}

// the digit plan of a frame's segment sort: live key bits only; a stream that is already non-decreasing in layer needs a
// stable sort by tile alone
SortPlan frame_sort_plan(forma_hip_ctx* ctx, uint64_t live44, bool layer_sorted, int digit_bits, bool speculated, bool* biased) {
    // the tile fields relative to their minima (one digit fewer on 4096- and 8192-pixel canvases): only on read-back-free
    // frames — the span is the PREVIOUS frame's, k_sort_hist checks this frame's keys against it and voids the frame otherwise
    const KeyRange* range = speculated && ctx->pred_range.valid && !ctx->bias_banned && !ctx->dbg.no_bias ? &ctx->pred_range : nullptr;
    return make_segment_sort_plan(live44, layer_sorted, digit_bits, range, biased);
}

// What the frame's FIRST kernel (k_line_len) clears for the later stages of a read-back-free frame — the sort's histograms,
// tickets and status rows, the tile tables — instead of memset operations on the stream: a
// band frame of a multi-device context is ~250 us of kernels, and every stream operation costs the host ~5 us and the device a
// dispatch of its own.  The buffers are grown here, before anything of the frame is enqueued, so that the stages' own
// `ensure` calls find them in place.  `sort_n`: the keys the frame's sort will see (its provisioning bound).
int plan_zero_jobs(forma_hip_ctx* ctx, uint32_t width, uint32_t height, uint32_t sort_n, uint32_t runs_n, ZeroJobs* Z,
                   forma_hip_ctx::PreZero* cleared) {
    memset(Z, 0, sizeof *Z);
    *cleared = forma_hip_ctx::PreZero();
    if (ctx->dbg.no_prezero) return FORMA_OK;
    const uint32_t tiles_w = (width + 15) / 16, tiles_h = (height + 15) / 16;
    const SortPlan plan = frame_sort_plan(ctx, ctx->pred_live44, ctx->pred_layer_sorted, ctx->digit_bits, true, nullptr);
    HIPCHECK(ctx->sort_counters.ensure(sort_scratch_words(std::max<size_t>(sort_n, 1)) * 4));
    HIPCHECK(ctx->row_tab.ensure(row_tab_total_words(tiles_w, tiles_h) * 4));
    HIPCHECK(ctx->runs_scratch.ensure(runs_scratch_words(std::max<size_t>(runs_n, 1)) * 4));
    const size_t sw = sort_n > 1 && plan.n_passes ? sort_zero_words(sort_n, plan) : 0;
    const size_t tw = row_tab_zero_words(tiles_w, tiles_h);
    if (sw > 0xFFFFFFFFull || tw > 0xFFFFFFFFull) return FORMA_OK;           // (absurd sizes: the stages clear for themselves)
    if (sw) { Z->p[Z->n] = ctx->sort_counters.as<uint32_t>(); Z->words[Z->n++] = (uint32_t)sw; cleared->sort_p = ctx->sort_counters.p; cleared->sort_words = sw; }
    Z->p[Z->n] = ctx->row_tab.as<uint32_t>(); Z->words[Z->n++] = (uint32_t)tw; cleared->tab_p = ctx->row_tab.p; cleared->tab_words = tw;
    // (the status words of the chained run kernel: a word per 2 048 segments)
    const size_t cw = runs_n ? runs_chain_words(runs_n) : 0;
    if (cw && cw <= 0xFFFFFFFFull) { Z->p[Z->n] = ctx->runs_scratch.as<uint32_t>(); Z->words[Z->n++] = (uint32_t)cw; cleared->chain_p = ctx->runs_scratch.p; cleared->chain_words = cw; }
    return FORMA_OK;
}

// stage 3 on `src` (nc segments, device): result pointer in ctx->sorted
int run_sort(forma_hip_ctx* ctx, const uint64_t* src, DevCount nc, bool timing, int digit_bits = 0,
             const ChunkedSrc* chunked = nullptr) {
    ctx->n_passes = 0;
    const size_t n = nc.bound;
    if (n >= (1ull << 30)) return fail(ctx, FORMA_E_CAPACITY, "more than 2^30-1 pixel segments on one device");
    if (digit_bits == 0) digit_bits = ctx->digit_bits;
    HIPCHECK(ctx->seg_a.ensure((std::max<size_t>(n, 1) + SEG_PAD) * 8));
    HIPCHECK(ctx->seg_b.ensure((std::max<size_t>(n, 1) + SEG_PAD) * 8));
    HIPCHECK(ctx->sort_counters.ensure(sort_scratch_words(std::max<size_t>(n, 1)) * 4));
    const SortPlan plan = frame_sort_plan(ctx, ctx->live44, ctx->layer_sorted, digit_bits, ctx->speculated && nc.ptr != nullptr, &ctx->plan_biased);
    ctx->n_passes = plan.n_passes;
    ctx->sort_range = n > 1 && plan.n_passes ? sort_range_words(ctx->sort_counters.as<uint32_t>()) : nullptr;
    ctx->sort_range_n = sort_hist_blocks(n);
    bool zeroed = ctx->pz.sort_p == ctx->sort_counters.p && ctx->pz.sort_words >= sort_zero_words(n, plan);
    ctx->pz.sort_p = nullptr;
    // did the rasterizer count exactly this plan's digits?  (if it counted another plan's, the scratch is no longer clear)
    bool hist_ready = ctx->ras_hist_on && zeroed && !chunked && plan.n_passes == ctx->ras_plan.n_passes && src == ctx->seg_u.as<uint64_t>();
    for (int p = 0; hist_ready && p < plan.n_passes; p++)
        hist_ready = plan.shift[p] == ctx->ras_plan.shift[p] && plan.mask[p] == ctx->ras_plan.mask[p] && plan.bias[p] == ctx->ras_plan.bias[p] &&
                     plan.fmask[p] == ctx->ras_plan.fmask[p];
    if (ctx->ras_hist_on && !hist_ready) zeroed = false;
    ctx->ras_hist_on = false;
    if (hist_ready) ctx->sort_range = nullptr;               // (the spans travel in the rasterizer's mask records: PendingMasks::has_range)
    stage_begin(ctx, ST_SORT, timing);
    ctx->sorted = (uint64_t*)launch_radix_sort(ctx->stream, src, ctx->seg_a.as<uint64_t>(), ctx->seg_b.as<uint64_t>(), nc, plan,
                                               digit_bits, ctx->sort_counters.as<uint32_t>(), &ctx->info.as<FrameInfo>()->error,
                                               chunked,
                                               ctx->info.as<FrameInfo>(), zeroed, hist_ready, sort_workgroups(ctx, n));
    stage_end(ctx, ST_SORT, timing);
    HIPCHECK(hipGetLastError());
    return FORMA_OK;
}


// h_info holds a fresh copy of the device FrameInfo: remember the sort-plan inputs of this frame and, if the plan was
// speculated from the previous frame, verify it.  Called at the run-count read-back, i.e. BEFORE any kernel that
// assumes a correctly sorted stream is launched (everything up to there is in-bounds for any digit plan).
int verify_speculation(forma_hip_ctx* ctx) {
    if (!ctx->have_unsorted || !ctx->n_seg) return FORMA_OK;
    const uint64_t k_or = (uint64_t)ctx->h_info->key_or | ((uint64_t)ctx->h_info->key_or_hi << 32);
    const uint64_t k_and = (uint64_t)ctx->h_info->key_and | ((uint64_t)ctx->h_info->key_and_hi << 32);
    if (k_or == 0 && k_and == ~0ull) return FORMA_OK;            // masks untouched: the rasterizer did not run this frame
    const uint64_t live = (k_or ^ k_and) & 0xFFFFFFFFFFFull;
    const bool sorted = ctx->h_info->layer_unsorted == 0;
    const bool wrong = ctx->speculated && (live != ctx->live44 || sorted != ctx->layer_sorted);
    ctx->pred_valid = true; ctx->pred_live44 = live; ctx->pred_layer_sorted = sorted;
    return wrong ? FORMA_RETRY : FORMA_OK;
}

struct PaintArgs {
    uint32_t width, height;
    const uint8_t* channels;
    const float* clear;
    const forma_rect_t* crop;
    int cache_id = -1;
};

// A synchronous frame into caller memory is PCIe from the painter's last tile on: 33 MB of a 4K image take ~600 us behind ~450 us
// of kernels.  Painted in bands of tile rows — one launch and one event per band — the first band's pixels leave after an
// eighth of the painter instead, on a second stream, and the link is busy while the rest is painted (the copies then end ~70 us
// earlier; what the bands cost the painter — no heavy-first order across launches, eight tails — hides under the copy).
// Only read-back-free frames of render_on that deliver into `dst` ask for it (split_want), never with a cache (the written-tile
// set decides what is copied) and not for images of a few MB.  Returns the number of bands, 0: one launch as ever.
#define SPLIT_MIN_RUNS 262144u
int split_plan(forma_hip_ctx* ctx, const PaintArgs& a, const PaintParams& P, uint32_t bound_j, bool timing) {
    if (!ctx->split_want || !bound_j || timing || a.cache_id >= 0 || ctx->dbg.paint_split == 0) return 0;
    const uint32_t rows = P.crop_y1 > P.crop_y0 ? P.crop_y1 - P.crop_y0 : 0u;
    const uint32_t cols = P.crop_x1 > P.crop_x0 ? P.crop_x1 - P.crop_x0 : 0u;
    const bool forced = ctx->dbg.paint_split > 1;
    if (!forced && ((uint64_t)rows * cols < 16384u || rows < 32u)) return 0;   // (< 16 MB of pixels: the copy is not the frame)
    // ... and a painter of a few dozen microseconds has nothing to hide a copy under: the second copy's hand-over (~20 us) costs
    // more than starting the first one early gains (the 4K spaceship frame: a dozen shapes, 32 400 mostly empty tiles)
    if (!forced && bound_j < SPLIT_MIN_RUNS) return 0;
    if (rows < 2u) return 0;
    if (!ctx->copy_stream) {
        if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) { ctx->copy_stream = nullptr; return 0; }
        for (int k = 0; k < forma_hip_ctx::SPLIT_MAX; k++)
            if (hipEventCreateWithFlags(&ctx->split_ev[k], hipEventDisableTiming) != hipSuccess) {
                ctx->split_ev[k] = nullptr;
                for (int q = 0; q < k; q++) { (void)hipEventDestroy(ctx->split_ev[q]); ctx->split_ev[q] = nullptr; }   // (no stream: one launch as ever)
                (void)hipStreamDestroy(ctx->copy_stream); ctx->copy_stream = nullptr; return 0;
            }
    }
    return (int)std::min<uint32_t>(forced ? (uint32_t)ctx->dbg.paint_split : 2u, rows);   // (the policy: two bands, run_paint places the cut)
}

// stage 4 on ctx->sorted (n segments): runs + carry pre-pass + per-tile painter -> ctx->image
// bound_j != 0: asynchronous — the run count stays on the device (info->n_runs), buffers are provisioned for bound_j runs.
int run_paint(forma_hip_ctx* ctx, DevCount nc, const PaintArgs& a, bool timing, uint32_t bound_j = 0) {
    const size_t n = nc.bound;
    // set_styles and set_images are separate calls: the image a texture style names is checked where both are known
    if (ctx->any_texture && ctx->max_image_index >= ctx->n_images) return fail(ctx, FORMA_E_ARG, "texture style names an image that was not uploaded");
    const uint32_t tiles_w = (a.width + 15) / 16, tiles_h = (a.height + 15) / 16;
    const uint32_t T = tiles_w * tiles_h;
    ctx->img_w = a.width; ctx->img_h = a.height;
    ctx->chain_rows = nullptr; ctx->n_chain_rows = 0;
    // where this frame is painted: the cache's own image (it must keep what its buffer showed last frame) or the scratch one
    TileCacheArgs tc{nullptr, nullptr};
    uint32_t clear_unchanged = 0;
    ctx->cur_cache = a.cache_id;
    if (a.cache_id >= 0) {
        forma_hip_ctx::TileCache& c = ctx->caches[a.cache_id];
        HIPCHECK(c.tiles.ensure((size_t)T * 8));
        HIPCHECK(c.image.ensure((size_t)a.width * a.height * 4));
        HIPCHECK(ctx->cache_written.ensure((size_t)T));
        if (c.w != a.width || c.h != a.height) {                       // renderer.rs:94-111: new size -> cache cleared
            c.w = a.width; c.h = a.height; c.has_clear = false;
            HIPCHECK(hipMemsetAsync(c.tiles.p, 0, (size_t)T * 8, ctx->stream));
            HIPCHECK(hipMemsetAsync(c.image.p, 0, (size_t)a.width * a.height * 4, ctx->stream));
        }
        HIPCHECK(hipMemsetAsync(ctx->cache_written.p, 0, (size_t)T, ctx->stream));
        tc.tiles = c.tiles.as<uint2>(); tc.written = ctx->cache_written.as<uint8_t>();
        clear_unchanged = c.has_clear && memcmp(c.clear, a.clear, sizeof c.clear) == 0 ? 1u : 0u;
        ctx->cur_image = c.image.as<uint8_t>();
    } else {
        HIPCHECK(ctx->image.ensure((size_t)a.width * a.height * 4));
        ctx->cur_image = ctx->image.as<uint8_t>();
    }
    // one buffer, zeroed by ONE memset per frame (launch_runs; layout: row_tab_zero_words in common.h): [row_count: tiles_h + 1]
    // [row_span_lo | row_span_cnt: one pair per (row, slice of the carry pre-pass), 8 tiles_h + 1 each] [painter overflow
    // counters: wave -> deep, deep -> huge] [first-run table, T words]; the painter's overflow lists (T words of tiles, 2 T
    // words of {tile, entries}) follow un-zeroed
    HIPCHECK(ctx->row_tab.ensure(row_tab_total_words(tiles_w, tiles_h) * 4));
    FrameInfo* dinfo = ctx->info.as<FrameInfo>();
    uint32_t* row_count = ctx->row_tab.as<uint32_t>();
    uint32_t* row_span_lo = row_count + (tiles_h + 1);
    uint32_t* row_span_cnt = row_span_lo + (CR_MAX_SLICES_HOST * tiles_h + 1);
    uint32_t* paint_overflow = row_span_cnt + (CR_MAX_SLICES_HOST * tiles_h + 1);   // [0], [1], [2] = the three counts (wave -> mid, deep -> huge, mid -> deep), then the first-run table ...
    uint32_t* over2_n = paint_overflow + 1;
    uint32_t* mid_n = paint_overflow + 2;
    uint32_t* tile_first_run = paint_overflow + 3;
    uint32_t* order_cnt = tile_first_run + T;                           // ... the painters' order counts (PaintParams::order_cnt_out) ...
    uint32_t* overflow_list = order_cnt + PAINT_ORDER_WORDS;            // ... then the lists themselves
    uint32_t* over2_list = overflow_list + T;
    uint32_t* row_base = over2_list + 2 * (size_t)T;                    // (chain numbering: where each row's runs begin)
    uint32_t* mid_list = row_base + (tiles_h + 1);                      // {tile, entries} of the tiles beyond the mid tier's lists: 2 T words
    uint32_t J = 0;
    HIPCHECK(ctx->blk_edge.ensure(runs_blocks(std::max<size_t>(n, 1)) * sizeof(BlkEdge)));
    HIPCHECK(ctx->runs_scratch.ensure(runs_scratch_words(std::max<size_t>(n, 1)) * 4));
    stage_begin(ctx, ST_CARRY, timing);
    const bool tables_zero = ctx->pz.tab_p == ctx->row_tab.p && ctx->pz.tab_words >= row_tab_zero_words(tiles_w, tiles_h);
    ctx->pz.tab_p = nullptr;
    // the rows' runs are ordered inside k_carry_rows when they fit its LDS; else by a global sort
    // (the in-LDS key holds 16 layer bits: every order a geom can produce has to fit, not just the style table)
    bool local_sort = ctx->n_orders <= 65536 && ctx->max_geom_order < 65536 && !ctx->global_runsort;
    // A read-back-free frame whose rows are ordered in LDS needs no dense run numbering (the global run sort does): its runs are
    // found by ONE kernel, numbered per tile row from the index of the row's first segment (launch_runs' chain) — no counting
    // pass, the sorted stream is read once less.  The record arrays are then indexed like the segments: provisioned for N.
    // It pays while the run kernel's workgroups are all resident at once (1080p: 9.8 + 15.3 -> 19.6 us, a 17-row band of the 4K
    // scene: 8.2 + 14.6 -> 16.0): the look-back is a round trip that a workgroup of 15 us cannot hide, and a frame of several
    // rounds of workgroups pays it in every round (4K: 23.8 + 52.8 -> 84.8 us, 8K: 16.4 + 31.0 -> 54.5) — those keep the counting pass.
    // With several frames in flight the waves that wait take issue slots from the other frames' kernels (1080p, three slots:
    // 9 942 -> 9 707 frames/s while the call alone gains 2.6 %): one frame in flight only, like the strip painters.
    const bool chain = bound_j != 0 && n > 0 && local_sort && ctx->pred_max_row <= carry_rows_local_cap() &&
                       (ctx->dbg.runs_chain < 0 ? (RUNS_CHAIN_DEFAULT != 0 && runs_chain_words(n) <= RUNS_CHAIN_MAX_TILES && one_frame_in_flight(ctx))
                                                : ctx->dbg.runs_chain != 0);
    const bool chain_zero = chain && ctx->pz.chain_p == ctx->runs_scratch.p && ctx->pz.chain_words >= runs_chain_words(n);
    ctx->pz.chain_p = nullptr;
    DevCount jc;
    // Several workgroups share a tile row, each a range of layers (k_carry_rows): as many as keep the chip busy for the rows
    // this frame paints (a multi-GPU band is a fraction of the canvas).  The small-LDS variant (a quarter of the keys per workgroup) is a
    // read-back-free frame's guess — its slices must fit 4096 runs, known only from a previous frame; a slice that does not
    // fit voids the frame (plan_bad), the synchronous re-run takes the large variant, and the guess is not made again.
    uint32_t crow0 = 0, crow1 = tiles_h;                    // the tile rows the painter visits (Rect::new, renderer.rs:43-52)
    if (a.crop) { crow0 = a.crop->y0 / 16; crow1 = std::min(tiles_h, (a.crop->y1 + 15) / 16); }
    const uint32_t rows_painted = std::max(crow1 > crow0 ? crow1 - crow0 : 0u, 1u);
    ctx->cur_rows_painted = rows_painted;
    // Slicing pays when a row is heavy (measured on a 17-row band of the 4K scene, 4 800 runs per row: carry stage 80 -> 57 us
    // with eight small workgroups per row) and costs when it is light (a 64-row band of the 8192 x 8192 scene, ~1 000 runs per
    // row: 48 -> 55 us): a slice should keep >= ~768 runs.
    auto slices_for = [&](uint32_t per_cu_budget, uint32_t max_row) {
        const uint32_t by_chip = std::max(1u, per_cu_budget / rows_painted), by_load = std::max(1u, max_row / 768u);
        return std::min<uint32_t>(CR_MAX_SLICES_HOST, std::min(by_chip, by_load));
    };
    uint32_t n_slices = 1;
    bool small = false, half = false, covl = false;
    if (bound_j) {
        local_sort = local_sort && ctx->pred_max_row <= carry_rows_local_cap();     // wrong guess -> plan_bad -> synchronous re-run
        const uint32_t pmr = ctx->pred_max_row == 0xFFFFFFFFu ? 0u : ctx->pred_max_row;
        n_slices = slices_for(256u, pmr);
        const uint32_t ks = slices_for(512u, pmr);
        // (a whole 4K frame — 135 rows, one workgroup each — gains nothing from three small workgroups per row and its painters
        //  pay for the extra span lists: measured 124 -> 126 us carry, 133 -> 139 us paint; bands of <= 128 rows do gain)
        if (local_sort && ks > 1u && rows_painted <= 128u && !ctx->small_banned && !ctx->no_small_carry && !ctx->force_slices) {
            const bool known = ctx->pred_slice_n == ks && ctx->pred_slice_small;
            const uint64_t guess = known ? (uint64_t)ctx->pred_max_slice * 10 / 9 : (uint64_t)ctx->pred_max_row * 5 / (3 * ks);
            if (guess <= carry_rows_small_cap()) { small = true; n_slices = ks; }
            // ... as 512-lane workgroups when the slices fit those AND the frame is many workgroups (the painters see the same ks
            // span lists either way).  Two measurements stand behind the count: 1080p, 68 rows x 3 slices of ~900 runs: 36.1 -> 27.3 us,
            // +4.7 % frames/s per call, +8.8 % with three slots; a 17-row band of the 4K scene, 6 slices of ~800 runs each with
            // long span lists: 25.4 -> 29.1 us — a hundred workgroups have a CU each and the 1 024-lane one finishes its slice sooner.
            if (small && CARRY_HALF_SLICED && ctx->dbg.carry_half == 1 && guess <= carry_rows_half_cap() && rows_painted * ks >= 192u) half = true;
        }
        // The HALF variant (512 lanes, one piece of <= 2048 runs, three workgroups per CU) for frames whose rows are LIGHT — the
        // 8192 x 8192 triangle scene (512 rows of ~1 000 runs), 1080p: one workgroup per row as before, but 768 of them resident
        // at once instead of 256, each without the large variant's 148 KB of LDS to itself (carry 57 -> 32 us on the 8K scene,
        // 42 -> 27 at 1080p).  Heavy rows (the 4K scene: 4 800 runs) gain nothing from 3-6 half workgroups per row — measured:
        // the carry kernel saves 0-12 us and the painters pay as much for the extra span lists.
        if (!small && local_sort && !ctx->small_banned && !ctx->no_small_carry && !ctx->force_slices && ctx->dbg.carry_half != 0 && pmr) {
            uint32_t kh = 1u;
            if (ctx->dbg.carry_half > 1) kh = (uint32_t)std::min<int>(ctx->dbg.carry_half, (int)CR_MAX_SLICES_HOST);   // (tools: N slices per row)
            const bool known = ctx->pred_slice_n == kh && ctx->pred_slice_small && ctx->pred_slice_half;
            const uint64_t guess = known ? (uint64_t)ctx->pred_max_slice * 10 / 9 : (kh > 1u ? (uint64_t)pmr * 5 / (3 * kh) : (uint64_t)pmr * 6 / 5);
            if (guess <= carry_rows_half_cap()) { small = true; half = true; n_slices = kh; }
        }
        ctx->small_tried = small;
    }
    // BLOCKS (launch_runs, k_carry_rows): a read-back-free frame whose tile rows take ONE carry workgroup each needs neither the counting
    // pass nor the chain: the run kernel numbers per 2 048-segment tile into sparse arrays and the row's workgroup compacts them.
    const uint32_t ns_final = ctx->force_slices ? std::min<uint32_t>(ctx->force_slices, CR_MAX_SLICES_HOST) : n_slices;
    // (... and takes a COVL variant of the carry kernel — decided below, from the same predictions)
    auto covl_choice = [&](uint32_t slices) {
        const uint32_t mxr = ctx->pred_max_row;
        const bool big = !small && !half && mxr != 0xFFFFFFFFu &&
                         (bound_j ? !ctx->covl_banned && (uint64_t)mxr + mxr / 64u <= carry_rows_covl_cap() : mxr <= carry_rows_covl_cap());
        return local_sort && slices == 1u && ctx->dbg.carry_covl != 0 && (big || (small && half && (ctx->dbg.carry_covl & 2) == 0));
    };
    // Measured (profiles/r06_experiments.txt, r6n-r6v): the 4K scene (135 heavy rows, a CU each) runs + carry 112.8 -> 94.1 us, frames/s per
    // call +3.9 %, three slots +3.8 %; the 8K triangle scene (512 light rows, two workgroups per CU) 72 -> 73 us — the head counts are one
    // more dependent round trip at the start of k_carry_rows, and with every workgroup of a full chip asking at once that costs what
    // the counting pass did: by default not for frames of more LIGHT rows (the 512-lane carry variant, several workgroups per CU) than CUs.
    // Sweep of the switch over 4 scene families x 5 canvases x 1 / 3 slots (profiles/r06_policy_sweep_blk.json): heavy rows gain at every
    // canvas, also with two rounds of row workgroups (the 30 000-layer scene at 8192 x 8192: +3.9 %); light rows gain up to 4K (+4 %) and lose
    // 0-2 % at 8192 x 8192.
    const bool blk = bound_j != 0 && n > 0 && !chain && covl_choice(ns_final) && n <= 0x3FFFFFFFull &&
                     (ctx->dbg.runs_blk < 0 ? (RUNS_BLK_DEFAULT != 0 && (rows_painted <= (uint32_t)ctx->n_cus || !(small && half))) : ctx->dbg.runs_blk != 0);
    if (bound_j) {
        jc = chain ? DevCount{nullptr, (uint32_t)n} : (blk ? DevCount{nullptr, bound_j} : DevCount{&dinfo->n_runs, bound_j});   // (chain: run indices are segment indices; blk: dense, counted by nobody before the tail)
        if (chain || blk) { ctx->chain_rows = row_count; ctx->n_chain_rows = tiles_h; }
    }
    // Records, run keys and digests are sized for the RUNS, not for the segments (a run needs a segment, so N would always do:
    // 440 MB of records per frame slot on the 4K scene, for 21 MB of runs).  A read-back-free frame has its bound from the last
    // verified frame; a synchronous one launches the counting kernel, reads the count and sizes the buffers before the kernel
    // that fills them.
    size_t cap = std::max<size_t>(chain ? n : bound_j, 1);
    auto runs = [&](int what) {
        launch_runs(ctx->stream, ctx->sorted, nc, tiles_w, tiles_h, blk ? ctx->rec_sp.as<TileRecord>() : ctx->records.as<TileRecord>(), (uint32_t)(blk ? n : cap),
                    ctx->rk_u.as<uint64_t>(), tile_first_run, ctx->blk_edge.as<BlkEdge>(), row_count,
                    ctx->runs_scratch.as<uint32_t>(), dinfo, /*verify_plan=*/bound_j != 0 && ctx->speculated, ctx->live44,
                    ctx->layer_sorted, ctx->pending_masks,
                    RunStyle{ctx->layer_sf.as<uint32_t>(), (uint32_t)ctx->n_orders,
                             (a.cache_id >= 0 && ctx->have_unchanged) ? ctx->unchanged.as<uint8_t>() : nullptr,
                             blk ? ctx->run_lt_sp.as<uint32_t>() : ctx->run_lt.as<uint32_t>()},
                    tables_zero, ctx->sort_range, ctx->sort_range_n, what, chain ? row_base : (blk ? ctx->row_sp.as<uint32_t>() : nullptr), chain_zero, blk);
    };
    if (!bound_j && n > 0) {
        runs(1);
        HIPCHECK(hipGetLastError());
        bool scanned = false;
        const uint32_t nt = runs_count_tiles(n, &scanned);
        uint64_t total = 0;
        if (scanned) { int rc = read_info(ctx); if (rc) return rc; total = ctx->h_info->n_runs; }
        else {
            std::vector<uint32_t> counts(nt);
            HIPCHECK(hipMemcpyAsync(counts.data(), ctx->runs_scratch.p, (size_t)nt * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipStreamSynchronize(ctx->stream));
            for (uint32_t c : counts) total += c;
        }
        cap = std::max<size_t>(std::min<uint64_t>(total, n), 1);
    }
    HIPCHECK(ctx->records.ensure(cap * sizeof(TileRecord)));
    HIPCHECK(ctx->rk_u.ensure((chain ? 1 : cap) * 8));                  // (the global run sort's keys)
    HIPCHECK(ctx->run_lt.ensure(cap * 4));
    if (blk) {                                                          // (the sparse arrays: indexed like the segments)
        HIPCHECK(ctx->rec_sp.ensure(std::max<size_t>(n, 1) * sizeof(TileRecord)));
        HIPCHECK(ctx->run_lt_sp.ensure(std::max<size_t>(n, 1) * 4));
        HIPCHECK(ctx->row_sp.ensure(((size_t)tiles_h + 1) * 4));
    }
    runs(bound_j || n == 0 ? 3 : 2);
    ctx->sort_range = nullptr;
    ctx->pending_masks = PendingMasks{nullptr, 0u};
    HIPCHECK(hipGetLastError());
    if (!bound_j) {
        if (n > 0) {
            HIPCHECK(hipMemcpyAsync(ctx->h_rows, row_count, (size_t)tiles_h * 4, hipMemcpyDeviceToHost, ctx->stream));
            int rc = read_info(ctx);
            if (rc) return rc;
            if ((rc = verify_speculation(ctx))) return rc;
            J = ctx->h_info->n_runs;
            uint32_t mx = 0;
            for (uint32_t r = crow0; r < crow1; r++) mx = std::max(mx, ctx->h_rows[r]);      // (rows that are not painted get no workgroup)
            ctx->pred_max_row = mx;
            local_sort = local_sort && mx <= carry_rows_local_cap();
            n_slices = slices_for(256u, mx);
        }
        jc = DevCount{nullptr, J};
    }
    bool fold_equals_paint = true;
    for (int i = 0; i < 4; i++) {
        uint8_t sel = a.channels[i];
        if (sel == FORMA_CH_ALPHA && a.clear[3] == 1.0f) sel = FORMA_CH_ONE;          // renderer.rs:85-92
        const bool colour = sel == FORMA_CH_RED || sel == FORMA_CH_GREEN || sel == FORMA_CH_BLUE;
        if (i < 3 ? sel == FORMA_CH_ALPHA : colour) fold_equals_paint = false;
    }
    SpanGroups groups{nullptr, nullptr, 0u, 0u};
    const uint32_t n_groups = (tiles_w + SPAN_GROUP_TILES - 1u) >> SPAN_GROUP_SHIFT;
    const uint64_t* sorted_keys = ctx->rk_u.as<uint64_t>();
    if (jc.bound > 0) {
        const size_t jb = jc.bound;
        HIPCHECK(ctx->span_key.ensure(jb * 8));
        HIPCHECK(ctx->span_cov.ensure(jb * 16));
        // (a frame whose rows hold few spans skips them altogether — the painters would load a table to learn "none": +10 us
        //  on the 8192 x 8192 triangle scene; the first frame of a geometry does not know and goes without)
        if (!ctx->no_span_groups && (ctx->force_span_groups || ctx->pred_row_spans > SPAN_GROUP_MIN_ROW)) {
            // the row's spans again by tile-column group: a pool of two entries per run (a span has a run to its left, and spans
            // longer than a group are the exception); a slice of a row that does not fit keeps only its row list
            const size_t pool = std::min<size_t>(2 * jb, 0xFFFFFFFFu);
            HIPCHECK(ctx->grp_tab.ensure((size_t)tiles_h * CR_MAX_SLICES_HOST * n_groups * sizeof(uint2)));
            HIPCHECK(ctx->grp_list.ensure(pool * sizeof(uint4)));
            groups = SpanGroups{ctx->grp_tab.as<uint2>(), ctx->grp_list.as<uint4>(), (uint32_t)pool, ctx->force_span_groups ? 0u : SPAN_GROUP_MIN_ROW};
        }
        if (!local_sort) {
            HIPCHECK(ctx->rk_a.ensure(jb * 8));
            HIPCHECK(ctx->rk_b.ensure(jb * 8));
            HIPCHECK(ctx->sort_counters.ensure(sort_scratch_words(jb) * 4));
            // (tile_y, layer) order: stable radix sort on bits 32..63 = [layer 21 | tile_y+1 11]; live bits come from the
            // rasterizer's varying-bit mask (layer = key bits 0..20, tile_y = key bits 33..43)
            uint64_t live = ((ctx->live44 & 0x1FFFFFull) | ((ctx->live44 >> 33) << 21)) << 32;
            const SortPlan rk_plan = make_sort_plan(live, 32, 64, ctx->digit_bits);
            sorted_keys = launch_radix_sort(ctx->stream, ctx->rk_u.as<uint64_t>(), ctx->rk_a.as<uint64_t>(),
                                            ctx->rk_b.as<uint64_t>(), jc, rk_plan, ctx->digit_bits,
                                            ctx->sort_counters.as<uint32_t>(), &dinfo->error, nullptr, nullptr);
            ctx->pz.sort_p = nullptr;
        }
        if (ctx->force_slices) n_slices = ctx->force_slices;  // FORMA_HIP_DEBUG=carry_slices (tests: every slice count on one GPU)
        ctx->cur_slices = n_slices; ctx->cur_small = small; ctx->cur_half = half;
        // COVL (paint.hip): a row in ONE slice whose runs fit carry_rows_covl_cap() — the 4K scene's 4 800 — stages its cover sums
        // and style summaries in LDS before the walk.  A read-back-free frame guesses from the last verified frame's heaviest row
        // (6 % head-room); a row beyond the cap voids the frame and the synchronous re-run knows the rows.
        {
            // (the 512-lane variant for light rows in one slice — the 8K triangle scene — has the room as well: 32 KB of covers, two
            //  workgroups per CU instead of three, still one round for 512 rows)
            covl = covl_choice(n_slices);
            ctx->covl_tried = covl && bound_j != 0;
        }
    } else {
        HIPCHECK(ctx->span_key.ensure(8));
        HIPCHECK(ctx->span_cov.ensure(16));
    }
    ctx->image_sent = false;
    uint32_t bin_shift = 0;                               // 256 layer bins over the orders in use
    while (bin_shift < 16 && (((uint64_t)std::max<size_t>(ctx->n_orders, 1) - 1) >> bin_shift) > 255) bin_shift++;

    PaintParams P;
    P.width = a.width; P.height = a.height; P.tiles_w = tiles_w; P.tiles_h = tiles_h;
    P.crop_x0 = 0; P.crop_x1 = tiles_w; P.crop_y0 = 0; P.crop_y1 = tiles_h;
    if (a.crop) {                                         // Rect::new, renderer.rs:43-52
        P.crop_x0 = a.crop->x0 / 16; P.crop_x1 = std::min(tiles_w, (a.crop->x1 + 15) / 16);
        P.crop_y0 = a.crop->y0 / 16; P.crop_y1 = std::min(tiles_h, (a.crop->y1 + 15) / 16);
    }
    uint8_t ch[4] = {a.channels[0], a.channels[1], a.channels[2], a.channels[3]};
    if (a.clear[3] == 1.0f)                               // renderer.rs:85-92: Alpha -> One when clear is opaque
        for (int i = 0; i < 4; i++) if (ch[i] == FORMA_CH_ALPHA) ch[i] = FORMA_CH_ONE;
    P.channels = (uint32_t)ch[0] | ((uint32_t)ch[1] << 8) | ((uint32_t)ch[2] << 16) | ((uint32_t)ch[3] << 24);
    for (int i = 0; i < 4; i++) P.clear[i] = a.clear[i];
    P.stride_px = a.width; P.scene_has_clips = ctx->scene_has_clips ? 1u : 0u; P.scene_simple = ctx->scene_simple ? 1u : 0u; P.n_orders = (uint32_t)ctx->n_orders; P.n_words = (uint32_t)ctx->n_words;
    P.clear_unchanged = clear_unchanged;
    P.n_slices = jc.bound > 0 ? n_slices : 1u;             // (no runs: the carry pre-pass did not run, the zeroed tables say "no spans")
    P.n_groups = n_groups;
    // occlusion culling (PaintParams::cull): not with a buffer-layer cache (a tile's layer count is state there), not with clips
    // ... and only for a geometry whose tiles have been seen to overflow the wave painter's lists (k_paint_deep ran): that is where
    // culling pays — 1080p cubics: painter 85 -> 24 us — while a scene of moderate lists (the 4K stand-in: 39 entries per tile,
    // phases bound by their dependent steps, not by the entries) only pays for the occluder scan, ~1 % of its frames/s.
    const bool cull = a.cache_id < 0 && !ctx->scene_has_clips && !ctx->dbg.no_cull && (ctx->cull_on || ctx->dbg.force_cull);
    P.cull = cull ? 1u : 0u;
    // heaviest tiles first (PaintParams::order_*): read-back-free frames without a cache, one wavefront per tile
    P.order_cnt_in = nullptr; P.order_list_in = nullptr; P.order_cnt_out = nullptr; P.order_list_out = nullptr;
    P.row_base = (chain || blk) ? row_base : nullptr; P.row_cnt = (chain || blk) ? row_count : nullptr;
    ctx->order_pending = -1; ctx->order_cnt_dev = nullptr; ctx->order_keep_dev = nullptr;
    const uint32_t tiles_painted = (P.crop_y1 > P.crop_y0 ? P.crop_y1 - P.crop_y0 : 0u) * tiles_w;
    const bool strips = paint_by_strips(ctx, tiles_painted);
    const bool quads = paint_by_quads(ctx, a.cache_id, chain ? bound_j : jc.bound, tiles_painted);   // (k_paint_quad ignores order_*)
    P.order_flag_in = nullptr; P.order_flag_out = nullptr; P.order_hcap = 0; P.order_thr = 0;
    if (ctx->dbg.order_thr >= 0) ctx->order_off = 0;
    if (ctx->order_off) ctx->order_off--;                 // (a flat scene: the order is retried every 256 frames)
    // ... of launches that are a handful of rounds of wavefronts: one tile's life is then a good part of the launch's.  A frame
    // of 32 rounds (the 8K scene: 262 144 tiles on 8 192 wave slots) has no tail worth 10 % of bookkeeping.
    // a frame that leaves in bands (split_plan below) paints in several launches: no heavy-first order across them
    const int split_n = split_plan(ctx, a, P, bound_j, timing);
    if (ctx->order_enable && bound_j && a.cache_id < 0 && !strips && !quads && !ctx->dbg.no_order && jc.bound > 0 && tiles_painted && !ctx->order_off && split_n == 0 &&
        tiles_painted <= 16u * paint_strip_tiles(ctx) && (one_frame_in_flight(ctx) || ctx->dbg.order_thr >= 0)) {
        // (heavy section: an eighth of the band's tiles, as PAINT_ORDER_SUBS lists of equal capacity)
        const size_t per = paint_band_tiles(P.crop_y1 > P.crop_y0 ? P.crop_y1 - P.crop_y0 : 0u, tiles_w), hcap = std::max<size_t>((per / 8 + PAINT_ORDER_SUBS - 1) / PAINT_ORDER_SUBS, 2) * PAINT_ORDER_SUBS;
        const size_t set_words = PAINT_ORDER_WORDS + 8 * hcap + (8 * per + 3) / 4;      // counts | lists | one flag byte per tile
        if (ctx->order_buf.cap < 2 * set_words * 4) {
            HIPCHECK(ctx->order_buf.ensure(2 * set_words * 4));
            HIPCHECK(hipMemsetAsync(ctx->order_buf.p, 0, ctx->order_buf.cap, ctx->stream));   // (flag bytes nobody has written yet say "not heavy")
            ctx->order_cur = -1;
        }
        const forma_hip_ctx::OrderSig sig{tiles_w, tiles_h, P.crop_x0, P.crop_x1, P.crop_y0, P.crop_y1};
        uint32_t* base = ctx->order_buf.as<uint32_t>();
        if (ctx->order_cur >= 0 && !(ctx->order_sig == sig)) ctx->order_cur = -1;
        const int w = ctx->order_cur == 0 ? 1 : 0;
        if (ctx->order_cur >= 0) {
            P.order_cnt_in = base + (size_t)ctx->order_cur * set_words; P.order_list_in = P.order_cnt_in + PAINT_ORDER_WORDS;
            P.order_flag_in = reinterpret_cast<const uint8_t*>(P.order_list_in + 8 * hcap);
        }
        uint32_t* wset = base + (size_t)w * set_words;
        P.order_cnt_out = order_cnt; P.order_list_out = wset + PAINT_ORDER_WORDS;
        P.order_flag_out = reinterpret_cast<uint8_t*>(wset + PAINT_ORDER_WORDS + 8 * hcap);
        P.order_hcap = (uint32_t)hcap; P.order_thr = ctx->dbg.order_thr >= 0 ? (uint32_t)ctx->dbg.order_thr : ctx->order_thr;
        ctx->order_pending = w; ctx->order_pending_sig = sig; ctx->order_tiles = tiles_painted;
        ctx->order_cnt_dev = order_cnt; ctx->order_keep_dev = wset;
    } else ctx->order_cur = -1;                              // (any other frame in between: the lists are stale)
    // the (empty) k_paint_deep launch costs ~5 us of every frame: a read-back-free frame without a cache skips it when the
    // last verified frame had no deep tile; a tile that needs it then voids the frame (re-run in full)
    const bool launch_deep = !(bound_j != 0 && a.cache_id < 0 && ctx->pred_no_deep);
    if (jc.bound > 0)
        launch_carry_rows(ctx->stream, local_sort, small, half, n_slices, bin_shift, sorted_keys, ctx->records.as<TileRecord>(),
                          ctx->blk_edge.as<BlkEdge>(), nc, jc, ctx->layer_sf.as<uint32_t>(),
                          (uint32_t)ctx->n_orders, tiles_w, tiles_h, row_count, row_span_lo,
                          row_span_cnt, ctx->span_key.as<uint64_t>(), ctx->span_cov.as<uint4>(),
                          (a.cache_id >= 0 && ctx->have_unchanged) ? ctx->unchanged.as<uint8_t>() : nullptr, dinfo,
                          runs_edge_segments(),
                          // invisible carries of a partial last tile row are dropped only when nothing can observe them: with a
                          // buffer-layer cache the layer count of a tile is state (passes/tile_unchanged.rs)
                          // ... nor when the channel order tells a folded tile from a painted one: the solid fold encodes output
                          // bytes 0..2 as sRGB and passes byte 3 through (to_srgb_bytes of the SELECTED channels, painter/mod.rs:
                          // 156-162, 692), a painted tile encodes r, g, b and then selects (compute_srgb, :466-483) — the same bytes
                          // unless alpha lands in bytes 0..2 or a colour in byte 3, and an invisible layer can block the fold
                          (a.cache_id < 0 && (a.height & 15u) && fold_equals_paint) ? (a.height & 15u) : 16u, crow0, crow1, groups, ctx->run_lt.as<uint32_t>(),
                          cull,
                          (a.cache_id >= 0 || !fold_equals_paint) ? (a.crop ? a.crop->x0 / 16 : 0u) : 0xFFFFFFFFu,
                          chain ? row_base : nullptr, covl,
                          blk ? BlkRuns{ctx->rec_sp.as<TileRecord>(), ctx->run_lt_sp.as<uint32_t>(), ctx->runs_scratch.as<uint32_t>(),
                                        ctx->row_sp.as<uint32_t>(), row_base, tile_first_run, ctx->run_lt.as<uint32_t>(),
                                        (uint32_t)ctx->dbg.blk_round}
                              : BlkRuns{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 256u});
    stage_end(ctx, ST_CARRY, timing);
    stage_begin(ctx, ST_PAINT, timing);
    ctx->split_n = 0;
    if (split_n > 1 && !launch_deep) {
        // the painter in bands of tile rows, an event behind each: render_on sends the bands' pixels out behind those events
        const uint32_t rows = P.crop_y1 - P.crop_y0;
        for (int k = 0; k <= split_n; k++) ctx->split_row[k] = P.crop_y0 + (uint32_t)((uint64_t)rows * (uint32_t)k / (uint32_t)split_n);
        // the policy's two bands: the first a quarter of the rows (split_first) — its copy must last until the rest is painted, or
        // the link idles between the two (10 %: slower than one launch; 22-30 %: the plateau; profiles/r06_experiments.txt, r8g)
        if (split_n == 2) ctx->split_row[1] = P.crop_y0 + std::min(std::max(rows * (uint32_t)ctx->dbg.split_first / 100u, 1u), rows - 1u);
        for (int k = 0; k < split_n; k++) {
            PaintParams Pk = P;
            Pk.crop_y0 = ctx->split_row[k]; Pk.crop_y1 = ctx->split_row[k + 1];
            launch_paint(ctx->stream, Pk, ctx->sorted, ctx->records.as<TileRecord>(), jc, tile_first_run, row_span_lo,
                         row_span_cnt, ctx->span_key.as<uint64_t>(), ctx->span_cov.as<uint4>(), ctx->layer_col.as<uint4>(),
                         ctx->style_off.as<uint32_t>(),
                         ctx->style_words.as<uint32_t>(), ctx->images.as<forma_image_t>(), ctx->texels.as<uint16_t>(),
                         ctx->cur_image, tc, dinfo, paint_overflow, overflow_list, over2_n, over2_list, false, groups,
                         strips, quads, mid_n, mid_list, ctx->n_cus);
            HIPCHECK(hipEventRecord(ctx->split_ev[k], ctx->stream));
        }
        ctx->split_n = split_n;
    } else
    launch_paint(ctx->stream, P, ctx->sorted, ctx->records.as<TileRecord>(), jc, tile_first_run, row_span_lo,
                 row_span_cnt, ctx->span_key.as<uint64_t>(), ctx->span_cov.as<uint4>(), ctx->layer_col.as<uint4>(),
                 ctx->style_off.as<uint32_t>(),
                 ctx->style_words.as<uint32_t>(), ctx->images.as<forma_image_t>(), ctx->texels.as<uint16_t>(),
                 ctx->cur_image, tc, dinfo, paint_overflow, overflow_list, over2_n, over2_list, launch_deep, groups,
                 strips, quads, mid_n, mid_list, ctx->n_cus);
    stage_end(ctx, ST_PAINT, timing);
    ctx->last_runs = J; ctx->last_entries = 0;
    HIPCHECK(hipGetLastError());
    // what a later launch_paint_huge needs (tiles deeper than the painter's LDS lists: finish_paint)
    ctx->huge = forma_hip_ctx::HugeArgs{P, jc, tc, tile_first_run, row_span_lo, row_span_cnt, over2_n, over2_list, T};
    return FORMA_OK;
}

// Tiles whose layer list exceeds the painter's 4096-entry LDS lists were only RECORDED by k_paint_deep ({tile, entries},
// info->error bit 3).  The reference has no limit on the layers of a tile (layer_workbench/mod.rs:250-278): paint them now
// with lists in global memory sized from the recorded counts.  Called with the frame's FrameInfo in h_info, before anything
// of the image is copied out.  Rare (thousands of layers in one 16 x 16 tile), so the extra round trip does not matter.
int finish_paint(forma_hip_ctx* ctx) {
    if (!(ctx->h_info->error & 8u)) return FORMA_OK;
    const forma_hip_ctx::HugeArgs& h = ctx->huge;
    uint32_t n = 0;
    HIPCHECK(hipMemcpyAsync(&n, h.over2_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    n = std::min(n, h.T);
    std::vector<uint32_t> list((size_t)2 * n);
    if (n) HIPCHECK(hipMemcpy(list.data(), h.over2_list, (size_t)8 * n, hipMemcpyDeviceToHost));
    std::vector<uint64_t> offs(n);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) { offs[i] = total; total += list[2 * i + 1]; }
    // (32 bytes of list per entry: 2^27 entries are 4 GB of scratch — beyond that the frame is refused, not attempted)
    if (total >= (1ull << 27)) return fail(ctx, FORMA_E_CAPACITY, "more than 2^27 layer-list entries in the deep tiles of one frame");
    HIPCHECK(ctx->huge_offs.ensure(std::max<size_t>(n, 1) * 8));
    HIPCHECK(ctx->huge_key.ensure(std::max<uint64_t>(total, 1) * 4 * 8));
    HIPCHECK(ctx->huge_tmp.ensure(std::max<uint64_t>(total, 1) * 8));
    HIPCHECK(ctx->huge_flag.ensure(std::max<uint64_t>(total, 1) * 4));
    if (n) HIPCHECK(hipMemcpyAsync(ctx->huge_offs.p, offs.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    // (a read-back-free frame ended with k_frame_tail: the device-side run count is back to zero — the host copy has it)
    const DevCount jc = h.jc.ptr ? DevCount{nullptr, std::min(ctx->h_info->n_runs, h.jc.bound)} : h.jc;
    launch_paint_huge(ctx->stream, h.P, ctx->sorted, ctx->records.as<TileRecord>(), jc, h.tile_first_run, h.row_span_lo, h.row_span_cnt,
                      ctx->span_key.as<uint64_t>(), ctx->span_cov.as<uint4>(), ctx->layer_col.as<uint4>(), ctx->style_off.as<uint32_t>(),
                      ctx->style_words.as<uint32_t>(), ctx->images.as<forma_image_t>(), ctx->texels.as<uint16_t>(), ctx->cur_image, h.tc,
                      ctx->info.as<FrameInfo>(), h.over2_list, n, ctx->huge_offs.as<uint64_t>(), ctx->huge_key.as<uint64_t>(),
                      ctx->huge_tmp.as<uint64_t>(), ctx->huge_flag.as<uint32_t>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(ctx->stream));           // (`offs` is host memory of this call)
    // what the huge pass itself reported lands in the frame's host copy: finish_frame looks at fresh error bits
    FrameInfo after;
    HIPCHECK(hipMemcpy(&after, ctx->info.p, sizeof after, hipMemcpyDeviceToHost));
    ctx->info_clean = false;
    ctx->h_info->error = (ctx->h_info->error & ~8u) | (after.error & ~8u);
    ctx->h_info->plan_bad |= after.plan_bad;
    return FORMA_OK;
}

// rows [py0, py1) x pixels [px0, px1) of the frame's device image -> the caller's buffer.  Whole rows without padding on
// either side are ONE linear copy (the DMA engine's best case); anything else is a pitched copy.
int copy_rows_out(forma_hip_ctx* ctx, hipStream_t s, uint8_t* dst, size_t stride, size_t px0, size_t px1, size_t py0, size_t py1, uint32_t width) {
    const size_t pitch = (size_t)width * 4;
    if (stride == pitch && px0 == 0 && px1 == width)
        HIPCHECK(hipMemcpyAsync(dst + py0 * stride, ctx->cur_image + py0 * pitch, (py1 - py0) * pitch, hipMemcpyDeviceToHost, s));
    else
        HIPCHECK(hipMemcpy2DAsync(dst + py0 * stride + px0 * 4, stride, ctx->cur_image + py0 * pitch + px0 * 4, pitch, (px1 - px0) * 4,
                                  py1 - py0, hipMemcpyDeviceToHost, s));
    return FORMA_OK;
}

// the bands of a split frame (run_paint): each leaves on copy_stream behind its painter launch's event.  Called when ALL of the
// frame's kernels are enqueued — a copy into pageable memory blocks the host until it is done.
int send_split_bands(forma_hip_ctx* ctx, uint8_t* dst, size_t stride, const PaintArgs& a) {
    if (ctx->split_n < 2 || !dst) return FORMA_OK;
    const uint32_t tiles_w = (a.width + 15) / 16;
    uint32_t tx0 = 0, tx1 = tiles_w;
    if (a.crop) { tx0 = a.crop->x0 / 16; tx1 = std::min(tiles_w, (a.crop->x1 + 15) / 16); }
    if (tx0 >= tx1) return FORMA_OK;
    const size_t px0 = (size_t)tx0 * 16, px1 = std::min<size_t>((size_t)tx1 * 16, a.width);
    ctx->split_sent = true;
    for (int k = 0; k < ctx->split_n; k++) {
        const size_t py0 = (size_t)ctx->split_row[k] * 16, py1 = std::min<size_t>((size_t)ctx->split_row[k + 1] * 16, a.height);
        HIPCHECK(hipStreamWaitEvent(ctx->copy_stream, ctx->split_ev[k], 0));
        if (py0 >= py1) continue;
        const int rc = copy_rows_out(ctx, ctx->copy_stream, dst, stride, px0, px1, py0, py1, a.width);
        if (rc) return rc;
    }
    ctx->image_sent = true;
    return FORMA_OK;
}
// ... and nothing touches `dst` (or repaints the device image) before they have landed
int settle_split(forma_hip_ctx* ctx) {
    if (!ctx->split_sent) return FORMA_OK;
    ctx->split_sent = false;
    HIPCHECK(hipStreamSynchronize(ctx->copy_stream));
    return FORMA_OK;
}

// Copy what the frame wrote into the caller's buffer — and nothing else: tiles outside the crop, and with a buffer-layer
// cache the tiles the painter skipped (TileWriteOp::None), keep whatever the caller's buffer holds (reference
// cpu/buffer/layout/mod.rs:264-295 writes tile by tile; forma/src/cpu/buffer/mod.rs doc test "skipped rendering").
int copy_image_out(forma_hip_ctx* ctx, uint8_t* dst, size_t stride, bool timing, const PaintArgs& a, bool already_there = false) {
    const uint32_t tiles_w = (a.width + 15) / 16, tiles_h = (a.height + 15) / 16;
    uint32_t tx0 = 0, tx1 = tiles_w, ty0 = 0, ty1 = tiles_h;
    if (a.crop) {
        tx0 = a.crop->x0 / 16; tx1 = std::min(tiles_w, (a.crop->x1 + 15) / 16);
        ty0 = a.crop->y0 / 16; ty1 = std::min(tiles_h, (a.crop->y1 + 15) / 16);
    }
    ctx->last_written = 0;
    ctx->lw_valid = true; ctx->lw_tiles_w = tiles_w; ctx->lw_tiles_h = tiles_h; ctx->lw_cache = a.cache_id >= 0; ctx->lw_flags_on_host = false;
    ctx->lw_tx0 = tx0; ctx->lw_tx1 = std::max(tx0, tx1); ctx->lw_ty0 = ty0; ctx->lw_ty1 = std::max(ty0, ty1);
    if (!dst) return FORMA_OK;
    if (tx0 >= tx1 || ty0 >= ty1) return FORMA_OK;
    ctx->last_written = (tx1 - tx0) * (ty1 - ty0);
    if (already_there) return FORMA_OK;                    // (a deferred frame into caller memory: the image left behind the frame's kernels)
    const size_t px0 = (size_t)tx0 * 16, px1 = std::min<size_t>((size_t)tx1 * 16, a.width);
    const size_t py0 = (size_t)ty0 * 16, py1 = std::min<size_t>((size_t)ty1 * 16, a.height);
    const size_t pitch = (size_t)a.width * 4;
    stage_begin(ctx, ST_D2H, timing);
    if (a.cache_id < 0) {
        { const int rc = copy_rows_out(ctx, ctx->stream, dst, stride, px0, px1, py0, py1, a.width); if (rc) return rc; }
        stage_end(ctx, ST_D2H, timing);
        return FORMA_OK;
    }
    // cache attached: which tiles were written?
    const size_t T = (size_t)tiles_w * tiles_h;
    if (ctx->h_written_cap < T) {
        if (ctx->h_written) (void)hipHostFree(ctx->h_written);
        ctx->h_written = nullptr; ctx->h_written_cap = 0;
        HIPCHECK(hipHostMalloc((void**)&ctx->h_written, T, hipHostMallocDefault));
        ctx->h_written_cap = T;
    }
    // While the flags travel, the device packs the written tiles' pixels (list order = row-major over the crop): a frame that
    // rewrote 1.5 % of a 4K canvas then moves 0.5 MB over PCIe instead of 33 MB.  More than a quarter of the crop written:
    // one strided copy of the crop through the staging image is cheaper than the per-tile scatter on the host.
    const size_t n_crop = (size_t)(tx1 - tx0) * (ty1 - ty0);
    const bool no_pack = ctx->dbg.no_packed_copy;                                                    // (A/B switch for tools/, tests)
    const uint32_t max_pack = no_pack ? 0u : (uint32_t)std::min<size_t>(std::max<size_t>(n_crop / 4, 1), 1u << 20);
    HIPCHECK(ctx->pack_list.ensure((n_crop + 1) * 4));
    HIPCHECK(ctx->pack_pix.ensure(std::max<size_t>((size_t)max_pack, 1) * 1024));
    launch_pack_written(ctx->stream, ctx->cache_written.as<uint8_t>(), tiles_w, tx0, tx1, ty0, ty1, ctx->pack_list.as<uint32_t>() + 1,
                        ctx->pack_list.as<uint32_t>(), max_pack, ctx->cur_image, a.width, a.height, ctx->pack_pix.as<uint32_t>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(ctx->h_written, ctx->cache_written.p, T, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    ctx->lw_flags_on_host = true;
    size_t n_written = 0;
    for (uint32_t ty = ty0; ty < ty1; ty++) for (uint32_t tx = tx0; tx < tx1; tx++) n_written += ctx->h_written[(size_t)ty * tiles_w + tx] ? 1 : 0;
    ctx->last_written = (uint32_t)n_written;
    if (n_written == n_crop) {                                         // everything was painted: one strided copy
        { const int rc = copy_rows_out(ctx, ctx->stream, dst, stride, px0, px1, py0, py1, a.width); if (rc) return rc; }
    } else if (n_written && n_written <= max_pack) {                   // the packed tiles, then each into its place
        const size_t bytes = n_written * 1024;
        if (ctx->h_stage_cap < bytes) {
            if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
            ctx->h_stage = nullptr; ctx->h_stage_cap = 0;
            HIPCHECK(hipHostMalloc((void**)&ctx->h_stage, std::max<size_t>(bytes, 1 << 20), hipHostMallocDefault));
            ctx->h_stage_cap = std::max<size_t>(bytes, 1 << 20);
        }
        HIPCHECK(hipMemcpyAsync(ctx->h_stage, ctx->pack_pix.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        size_t k = 0;
        for (uint32_t ty = ty0; ty < ty1; ty++)
            for (uint32_t tx = tx0; tx < tx1; tx++) {
                if (!ctx->h_written[(size_t)ty * tiles_w + tx]) continue;
                const size_t x0 = (size_t)tx * 16, x1 = std::min<size_t>(x0 + 16, a.width);
                const size_t y0 = (size_t)ty * 16, y1 = std::min<size_t>(y0 + 16, a.height);
                const uint8_t* src = ctx->h_stage + k * 1024;
                for (size_t y = y0; y < y1; y++) memcpy(dst + y * stride + x0 * 4, src + (y - y0) * 64, (x1 - x0) * 4);
                k++;
            }
    } else if (n_written) {                                            // stage, then copy only the written tiles
        const size_t bytes = pitch * a.height;
        if (ctx->h_stage_cap < bytes) {
            if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
            ctx->h_stage = nullptr; ctx->h_stage_cap = 0;
            HIPCHECK(hipHostMalloc((void**)&ctx->h_stage, bytes, hipHostMallocDefault));
            ctx->h_stage_cap = bytes;
        }
        HIPCHECK(hipMemcpy2DAsync(ctx->h_stage + py0 * pitch + px0 * 4, pitch, ctx->cur_image + py0 * pitch + px0 * 4, pitch,
                                  (px1 - px0) * 4, py1 - py0, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        for (uint32_t ty = ty0; ty < ty1; ty++)
            for (uint32_t tx = tx0; tx < tx1; tx++) {
                if (!ctx->h_written[(size_t)ty * tiles_w + tx]) continue;
                const size_t x0 = (size_t)tx * 16, x1 = std::min<size_t>(x0 + 16, a.width);
                const size_t y0 = (size_t)ty * 16, y1 = std::min<size_t>(y0 + 16, a.height);
                for (size_t y = y0; y < y1; y++) memcpy(dst + y * stride + x0 * 4, ctx->h_stage + y * pitch + x0 * 4, (x1 - x0) * 4);
            }
    }
    stage_end(ctx, ST_D2H, timing);
    return FORMA_OK;
}

int finish_frame(forma_hip_ctx* ctx, forma_timings_t* t, bool have_info = false) {
    if (!have_info) {
        HIPCHECK(hipMemcpyAsync(ctx->h_info, ctx->info.p, sizeof(FrameInfo), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
    }
    // device-side invariant flags
    ctx->pred_no_deep = !(ctx->h_info->error & 16u);       // (8, 16: bookkeeping bits, not errors)
    if (ctx->h_info->error & 16u) ctx->cull_on = true;     // deep tiles: from now on the painters cull (until the geometry changes)
    if (!ctx->h_info->plan_bad) {                          // what the keys' tile fields spanned (a sort that did not run leaves min > max)
        const uint32_t* r = ctx->h_info->tile_range;
        ctx->pred_range = KeyRange{~r[0], r[1], ~r[2], r[3], true};
    }
    if (!ctx->h_info->plan_bad) ctx->pred_row_spans = ctx->h_info->n_spans / ctx->cur_rows_painted;
    if (!ctx->h_info->plan_bad) { ctx->pred_max_slice = ctx->h_info->max_slice_runs; ctx->pred_slice_n = ctx->cur_slices; ctx->pred_slice_small = ctx->cur_small; ctx->pred_slice_half = ctx->cur_half; }
    if ((ctx->h_info->error & ~24u) == 1u)                   // (bit 0: k_carry_rows met a run of a layer without a style)
        return fail(ctx, FORMA_E_ARG, "a geometry entry names an order that has no style (forma_hip_set_styles: offset FORMA_NONE or beyond the table)");
    if (ctx->h_info->error & ~24u) return fail(ctx, FORMA_E_INTERNAL, "device-side invariant violated");
    if (!t) return FORMA_OK;
    memset(t, 0, sizeof *t);
    float* dstv[ST_COUNT] = {&t->prepare_us, &t->rasterize_us, &t->sort_us, &t->carry_us, &t->paint_us, &t->d2h_us, &t->exchange_us};
    // stage = the sum of its kernels' own durations; total = first kernel's start -> last kernel's end (gaps included)
    float pass = 0; int np = 0;
    KernelTimer& kt = ctx->kt;
    for (int i = 0; i < kt.n; i++) {
        float ms = 0;
        kt_us(ctx)[i] = 0.0f;
        if (hipEventElapsedTime(&ms, kt.e0[i], kt.e1[i]) != hipSuccess) continue;
        kt_us(ctx)[i] = ms * 1000.0f;
        if (kt.stage[i] >= 0 && kt.stage[i] < ST_COUNT) *dstv[kt.stage[i]] += ms * 1000.0f;
        if (strstr(kt.name[i], "k_onesweep")) { pass += ms * 1000.0f; np++; }   // (#kern text: "(k_onesweep<8>)" — any bracketing)
    }
    if (kt.n) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, kt.e0[0], kt.e1[kt.n - 1]) == hipSuccess) t->total_us = ms * 1000.0f;
        for (int i = 0; i < kt.n; i++) {
            ms = 0;
            ctx->kt_start_us[i] = hipEventElapsedTime(&ms, kt.e0[0], kt.e0[i]) == hipSuccess ? ms * 1000.0f : 0.0f;
        }
    }
    ctx->kt_n_done = kt.n;
    if (ctx->stage_used[ST_D2H]) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->ev0[ST_D2H], ctx->ev1[ST_D2H]) == hipSuccess) t->d2h_us = ms * 1000.0f;
    }
    t->sort_pass_us = np ? pass / np : 0.0f;
    t->n_lines = (uint32_t)ctx->n_lines; t->n_segments = (uint32_t)ctx->n_seg; t->n_sort_passes = (uint32_t)ctx->n_passes;
    t->n_runs = ctx->last_runs ? ctx->last_runs : ctx->h_info->n_runs; t->n_tile_entries = ctx->h_info->n_spans;
    t->n_tiles_written = ctx->last_written;
    return FORMA_OK;
}

void clear_stage_flags(forma_hip_ctx* ctx) {
    for (int s = 0; s < ST_COUNT; s++) ctx->stage_used[s] = false;
    ctx->kt.n = 0; ctx->kt.dropped = 0; g_ktimer = nullptr;
    ctx->order_cnt_dev = nullptr; ctx->order_keep_dev = nullptr;       // (a frame that never reached its k_frame_tail)
}

int check_paint_args(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride,
                     const uint8_t* channels, const float* clear) {
    int rc = check_canvas(ctx, width, height);
    if (rc) return rc;
    if (!channels || !clear) return fail(ctx, FORMA_E_ARG, "null channels/clear_color");
    for (int i = 0; i < 4; i++) if (channels[i] > FORMA_CH_ONE) return fail(ctx, FORMA_E_ARG, "invalid channel selector");
    if (dst && (size_t)width * 4 > stride) return fail(ctx, FORMA_E_ARG, "width exceeds width stride");   // layout/mod.rs:188-193
    return FORMA_OK;
}

template <class T>
int upload(forma_hip_ctx* ctx, DevBuf& b, const T* src, size_t n) {
    HIPCHECK(b.ensure(std::max<size_t>(n, 1) * sizeof(T)));
    if (n) HIPCHECK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return FORMA_OK;
}

}  // namespace

// entry points that work on the context's own frame buffers: frames in flight are finished first; on a multi-device context
// the stage entry points (host arrays in, host arrays out) run on its first device, the single-device frame plumbing is refused
#define ENTER_STAGE(ctx)                                                                        \
    do {                                                                                        \
        if (ctx->multi) ctx = multi_first(ctx);                                                 \
        else { const int _rc = fd_drain(ctx); if (_rc) return _rc; ctx->last = ctx; }           \
    } while (0)
#define ENTER_SINGLE(ctx)                                                                       \
    do {                                                                                        \
        if (!ctx) return FORMA_E_ARG;                                                           \
        if (ctx->multi) return fail(ctx, FORMA_E_STATE, "not available on a multi-device context (forma_hip_render does the whole frame)"); \
        const int _rc = fd_drain(ctx); if (_rc) return _rc; ctx->last = ctx;                    \
    } while (0)

extern "C" {

const char* forma_hip_version(void) { return "forma_hip 0.1.0 gfx950"; }

int forma_hip_create(forma_hip_ctx** out, int device) {
    if (!out) return FORMA_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FORMA_E_NO_DEVICE;
    if (device < 0 || device >= count) return FORMA_E_ARG;
    // the kernels are code objects of ONE architecture (no fat binary) — the Makefile's ARCH, gfx950: a device of any other is
    // "no device" (forma_hip.h).  The CU count sizes the policies that speak of "the chip" (sort_workgroups, paint_by_strips).
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, FORMA_ARCH, strlen(FORMA_ARCH)) != 0) return FORMA_E_NO_DEVICE;
    forma_hip_ctx* ctx = new (std::nothrow) forma_hip_ctx();
    if (!ctx) return FORMA_E_INTERNAL;
    ctx->device = device;
    ctx->n_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    ctx->dbg = forma_debug_parse();                       // FORMA_HIP_DEBUG (debug.h): test / tool switches, never set in deployment
    if (ctx->dbg.digit_bits == 4 || ctx->dbg.digit_bits == 8 || ctx->dbg.digit_bits == 9) ctx->digit_bits = ctx->dbg.digit_bits;
    ctx->no_async = ctx->dbg.sync;
    ctx->global_runsort = ctx->dbg.global_runsort;
    ctx->xgather_always = ctx->dbg.xgather;
    ctx->no_small_carry = ctx->dbg.no_small_carry;
    ctx->no_span_groups = ctx->dbg.no_span_groups;
    ctx->force_span_groups = ctx->dbg.span_groups;
    if (ctx->dbg.carry_slices > 0) ctx->force_slices = (uint32_t)std::min(std::max(ctx->dbg.carry_slices, 1), (int)CR_MAX_SLICES_HOST);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx; return FORMA_E_HIP;
    }
    // (coherent = fine-grained: the host polls the word behind the FrameInfo while the frame's kernels still run)
    bool ok = hipHostMalloc((void**)&ctx->h_info, sizeof(FrameInfo) + 64, hipHostMallocCoherent) == hipSuccess &&
              hipHostMalloc((void**)&ctx->h_rows, 2049 * 4, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void**)&ctx->h_xlocal, 2 * 4, hipHostMallocDefault) == hipSuccess &&
              ctx->info.ensure(sizeof(FrameInfo)) == hipSuccess && ctx->info_init.ensure(sizeof(FrameInfo)) == hipSuccess;
    if (ok) { ctx->h_seq = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->h_info) + ((sizeof(FrameInfo) + 15) & ~(size_t)15)); *ctx->h_seq = 0; }
    if (ok) {
        FrameInfo fi;
        memset(&fi, 0, sizeof fi);
        fi.key_and = 0xFFFFFFFFu; fi.key_and_hi = 0xFFFFFFFFu;
        ok = hipMemcpy(ctx->info_init.p, &fi, sizeof fi, hipMemcpyHostToDevice) == hipSuccess;
    }
    for (int s = 0; s < ST_COUNT && ok; s++) {
        ok = hipEventCreate(&ctx->ev0[s]) == hipSuccess && hipEventCreate(&ctx->ev1[s]) == hipSuccess;
        ctx->stage_used[s] = false;
    }
    if (!ok) { delete ctx; return FORMA_E_HIP; }
    // empty-scene defaults so that a render before any upload is well defined
    ok = ctx->style_off.ensure(4) == hipSuccess && ctx->style_words.ensure(4) == hipSuccess && ctx->layer_sf.ensure(4) == hipSuccess &&
         ctx->layer_col.ensure(16) == hipSuccess && ctx->geoms.ensure(sizeof(forma_geom_t)) == hipSuccess &&
         ctx->images.ensure(sizeof(forma_image_t)) == hipSuccess && ctx->texels.ensure(8) == hipSuccess &&
         ctx->x.ensure(4) == hipSuccess && ctx->y.ensure(4) == hipSuccess && ctx->line_slot.ensure(4) == hipSuccess;
    if (!ok) { forma_hip_destroy(ctx); return FORMA_E_HIP; }
    *out = ctx;
    return FORMA_OK;
}

int forma_hip_create_multi(forma_hip_ctx** out, const int* devices, int n) {
    if (!out || !devices || n < 1 || n > FORMA_MAX_RANKS) return FORMA_E_ARG;
    // one device: a plain context, unless FORMA_HIP_DEBUG=force_exchange asks for the exchange path with a world of one
    // (rehearsal on single-GPU machines: same planner, same workers, same collective calls)
    if (n == 1 && !forma_debug_parse().force_exchange) return forma_hip_create(out, devices[0]);
    return multi_create(out, devices, n);
}

void forma_hip_destroy(forma_hip_ctx* ctx) {
    if (!ctx) return;
    if (ctx->multi) { multi_destroy(ctx); return; }
    (void)hipSetDevice(ctx->device);
    (void)fd_drain(ctx);
    for (forma_hip_ctx* sl : ctx->slots) if (sl != ctx) { sl->owner = nullptr; forma_hip_destroy(sl); }
    ctx->slots.clear();
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* all[] = {&ctx->x, &ctx->y, &ctx->line_slot, &ctx->geoms, &ctx->style_off, &ctx->style_words, &ctx->layer_sf, &ctx->layer_col, &ctx->unchanged,
                     &ctx->images, &ctx->texels, &ctx->l_order, &ctx->l_x0, &ctx->l_y0, &ctx->l_dx, &ctx->l_dy, &ctx->l_a,
                     &ctx->l_b, &ctx->l_c, &ctx->l_d, &ctx->l_len, &ctx->scan_tmp, &ctx->cl_idx, &ctx->cl_start,
                     &ctx->block_first, &ctx->prep_scratch, &ctx->seg_u, &ctx->seg_a, &ctx->seg_b,
                     &ctx->sort_counters, &ctx->info, &ctx->info_init, &ctx->records, &ctx->rk_u, &ctx->rk_a, &ctx->rk_b,
                     &ctx->blk_edge, &ctx->runs_scratch, &ctx->row_tab, &ctx->span_key, &ctx->span_cov,
                     &ctx->image, &ctx->xsend, &ctx->xrecv, &ctx->xscratch,
                     &ctx->ras_masks, &ctx->xmask, &ctx->huge_offs, &ctx->huge_key, &ctx->huge_tmp, &ctx->huge_flag,
                     &ctx->grp_tab, &ctx->grp_list, &ctx->run_lt, &ctx->rec_sp, &ctx->run_lt_sp, &ctx->row_sp, &ctx->pack_list, &ctx->pack_pix, &ctx->cache_written, &ctx->order_buf};
    for (DevBuf* b : all) b->release();
    for (int s = 0; s < ST_COUNT; s++) { (void)hipEventDestroy(ctx->ev0[s]); (void)hipEventDestroy(ctx->ev1[s]); }
    if (g_ktimer == &ctx->kt) g_ktimer = nullptr;           // (a timed frame of this context that failed between stage_begin and stage_end)
    for (int i = 0; i < ctx->kt.made; i++) { (void)hipEventDestroy(ctx->kt.e0[i]); (void)hipEventDestroy(ctx->kt.e1[i]); }
    if (ctx->h_info) (void)hipHostFree(ctx->h_info);
    if (ctx->h_rows) (void)hipHostFree(ctx->h_rows);
    if (ctx->h_xlocal) (void)hipHostFree(ctx->h_xlocal);
    if (ctx->h_written) (void)hipHostFree(ctx->h_written);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    for (auto& c : ctx->caches) { c.tiles.release(); c.image.release(); }
    ctx->cache_written.release();
    for (auto& r : ctx->registered) (void)hipHostUnregister(r.first);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        for (int k = 0; k < forma_hip_ctx::SPLIT_MAX; k++) if (ctx->split_ev[k]) (void)hipEventDestroy(ctx->split_ev[k]);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* forma_hip_last_error(const forma_hip_ctx* ctx) { return ctx ? ctx->err : "null context"; }

// ---- scene upload --------------------------------------------------------------------------------
int forma_hip_set_geometry(forma_hip_ctx* ctx, const float* x, const float* y, const uint32_t* line_slot, size_t n_points) {
    if (!ctx) return FORMA_E_ARG;
    if (n_points && (!x || !y)) return fail(ctx, FORMA_E_ARG, "null geometry");
    if (n_points > 1 && !line_slot) return fail(ctx, FORMA_E_ARG, "null line_slot");
    if (n_points >= (1ull << 30)) return fail(ctx, FORMA_E_ARG, "too many points");
    if (ctx->multi) return multi_set_geometry(ctx, x, y, line_slot, n_points);
    HIPCHECK(hipSetDevice(ctx->device));
    int rc;
    if ((rc = fd_drain(ctx))) return rc;                  // frames in flight still read the old buffers
    if ((rc = upload(ctx, ctx->x, x, n_points))) return rc;
    if ((rc = upload(ctx, ctx->y, y, n_points))) return rc;
    if ((rc = upload(ctx, ctx->line_slot, line_slot, n_points ? n_points - 1 : 0))) return rc;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    ctx->n_points = n_points;
    invalidate_counts(ctx);                               // new geometry: the next frame re-learns N and J synchronously
    share_scene(ctx);
    return FORMA_OK;
}

int forma_hip_set_geoms(forma_hip_ctx* ctx, const forma_geom_t* geoms, size_t n_geoms) {
    if (!ctx || (n_geoms && !geoms)) return fail(ctx, FORMA_E_ARG, "null geoms");
    uint32_t max_order = 0;
    for (size_t i = 0; i < n_geoms; i++) {
        if (geoms[i].order == FORMA_NONE) continue;
        if (geoms[i].order > FORMA_LAYER_LIMIT)                                      // utils/order.rs:44-66
            return fail(ctx, FORMA_E_ARG, "order exceeds LAYER_LIMIT");
        max_order = std::max(max_order, geoms[i].order);
    }
    if (ctx->multi) return multi_set_geoms(ctx, geoms, n_geoms);
    HIPCHECK(hipSetDevice(ctx->device));
    int rc = fd_drain(ctx);
    if (rc) return rc;
    if ((rc = upload(ctx, ctx->geoms, geoms, n_geoms))) return rc;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    ctx->n_geoms = n_geoms; ctx->max_geom_order = max_order;
    share_scene(ctx);
    return FORMA_OK;
}

int forma_hip_set_styles(forma_hip_ctx* ctx, const uint32_t* style_offsets, size_t n_orders, const uint32_t* style_words,
                         size_t n_words, const uint8_t* unchanged) {
    if (!ctx || (n_orders && !style_offsets) || (n_words && !style_words)) return fail(ctx, FORMA_E_ARG, "null styles");
    if (n_orders > (size_t)FORMA_LAYER_LIMIT + 1) return fail(ctx, FORMA_E_ARG, "order exceeds LAYER_LIMIT");
    if (ctx->multi) return multi_set_styles(ctx, style_offsets, n_orders, style_words, n_words, unchanged);
    bool clips = false, simple = true;
    size_t costly = 0;                                    // layers with a gradient, texture, blend mode or clip (strip painters)
    // per order: what the carry pre-pass attaches to every run and span of the layer (one gather instead of a chain through
    // the offset table and the style words): SF_* flags and the four words the painter's fast paths read
    ctx->h_layer_sf.assign(n_orders, 0u);
    ctx->h_layer_col.assign(n_orders * 4, 0u);
    uint32_t max_image = 0; bool any_texture = false;
    for (size_t o = 0; o < n_orders; o++) {
        const size_t off = style_offsets[o];                 // size_t: `off + 2` must not wrap in 32 bits
        if (style_offsets[o] == FORMA_NONE) continue;
        if (off + 2 > n_words) return fail(ctx, FORMA_E_ARG, "style offset out of range");
        uint32_t h = style_words[off];
        size_t need = 2;
        if (!FORMA_STYLE_IS_CLIP(h)) {
            uint32_t ft = FORMA_STYLE_FILL(h);
            need = ft == FORMA_FILL_SOLID ? 6 : (ft == FORMA_FILL_TEXTURE ? 9 : 6 + 5 * (size_t)FORMA_STYLE_STOPS(h));
            if (ft == FORMA_FILL_LINEAR || ft == FORMA_FILL_RADIAL) if (FORMA_STYLE_STOPS(h) < 1) return fail(ctx, FORMA_E_ARG, "gradient without stops");
        }
        if (off + need > n_words) return fail(ctx, FORMA_E_ARG, "style payload out of range");
        if (!FORMA_STYLE_IS_CLIP(h) && FORMA_STYLE_FILL(h) == FORMA_FILL_TEXTURE) { any_texture = true; max_image = std::max(max_image, style_words[off + 8]); }
        if (FORMA_STYLE_IS_CLIP(h) || FORMA_STYLE_CLIPPED(h)) clips = true;
        if (FORMA_STYLE_IS_CLIP(h) || FORMA_STYLE_CLIPPED(h) || FORMA_STYLE_FILL(h) != FORMA_FILL_SOLID || FORMA_STYLE_BLEND(h) != 0u) simple = false;
        if (!FORMA_STYLE_IS_CLIP(h) && (FORMA_STYLE_FILL(h) != FORMA_FILL_SOLID || FORMA_STYLE_BLEND(h) != 0u || FORMA_STYLE_CLIPPED(h))) costly++;
        uint32_t sfl = (FORMA_STYLE_EVENODD(h) ? SF_EVENODD : 0u) | (FORMA_STYLE_BLEND(h) << SF_BLEND_SHIFT) | (FORMA_STYLE_FILL(h) << SF_FILL_SHIFT);
        uint32_t* col = &ctx->h_layer_col[o * 4];
        if (FORMA_STYLE_IS_CLIP(h)) { sfl |= SF_IS_CLIP; col[0] = style_words[off + 1]; }
        else {
            if (FORMA_STYLE_CLIPPED(h)) sfl |= SF_CLIPPED;
            for (int k = 0; k < 4; k++) col[k] = style_words[off + 2 + k];
            float alpha; memcpy(&alpha, &col[3], 4);
            if (FORMA_STYLE_FILL(h) == FORMA_FILL_SOLID && alpha == 1.0f) sfl |= SF_OPAQUE;
        }
        ctx->h_layer_sf[o] = sfl | LSF_VALID;
    }
    HIPCHECK(hipSetDevice(ctx->device));
    int rc;
    if ((rc = fd_drain(ctx))) return rc;
    if ((rc = upload(ctx, ctx->style_off, style_offsets, n_orders))) return rc;
    if ((rc = upload(ctx, ctx->style_words, style_words, n_words))) return rc;
    if ((rc = upload(ctx, ctx->layer_sf, ctx->h_layer_sf.data(), n_orders))) return rc;
    if ((rc = upload(ctx, ctx->layer_col, ctx->h_layer_col.data(), n_orders * 4))) return rc;
    if (unchanged && (rc = upload(ctx, ctx->unchanged, unchanged, n_orders))) return rc;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    ctx->n_orders = n_orders; ctx->n_words = n_words; ctx->scene_has_clips = clips; ctx->scene_simple = simple;
    ctx->costly_layers = costly;
    ctx->have_unchanged = unchanged != nullptr;
    ctx->any_texture = any_texture; ctx->max_image_index = max_image;
    share_scene(ctx);
    return FORMA_OK;
}

int forma_hip_set_images(forma_hip_ctx* ctx, const forma_image_t* images, size_t n_images, const uint16_t* texels,
                         size_t n_texels) {
    if (!ctx || (n_images && !images) || (n_texels && !texels)) return fail(ctx, FORMA_E_ARG, "null images");
    for (size_t i = 0; i < n_images; i++)
        if (images[i].texel_offset + (uint64_t)images[i].width * images[i].height > n_texels || images[i].width == 0 || images[i].height == 0)
            return fail(ctx, FORMA_E_ARG, "image outside texel pool");
    if (ctx->multi) return multi_set_images(ctx, images, n_images, texels, n_texels);
    HIPCHECK(hipSetDevice(ctx->device));
    int rc;
    if ((rc = fd_drain(ctx))) return rc;
    if ((rc = upload(ctx, ctx->images, images, n_images))) return rc;
    if ((rc = upload(ctx, ctx->texels, texels, n_texels * 4))) return rc;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    ctx->n_images = n_images;
    share_scene(ctx);
    return FORMA_OK;
}

// ---- stage 1 ---------------------------------------------------------------------------------------
int forma_hip_flatten(forma_hip_ctx* ctx, const forma_flatten_tables_t* t, float* out_x, float* out_y) {
    if (!ctx || !t || (t->n_points && (!out_x || !out_y))) return fail(ctx, FORMA_E_ARG, "null flatten tables");
    if (t->n_points == 0) return FORMA_OK;
    if (ctx->multi) ctx = multi_first(ctx);
    HIPCHECK(hipSetDevice(ctx->device));
    std::vector<void*> owned;
    auto up = [&](const void* src, size_t bytes, const void** dst) -> hipError_t {
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, std::max<size_t>(bytes, 4));
        if (e != hipSuccess) return e;
        owned.push_back(d);
        if (bytes) e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream);
        *dst = d;
        return e;
    };
    forma_flatten_tables_t d = *t;
    hipError_t e = hipSuccess;
    const size_t np = t->n_points, nq = t->n_quads, ns = t->n_splines;
#define UP(field, count, type) if (e == hipSuccess) e = up(t->field, (count) * sizeof(type), (const void**)&d.field)
    UP(point_commands, np, uint32_t); UP(point_indices, np, uint32_t); UP(quad_indices, np, uint32_t);
    UP(qx, 3 * nq, float); UP(qy, 3 * nq, float); UP(qw, 3 * nq, float);
    UP(x0, nq, float); UP(dx_recip, nq, float); UP(k0, nq, float); UP(dk, nq, float); UP(curvatures_recip, nq, float);
    UP(partial_spline, nq, uint32_t); UP(partial_curv, nq, float);
    UP(sp0x, ns, float); UP(sp0y, ns, float); UP(sp2x, ns, float); UP(sp2y, ns, float);
#undef UP
    float *dx = nullptr, *dy = nullptr;
    if (e == hipSuccess) e = hipMalloc((void**)&dx, np * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dy, np * 4);
    if (e == hipSuccess) {
        launch_flatten(ctx->stream, &d, dx, dy);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_x, dx, np * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_y, dy, np * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = e2;
    for (void* p : owned) (void)hipFree(p);
    if (dx) (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (e != hipSuccess) return fail(ctx, FORMA_E_HIP, "flatten", e);
    return FORMA_OK;
}

// ---- stage entry points ------------------------------------------------------------------------------
int forma_hip_prepare_lines(forma_hip_ctx* ctx, uint32_t width, uint32_t height, uint32_t* orders, float* x0, float* y0,
                            float* dx, float* dy, float* a, float* b, float* c, float* d, uint32_t* lengths) {
    if (!ctx) return FORMA_E_ARG;
    ENTER_STAGE(ctx);
    HIPCHECK(hipSetDevice(ctx->device));
    const size_t n = ctx->n_points ? ctx->n_points - 1 : 0;
    if (n && (!orders || !x0 || !y0 || !dx || !dy || !a || !b || !c || !d || !lengths)) return fail(ctx, FORMA_E_ARG, "null output");
    ctx->n_lines = n;
    if (n == 0) return FORMA_OK;
    int rc = reset_info(ctx);
    if (rc) return rc;
    DevBuf* lb[] = {&ctx->l_order, &ctx->l_x0, &ctx->l_y0, &ctx->l_dx, &ctx->l_dy, &ctx->l_a, &ctx->l_b, &ctx->l_c, &ctx->l_d, &ctx->l_len};
    for (DevBuf* bb : lb) HIPCHECK(bb->ensure(n * 4));
    HIPCHECK(ctx->scan_tmp.ensure(scan_tmp_words(std::max<size_t>(n, 1 << 16)) * 4));
    launch_prepare_lines(ctx->stream, ctx->x.as<float>(), ctx->y.as<float>(), ctx->line_slot.as<uint32_t>(), (uint32_t)n,
                         ctx->geoms.as<forma_geom_t>(), (uint32_t)ctx->n_geoms, (float)width, (float)height, -3.0e38f, 3.0e38f,
                         ctx->l_order.as<uint32_t>(), ctx->l_x0.as<float>(), ctx->l_y0.as<float>(), ctx->l_dx.as<float>(),
                         ctx->l_dy.as<float>(), ctx->l_a.as<float>(), ctx->l_b.as<float>(), ctx->l_c.as<float>(),
                         ctx->l_d.as<float>(), ctx->l_len.as<uint32_t>());
    launch_inclusive_scan_u32(ctx->stream, ctx->l_len.as<uint32_t>(), n, ctx->scan_tmp.as<uint32_t>(),
                              &ctx->info.as<FrameInfo>()->n_segments);
    HIPCHECK(hipGetLastError());
    void* outs[] = {orders, x0, y0, dx, dy, a, b, c, d, lengths};
    for (int i = 0; i < 10; i++) HIPCHECK(hipMemcpyAsync(outs[i], lb[i]->p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return FORMA_OK;
}

int forma_hip_rasterize(forma_hip_ctx* ctx, size_t n_lines, const uint32_t* orders, const float* x0, const float* y0,
                        const float* dx, const float* dy, const float* a, const float* b, const float* c, const float* d,
                        const uint32_t* lengths, uint64_t* out_segments, size_t capacity, size_t* out_n) {
    if (!ctx || !out_n) return FORMA_E_ARG;
    *out_n = 0;
    if (n_lines == 0) return FORMA_OK;
    if (!orders || !x0 || !y0 || !dx || !dy || !a || !b || !c || !d || !lengths) return fail(ctx, FORMA_E_ARG, "null line arrays");
    ENTER_STAGE(ctx);
    HIPCHECK(hipSetDevice(ctx->device));
    const size_t N = lengths[n_lines - 1];
    *out_n = N;
    if (N > capacity) return fail(ctx, FORMA_E_CAPACITY, "segment capacity too small");
    if (N == 0) return FORMA_OK;
    int rc = reset_info(ctx);
    if (rc) return rc;
    DevBuf* lb[] = {&ctx->l_order, &ctx->l_x0, &ctx->l_y0, &ctx->l_dx, &ctx->l_dy, &ctx->l_a, &ctx->l_b, &ctx->l_c, &ctx->l_d, &ctx->l_len};
    const void* ins[] = {orders, x0, y0, dx, dy, a, b, c, d, lengths};
    for (int i = 0; i < 10; i++) {
        HIPCHECK(lb[i]->ensure(n_lines * 4));
        HIPCHECK(hipMemcpyAsync(lb[i]->p, ins[i], n_lines * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    LineSource S;
    memset(&S, 0, sizeof S);
    S.sums = ctx->l_len.as<uint32_t>(); S.orders = ctx->l_order.as<uint32_t>();
    S.x0 = ctx->l_x0.as<float>(); S.y0 = ctx->l_y0.as<float>(); S.dx = ctx->l_dx.as<float>(); S.dy = ctx->l_dy.as<float>();
    S.a = ctx->l_a.as<float>(); S.b = ctx->l_b.as<float>(); S.c = ctx->l_c.as<float>(); S.d = ctx->l_d.as<float>();
    if ((rc = run_line_table(ctx, S, n_lines, false))) return rc;
    if (ctx->n_seg != N) return fail(ctx, FORMA_E_INTERNAL, "prefix sums disagree");
    HIPCHECK(ctx->seg_u.ensure((N + SEG_PAD) * 8));
    HIPCHECK(ctx->ras_masks.ensure((N / RAS_TILE + 2) * 32));
    launch_rasterize(ctx->stream, S, DevCount{nullptr, (uint32_t)ctx->n_compact}, DevCount{nullptr, (uint32_t)N}, ctx->cl_idx.as<uint32_t>(),
                     ctx->cl_start.as<uint32_t>(), ctx->block_first.as<uint32_t>(), ctx->seg_u.as<uint64_t>(),
                     ctx->info.as<FrameInfo>(), 0, 0, ctx->ras_masks.as<uint32_t>(), true);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(out_segments, ctx->seg_u.p, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return FORMA_OK;
}

static uint64_t host_live44(const uint64_t* v, size_t n) {
    uint64_t o = 0, a = ~0ull;
    for (size_t i = 0; i < n; i++) { o |= v[i]; a &= v[i]; }
    return ((o ^ a) >> 20) & 0xFFFFFFFFFFFull;
}

int forma_hip_sort(forma_hip_ctx* ctx, uint64_t* segments, size_t n, int digit_bits) {
    if (!ctx || (n && !segments)) return FORMA_E_ARG;
    if (digit_bits != 0 && digit_bits != 4 && digit_bits != 8 && digit_bits != 9) return fail(ctx, FORMA_E_ARG, "digit_bits must be 0, 4, 8 or 9");
    if (n > 0xFFFFFFF0ull) return fail(ctx, FORMA_E_ARG, "too many segments");   // u32 prefix sums, segment.rs:90-98
    if (n == 0) return FORMA_OK;
    ENTER_STAGE(ctx);
    HIPCHECK(hipSetDevice(ctx->device));
    HIPCHECK(ctx->seg_u.ensure((n + SEG_PAD) * 8));
    HIPCHECK(hipMemcpyAsync(ctx->seg_u.p, segments, n * 8, hipMemcpyHostToDevice, ctx->stream));
    ctx->live44 = host_live44(segments, n);
    ctx->layer_sorted = false;
    ctx->pz = forma_hip_ctx::PreZero();
    int rc = reset_info(ctx);
    if (rc) return rc;
    if ((rc = run_sort(ctx, ctx->seg_u.as<uint64_t>(), DevCount{nullptr, (uint32_t)n}, false, digit_bits))) return rc;
    HIPCHECK(hipMemcpyAsync(segments, ctx->sorted, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = read_info(ctx))) return rc;
    if (ctx->h_info->error) return fail(ctx, FORMA_E_INTERNAL, "device-side invariant violated (radix look-back)");
    return FORMA_OK;
}

int forma_hip_paint(forma_hip_ctx* ctx, const uint64_t* sorted_segments, size_t n, uint8_t* dst, uint32_t width,
                    uint32_t height, size_t stride_bytes, const uint8_t channels[4], const float clear_color[4],
                    const forma_rect_t* crop_or_null) {
    if (!ctx || (n && !sorted_segments)) return FORMA_E_ARG;
    int rc = check_paint_args(ctx, dst, width, height, stride_bytes, channels, clear_color);
    if (rc) return rc;
    ENTER_STAGE(ctx);
    HIPCHECK(hipSetDevice(ctx->device));
    clear_stage_flags(ctx);
    ctx->pz = forma_hip_ctx::PreZero();
    if ((rc = reset_info(ctx))) return rc;
    HIPCHECK(ctx->seg_a.ensure((std::max<size_t>(n, 1) + SEG_PAD) * 8));
    if (n) HIPCHECK(hipMemcpyAsync(ctx->seg_a.p, sorted_segments, n * 8, hipMemcpyHostToDevice, ctx->stream));
    ctx->sorted = ctx->seg_a.as<uint64_t>();
    ctx->n_seg = n; ctx->have_unsorted = false;
    ctx->live44 = n ? host_live44(sorted_segments, n) : 0;
    PaintArgs a{width, height, channels, clear_color, crop_or_null};
    if ((rc = run_paint(ctx, DevCount{nullptr, (uint32_t)n}, a, false))) return rc;
    if ((rc = read_info(ctx)) || (rc = finish_paint(ctx))) return rc;
    if ((rc = copy_image_out(ctx, dst, stride_bytes, false, a))) return rc;
    if (dst) HIPCHECK(hipStreamSynchronize(ctx->stream));
    return finish_frame(ctx, nullptr, true);
}

// ---- the frame ---------------------------------------------------------------------------------------
}  // extern "C"

namespace {

void frame_done(forma_hip_ctx* ctx, int rc, const PaintArgs& a) {      // renderer.rs:217-218: remember the clear colour in the cache
    if (rc == FORMA_OK && a.cache_id >= 0) { ctx->caches[a.cache_id].has_clear = true; memcpy(ctx->caches[a.cache_id].clear, a.clear, 16); }
}

// FORMA_HIP_POISON_FRAME=<byte> (tests, tools): every per-frame buffer is refilled with that byte when a frame starts.  Frames of
// a test re-render one scene, so a kernel that reads what THIS frame never wrote normally finds last frame's (identical,
// "right") data there; with the refill it finds the poison.  Scene uploads, caches and the scratch image (a cropped frame
// legitimately leaves the rest of it alone) are not touched.
int poison_frame_buffers(forma_hip_ctx* c) {
    if (c->dbg.poison_frame < 0) return FORMA_OK;
    forma_hip_ctx* ctx = c;
    const int byte = c->dbg.poison_frame;
    DevBuf* frame[] = {&c->scan_tmp, &c->cl_idx, &c->cl_start, &c->block_first, &c->prep_scratch, &c->seg_u, &c->seg_a, &c->seg_b,
                       &c->sort_counters, &c->records, &c->rk_u, &c->rk_a, &c->rk_b, &c->blk_edge, &c->runs_scratch, &c->row_tab,
                       &c->span_key, &c->span_cov, &c->ras_masks, &c->huge_offs, &c->huge_key, &c->huge_tmp, &c->huge_flag,
                       &c->grp_tab, &c->grp_list, &c->run_lt, &c->rec_sp, &c->run_lt_sp, &c->row_sp, &c->pack_list, &c->pack_pix};
    for (DevBuf* b : frame) if (b->p && !b->borrowed) HIPCHECK(hipMemsetAsync(b->p, byte, b->cap, c->stream));
    return FORMA_OK;
}

// A read-back-free frame, first half: everything is enqueued on the context's stream, nothing waits.  N, J and the sort
// plan are predicted from the previous frame (bounds with slack); device-side guards keep a wrong guess memory-safe.
int enqueue_async_frame(forma_hip_ctx* ctx, const PaintArgs& a, bool timing, uint32_t* bN_out, uint32_t* bJ_out) {
    const uint32_t bN = ctx->pred_N + ctx->pred_N / 16 + 4096, bJ = ctx->pred_J + ctx->pred_J / 16 + 4096;
    *bN_out = bN; *bJ_out = bJ;
    FrameInfo* dinfo = ctx->info.as<FrameInfo>();
    int rc;
    if ((rc = poison_frame_buffers(ctx))) return rc;
    // a frame of kernels only: what later stages need cleared is cleared by the first kernel, FrameInfo reaches the host
    // (and returns to its pristine state) through the last one
    ZeroJobs Z; forma_hip_ctx::PreZero cleared;
    if ((rc = plan_zero_jobs(ctx, a.width, a.height, bN, bN, &Z, &cleared))) return rc;
    if ((rc = run_rasterize_frame(ctx, a.width, a.height, timing, true, bN, &Z, &cleared, /*hist_too=*/true))) return rc;
    if ((rc = run_sort(ctx, ctx->seg_u.as<uint64_t>(), DevCount{&dinfo->n_segments, bN}, timing))) return rc;
    ctx->order_enable = true;                             // (this frame ends with k_frame_tail: the painters may leave their order lists)
    rc = run_paint(ctx, DevCount{&dinfo->n_segments, bN}, a, timing, bJ);
    ctx->order_enable = false;
    if (rc) return rc;
    return frame_tail(ctx, true, nullptr);
}

// ... second half: wait, verify.  FORMA_RETRY: a prediction failed, nothing of the frame may be used (the caller re-runs it
// synchronously).  The frame is verified BEFORE anything lands in caller memory: a mispredicted frame never shows in `dst`.
// The end of a read-back-free frame as the host sees it.  k_frame_tail writes the frame's number into pinned memory behind the
// FrameInfo (system-scope release): polling that word returns a few microseconds before hipStreamSynchronize would — the stream's
// completion signal takes the runtime's wake-up path — and the frame's kernels are all complete when its last one has stored.
// Timed frames wait for the stream: their events must have been recorded.  A frame that never arrives (a fault) falls back to
// the stream's own error after a bounded spin.
int wait_frame_tail(forma_hip_ctx* ctx, bool timing) {
    if (!timing && ctx->dbg.tail_poll && ctx->h_seq) {
        const uint32_t want = ctx->tail_seq;
        for (uint32_t spin = 0; spin < (1u << 21); spin++) {            // (~50 ms: a frame that long waits on the stream instead)
            if (__atomic_load_n(ctx->h_seq, __ATOMIC_ACQUIRE) == want) return FORMA_OK;
            __builtin_ia32_pause();
            if ((spin & 0xFFFFu) == 0xFFFFu && hipStreamQuery(ctx->stream) != hipErrorNotReady) break;   // done without the word, or failed
        }
    }
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return FORMA_OK;
}

int complete_async_frame(forma_hip_ctx* ctx, const PaintArgs& a, uint8_t* dst, size_t stride_bytes, bool timing,
                         forma_timings_t* timings, uint32_t bN, uint32_t bJ) {
    { const int wrc = wait_frame_tail(ctx, timing); if (wrc) return wrc; }
    const uint32_t N = ctx->h_info->n_segments, J = ctx->h_info->n_runs;
    ctx->n_seg = N; ctx->n_compact = ctx->h_info->n_compact; ctx->last_runs = J;
    const bool ok = !ctx->h_info->plan_bad && N <= bN && J <= bJ;
    if (!ok) {
        { const int src = settle_split(ctx); if (src) return src; }               // (a split frame's bands are on their way: the re-run writes `dst` again)
        if (ctx->small_tried && ctx->h_info->plan_bad) ctx->small_banned = true;   // (one cause of plan_bad: a slice beyond the small variant)
        if (ctx->covl_tried && ctx->h_info->plan_bad) ctx->covl_banned = true;     // (another: a row beyond the COVL carry variant's LDS)
        if (ctx->plan_biased && ctx->h_info->plan_bad) ban_bias(ctx);              // (another: a key outside the span the digits were planned for)
        ctx->pred_counts_valid = false;                   // the synchronous path re-learns everything
        ctx->order_cur = -1; ctx->order_pending = -1;
        clear_stage_flags(ctx);
        return FORMA_RETRY;
    }
    if (ctx->order_pending >= 0) {
        ctx->order_cur = ctx->order_pending; ctx->order_sig = ctx->order_pending_sig; ctx->order_pending = -1;
        // steer the threshold: the heavy section should hold the few percent of the tiles that make the launch's tail
        const uint32_t nh = ctx->h_info->n_heavy, nt = std::max(ctx->order_tiles, 1u);
        if (nh * 16u > nt) ctx->order_thr = std::min<uint32_t>(ctx->order_thr + ctx->order_thr / 4u, 1u << 24);        // > 6 %
        else if (nh * 50u < nt) ctx->order_thr = std::max<uint32_t>(ctx->order_thr - ctx->order_thr / 5u, 1u << 12);     // < 2 %
        // ... of a scene that HAS a tail: a tile is heavy from twice the average on (sampled: FrameInfo::cost_*)
        if (ctx->h_info->cost_n) {
            const uint64_t mean = ((uint64_t)ctx->h_info->cost_sum << 8) / ctx->h_info->cost_n;
            ctx->order_thr = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ctx->order_thr, 2 * mean), 1u << 24);
            // a FLAT scene (the 8K triangle scene: 262 144 tiles of ~21 k clocks, none beyond twice that) has no tail to hide, and
            // the bookkeeping of the order — a flag byte per tile, the empty heavy section — costs its painter 10 %: three such
            // frames in a row switch the order off for the next 256
            if (ctx->order_thr <= 2 * mean && nh * 100u < nt) { if (++ctx->order_flat >= 3) { ctx->order_off = 256; ctx->order_flat = 0; ctx->order_cur = -1; } }
            else ctx->order_flat = 0;
        }
    }
    ctx->pred_N = N; ctx->pred_J = J; ctx->pred_max_row = ctx->h_info->max_row_runs;
    if (ctx->bias_banned) ctx->bias_banned--;
    int rc;
    // (a deferred frame's image left speculatively behind its kernels; tiles that k_paint_huge paints only now: the crop is
    //  copied again)
    const bool in_place = ctx->image_sent && !(ctx->h_info->error & 8u);
    const bool was_split = ctx->split_sent;
    if ((rc = settle_split(ctx))) return rc;               // (a split frame: its bands have landed — or land before the crop is copied again)
    if ((rc = finish_paint(ctx))) return rc;
    if ((rc = copy_image_out(ctx, dst, stride_bytes, timing, a, in_place))) return rc;
    if (dst && !(was_split && in_place)) HIPCHECK(hipStreamSynchronize(ctx->stream));   // (split: the tail has run, the copy stream is drained)
    rc = finish_frame(ctx, timings, true);
    frame_done(ctx, rc, a);
    return rc;
}

// the synchronous frame (first frame of a scene, or a prediction failed): N, the key masks and J are read back
int render_sync(forma_hip_ctx* ctx, const PaintArgs& a, uint8_t* dst, size_t stride_bytes, bool timing, forma_timings_t* timings) {
    int rc;
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = poison_frame_buffers(ctx))) return rc;
        if ((rc = run_rasterize_frame(ctx, a.width, a.height, timing, /*speculate=*/attempt == 0))) return rc;
        if ((rc = run_sort(ctx, ctx->seg_u.as<uint64_t>(), DevCount{nullptr, (uint32_t)ctx->n_seg}, timing))) return rc;
        rc = run_paint(ctx, DevCount{nullptr, (uint32_t)ctx->n_seg}, a, timing);
        if (rc == FORMA_RETRY) { clear_stage_flags(ctx); continue; }
        if (rc) return rc;
        if ((rc = read_info(ctx)) || (rc = finish_paint(ctx))) return rc;
        if ((rc = copy_image_out(ctx, dst, stride_bytes, timing, a))) return rc;
        if (dst) HIPCHECK(hipStreamSynchronize(ctx->stream));
        rc = finish_frame(ctx, timings, true);
        if (rc == FORMA_OK) { ctx->pred_N = (uint32_t)ctx->n_seg; ctx->pred_J = ctx->last_runs; ctx->pred_counts_valid = true; }
        frame_done(ctx, rc, a);
        return rc;
    }
    return fail(ctx, FORMA_E_INTERNAL, "sort plan did not converge");
}

// one whole frame on one slot, returning when `dst` (if any) is written
int render_on(forma_hip_ctx* ctx, uint8_t* dst, const PaintArgs& a, size_t stride_bytes, forma_timings_t* timings) {
    const bool timing = timings != nullptr;
    clear_stage_flags(ctx);
    if (a.width != ctx->pred_w || a.height != ctx->pred_h) { ctx->pred_counts_valid = false; ctx->pred_w = a.width; ctx->pred_h = a.height; }
    if (ctx->pred_valid && ctx->pred_counts_valid && !ctx->no_async) {
        uint32_t bN, bJ;
        ctx->split_want = dst != nullptr && !timing;        // (a frame into caller memory: the painter may run in bands, run_paint)
        ctx->split_n = 0;
        int rc = enqueue_async_frame(ctx, a, timing, &bN, &bJ);
        ctx->split_want = false;
        if (rc) return rc;
        if ((rc = send_split_bands(ctx, dst, stride_bytes, a))) { (void)settle_split(ctx); return rc; }
        rc = complete_async_frame(ctx, a, dst, stride_bytes, timing, timings, bN, bJ);
        if (rc != FORMA_RETRY) return rc;
    }
    return render_sync(ctx, a, dst, stride_bytes, timing, timings);
}

// frames in flight: finish the frame a slot still owes (wait, verify, re-run synchronously if a prediction failed)
int settle_slot(forma_hip_ctx* sl) {
    if (!sl->pending) return FORMA_OK;
    sl->pending = false;
    forma_hip_ctx* ctx = sl;
    HIPCHECK(hipSetDevice(sl->device));
    const forma_hip_ctx::Deferred& d = sl->def;
    PaintArgs a{d.width, d.height, d.channels, d.clear, d.has_crop ? &d.crop : nullptr, -1};
    int rc = complete_async_frame(sl, a, d.dst, d.stride, false, nullptr, d.bN, d.bJ);
    if (rc == FORMA_RETRY) rc = render_sync(sl, a, d.dst, d.stride, false, nullptr);
    return rc;
}

}  // namespace

// every frame the context still owes is finished; the first error of a deferred frame is reported here (and kept as the
// owner's last error text)
int fd_drain(forma_hip_ctx* ctx) {
    int first = FORMA_OK;
    for (forma_hip_ctx* sl : ctx->slots) {
        const int rc = settle_slot(sl);
        if (rc && !first) { first = rc; if (sl != ctx) memcpy(ctx->err, sl->err, sizeof ctx->err); }
    }
    return first;
}

namespace {

// slots[1..] see the owner's scene through borrowed buffers: refresh the views and the scalars after every upload
void share_scene(forma_hip_ctx* o) {
    for (forma_hip_ctx* sl : o->slots) {
        if (sl == o) continue;
        sl->x.borrow(o->x); sl->y.borrow(o->y); sl->line_slot.borrow(o->line_slot); sl->geoms.borrow(o->geoms);
        sl->style_off.borrow(o->style_off); sl->style_words.borrow(o->style_words); sl->unchanged.borrow(o->unchanged);
        sl->images.borrow(o->images); sl->texels.borrow(o->texels); sl->layer_sf.borrow(o->layer_sf); sl->layer_col.borrow(o->layer_col);
        sl->n_points = o->n_points; sl->n_geoms = o->n_geoms; sl->n_orders = o->n_orders; sl->n_words = o->n_words; sl->n_images = o->n_images;
        sl->max_geom_order = o->max_geom_order; sl->max_image_index = o->max_image_index; sl->any_texture = o->any_texture;
        sl->scene_has_clips = o->scene_has_clips; sl->scene_simple = o->scene_simple; sl->costly_layers = o->costly_layers; sl->have_unchanged = o->have_unchanged;
        sl->band_row0 = o->band_row0; sl->band_row1 = o->band_row1;
        sl->line_ranged = o->line_ranged; sl->line_lo = o->line_lo; sl->line_hi = o->line_hi;
    }
}
void invalidate_counts(forma_hip_ctx* o) {                 // new geometry / band: every slot re-learns N and J synchronously
    o->pred_counts_valid = false; o->xpred_valid = false; o->small_banned = false; o->covl_banned = false; o->bias_banned = 0; o->bias_ban_len = 0; o->pred_range.valid = false;
    o->order_off = 0; o->order_flat = 0; o->order_cur = -1; o->cull_on = false;
    for (forma_hip_ctx* sl : o->slots) { sl->order_off = 0; sl->order_flat = 0; sl->order_cur = -1; sl->cull_on = false; }
    for (forma_hip_ctx* sl : o->slots) { sl->pred_counts_valid = false; sl->xpred_valid = false; sl->small_banned = false; sl->covl_banned = false; sl->bias_banned = 0; sl->bias_ban_len = 0; sl->pred_range.valid = false; }
}
}  // namespace

extern "C" {

static int render_impl(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes,
                       const uint8_t channels[4], const float clear_color[4], const forma_rect_t* crop_or_null, int cache_id,
                       forma_timings_t* timings, bool defer_dst) {
    if (!ctx) return FORMA_E_ARG;
    int rc = check_paint_args(ctx, dst, width, height, stride_bytes, channels, clear_color);
    if (rc) return rc;
    if (cache_id >= 32) return fail(ctx, FORMA_E_ARG, "cache_id out of range");   // SmallBitSet u32, small_bit_set.rs:17-57
    if (ctx->multi) return multi_render(ctx, dst, width, height, stride_bytes, channels, clear_color, crop_or_null, cache_id, timings);
    HIPCHECK(hipSetDevice(ctx->device));
    PaintArgs a{width, height, channels, clear_color, crop_or_null, cache_id};
    // Several frames in flight: a device-resident frame without a cache is ENQUEUED on the next slot and this call returns;
    // it is verified (and, if a prediction failed, re-run) when the slot is needed again or when any call needs its result.
    // Frames that write caller memory, use a buffer-layer cache (frame k + 1 reads what frame k left in it) or ask for
    // timings keep the synchronous contract of the reference: `dst` is fully written when the call returns — unless the
    // caller asked for the deferred form (forma_hip_render_enqueue): then the image also travels while later frames run.
    if (ctx->slots.size() > 1 && (!dst || defer_dst) && cache_id < 0 && !timings) {
        forma_hip_ctx* sl = ctx->slots[ctx->next_slot++ % ctx->slots.size()];
        if ((rc = settle_slot(sl))) { if (sl != ctx) memcpy(ctx->err, sl->err, sizeof ctx->err); return rc; }
        ctx->last = sl;
        clear_stage_flags(sl);
        if (width != sl->pred_w || height != sl->pred_h) { sl->pred_counts_valid = false; sl->pred_w = width; sl->pred_h = height; }
        if (sl->pred_valid && sl->pred_counts_valid && !sl->no_async) {
            forma_hip_ctx::Deferred& d = sl->def;
            d.width = width; d.height = height; memcpy(d.channels, channels, 4); memcpy(d.clear, clear_color, 16);
            d.has_crop = crop_or_null != nullptr; if (crop_or_null) d.crop = *crop_or_null;
            d.dst = dst; d.stride = stride_bytes;
            PaintArgs as{width, height, d.channels, d.clear, d.has_crop ? &d.crop : nullptr, -1};
            sl->frame_has_dst = dst != nullptr;                   // (sort_workgroups: such a frame's digit passes keep the whole chip)
            rc = enqueue_async_frame(sl, as, false, &d.bN, &d.bJ);
            sl->frame_has_dst = false;
            if (rc == FORMA_OK) sl->pending = true;
            if (rc == FORMA_OK && dst) {
                // the image leaves speculatively, in one piece behind the frame's kernels (verified at settle time; a void frame
                // is run again into `dst`): it crosses PCIe while the NEXT frames are rasterized, sorted and painted
                rc = copy_image_out(sl, dst, stride_bytes, false, as);
                sl->image_sent = rc == FORMA_OK;
            }
        } else {
            rc = render_sync(sl, a, dst, stride_bytes, false, nullptr);
        }
        if (rc && sl != ctx) memcpy(ctx->err, sl->err, sizeof ctx->err);
        return rc;
    }
    if ((rc = fd_drain(ctx))) return rc;
    ctx->last = ctx;
    return render_on(ctx, dst, a, stride_bytes, timings);
}

int forma_hip_render(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes,
                     const uint8_t channels[4], const float clear_color[4], const forma_rect_t* crop_or_null, int cache_id,
                     forma_timings_t* timings) {
    return render_impl(ctx, dst, width, height, stride_bytes, channels, clear_color, crop_or_null, cache_id, timings, false);
}

int forma_hip_render_enqueue(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes,
                             const uint8_t channels[4], const float clear_color[4], const forma_rect_t* crop_or_null) {
    return render_impl(ctx, dst, width, height, stride_bytes, channels, clear_color, crop_or_null, -1, nullptr, true);
}

// Caller buffers the renderer writes often (a window's frame buffer): page-locked once, the device-to-host copies into them are
// truly asynchronous and run at the link's rate; memory HIP does not know is pinned and unpinned by the runtime on every copy.
int forma_hip_register_buffer(forma_hip_ctx* ctx, void* ptr, size_t bytes) {
    if (!ctx || !ptr || !bytes) return FORMA_E_ARG;
    forma_hip_ctx* k = ctx->multi ? multi_first(ctx) : ctx;
    HIPCHECK(hipSetDevice(k->device));
    for (auto& r : k->registered) if (r.first == ptr) return fail(ctx, FORMA_E_STATE, "buffer already registered");
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) return fail(ctx, FORMA_E_HIP, "hipHostRegister", e);
    k->registered.emplace_back(ptr, bytes);
    return FORMA_OK;
}
int forma_hip_unregister_buffer(forma_hip_ctx* ctx, void* ptr) {
    if (!ctx || !ptr) return FORMA_E_ARG;
    { const int rc = forma_hip_sync(ctx); if (rc) return rc; }           // (nothing may still be on its way into it)
    forma_hip_ctx* k = ctx->multi ? multi_first(ctx) : ctx;
    for (size_t i = 0; i < k->registered.size(); i++)
        if (k->registered[i].first == ptr) {
            (void)hipHostUnregister(ptr);
            k->registered.erase(k->registered.begin() + (long)i);
            return FORMA_OK;
        }
    return fail(ctx, FORMA_E_ARG, "buffer was not registered");
}

int forma_hip_set_frames_in_flight(forma_hip_ctx* ctx, int n) {
    if (!ctx) return FORMA_E_ARG;
    if (n < 1 || n > FORMA_MAX_FRAMES_IN_FLIGHT) return fail(ctx, FORMA_E_ARG, "frames in flight: 1 .. 8");
    if (ctx->multi) return multi_set_frames_in_flight(ctx, n);
    int rc = fd_drain(ctx);
    if (rc) return rc;
    if (ctx->slots.empty()) ctx->slots.push_back(ctx);
    while ((int)ctx->slots.size() > n) { forma_hip_ctx* sl = ctx->slots.back(); ctx->slots.pop_back(); forma_hip_destroy(sl); }
    while ((int)ctx->slots.size() < n) {
        forma_hip_ctx* sl = nullptr;
        if ((rc = forma_hip_create(&sl, ctx->device))) return fail(ctx, rc, "frames in flight: cannot create a frame slot");
        sl->owner = ctx;
        sl->digit_bits = ctx->digit_bits; sl->no_async = ctx->no_async; sl->global_runsort = ctx->global_runsort;
        ctx->slots.push_back(sl);
    }
    if (ctx->slots.size() == 1) ctx->slots.clear();
    ctx->next_slot = 0; ctx->last = ctx;
    HIPCHECK(hipSetDevice(ctx->device));
    share_scene(ctx);
    return FORMA_OK;
}

int forma_hip_multi_layout(forma_hip_ctx* ctx, int layout) {
    if (!ctx) return FORMA_E_ARG;
    if (!ctx->multi) return fail(ctx, FORMA_E_STATE, "not a multi-device context");
    return multi_set_layout(ctx, layout);
}

int forma_hip_sync(forma_hip_ctx* ctx) {
    if (!ctx) return FORMA_E_ARG;
    if (ctx->multi) return multi_sync(ctx);
    return fd_drain(ctx);
}

int forma_hip_context_info(forma_hip_ctx* ctx, forma_context_info_t* out) {
    if (!ctx || !out) return FORMA_E_ARG;
    memset(out, 0, sizeof *out);
    if (ctx->multi) { multi_info(ctx, out); return FORMA_OK; }
    out->n_devices = 1; out->devices[0] = ctx->device; out->transport = FORMA_TRANSPORT_NONE;
    out->frames_in_flight = ctx->slots.empty() ? 1u : (uint32_t)ctx->slots.size();
    return FORMA_OK;
}

int forma_hip_kernel_times(forma_hip_ctx* ctx, forma_kernel_time_t* out, size_t capacity, size_t* out_n) {
    if (!ctx || !out_n || (capacity && !out)) return FORMA_E_ARG;
    if (ctx->multi) return fail(ctx, FORMA_E_STATE, "kernel times are kept per device: ask a single-device context");
    const forma_hip_ctx* c = last_slot(ctx);
    *out_n = (size_t)c->kt_n_done + (size_t)c->kt.dropped;   // (> KernelTimer::CAP: that many launches of the frame ran untimed)
    for (int i = 0; i < c->kt_n_done && (size_t)i < capacity; i++) {       // (dropped launches have no entry: only the count says so)
        forma_kernel_time_t& o = out[i];
        memset(&o, 0, sizeof o);
        const char* nm = c->kt.name[i];                   // "#kern" of FORMA_LAUNCH: "k_name" or "(k_name<ARGS>)"
        while (*nm == '(' || *nm == ' ') nm++;
        size_t k = 0;
        while (nm[k] && nm[k] != '<' && nm[k] != ')' && k + 1 < sizeof o.name) { o.name[k] = nm[k]; k++; }
        o.start_us = c->kt_start_us[i]; o.us = c->kt_dur_us[i]; o.stage = (uint32_t)c->kt.stage[i];
    }
    return FORMA_OK;
}

// The digit plan of a frame's segment sort, as the library would make it (host logic only: no device is touched).
int forma_hip_sort_plan(uint64_t live_key_bits, int layer_sorted, int digit_bits, const uint32_t* field_range, forma_sort_plan_t* out) {
    if (!out || (digit_bits != 0 && digit_bits != 4 && digit_bits != 8 && digit_bits != 9)) return FORMA_E_ARG;
    KeyRange R{0u, 0u, 0u, 0u, false};
    if (field_range) R = KeyRange{field_range[0], field_range[1], field_range[2], field_range[3], true};
    bool biased = false;
    const SortPlan P = make_segment_sort_plan(live_key_bits & 0xFFFFFFFFFFFull, layer_sorted != 0, digit_bits, field_range ? &R : nullptr, &biased);
    memset(out, 0, sizeof *out);
    out->n_passes = (uint32_t)P.n_passes; out->biased = biased ? 1u : 0u;
    for (int p = 0; p < P.n_passes && p < FORMA_SORT_MAX_PASSES; p++) {
        out->shift[p] = (uint32_t)P.shift[p]; out->mask[p] = P.mask[p]; out->bias[p] = P.bias[p];
    }
    return FORMA_OK;
}

// Per-frame device memory is grown to the largest frame seen and kept (a steady-state renderer never allocates).  trim gives
// it back: everything a frame writes before it reads — streams, records, tables, the scratch image — of the context and of
// its frame slots.  The scene (geometry, styles, images) and the buffer-layer caches (state across frames) stay.  The next
// frame allocates what it needs and runs synchronously, like the first frame of a geometry.
int forma_hip_trim(forma_hip_ctx* ctx) {
    if (!ctx) return FORMA_E_ARG;
    if (ctx->multi) return multi_trim(ctx);
    { const int rc = fd_drain(ctx); if (rc) return rc; }
    std::vector<forma_hip_ctx*> all{ctx};
    for (forma_hip_ctx* sl : ctx->slots) if (sl != ctx) all.push_back(sl);
    for (forma_hip_ctx* c : all) {
        HIPCHECK(hipSetDevice(c->device));
        HIPCHECK(hipStreamSynchronize(c->stream));
        DevBuf* frame[] = {&c->l_order, &c->l_x0, &c->l_y0, &c->l_dx, &c->l_dy, &c->l_a, &c->l_b, &c->l_c, &c->l_d, &c->l_len,
                           &c->scan_tmp, &c->cl_idx, &c->cl_start, &c->block_first, &c->prep_scratch, &c->seg_u, &c->seg_a, &c->seg_b,
                           &c->sort_counters, &c->records, &c->rk_u, &c->rk_a, &c->rk_b, &c->blk_edge, &c->runs_scratch, &c->row_tab,
                           &c->span_key, &c->span_cov, &c->image, &c->xscratch, &c->ras_masks, &c->xmask,
                           &c->huge_offs, &c->huge_key, &c->huge_tmp, &c->huge_flag, &c->grp_tab, &c->grp_list, &c->run_lt, &c->rec_sp, &c->run_lt_sp, &c->row_sp, &c->pack_list, &c->pack_pix,
                           &c->order_buf};
        c->order_cur = -1; c->order_pending = -1; c->order_cnt_dev = nullptr; c->order_keep_dev = nullptr;
        size_t freed = 0;
        for (DevBuf* b : frame) { if (!b->borrowed) freed += b->cap; b->release(); }
        if (ctx->dbg.trim_debug) {
            size_t kept = 0;
            DevBuf* rest[] = {&c->x, &c->y, &c->line_slot, &c->geoms, &c->style_off, &c->style_words, &c->layer_sf, &c->layer_col, &c->unchanged,
                              &c->images, &c->texels, &c->info, &c->info_init, &c->cache_written, &c->xsend, &c->xrecv};
            for (DevBuf* b : rest) if (!b->borrowed) kept += b->cap;
            for (auto& tcache : c->caches) kept += tcache.tiles.cap + tcache.image.cap;
            fprintf(stderr, "[forma_hip_trim] context %p: released %zu bytes, keeps %zu\n", (void*)c, freed, kept);
        }
        if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_cap = 0; }   // (pinned: a whole 4K image after a cache frame)
        c->sorted = nullptr; c->n_seg = 0; c->have_unsorted = false;
        c->cur_image = nullptr; c->img_w = 0; c->img_h = 0;
        c->pending_masks = PendingMasks{nullptr, 0u};
        c->last_written = 0;
        clear_stage_flags(c);
    }
    invalidate_counts(ctx);
    ctx->pred_counts_valid = false; ctx->xpred_valid = false;
    return FORMA_OK;
}

int forma_hip_cache_clear(forma_hip_ctx* ctx, int cache_id) {
    if (!ctx || cache_id < 0 || cache_id >= 32) return fail(ctx, FORMA_E_ARG, "cache_id out of range");
    if (ctx->multi) return multi_cache_clear(ctx, cache_id);
    { const int rc = fd_drain(ctx); if (rc) return rc; }
    HIPCHECK(hipSetDevice(ctx->device));
    forma_hip_ctx::TileCache& c = ctx->caches[cache_id];               // BufferLayerCache::clear, buffer/mod.rs:189-196
    c.has_clear = false;
    if (c.tiles.p) HIPCHECK(hipMemsetAsync(c.tiles.p, 0, c.tiles.cap, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return FORMA_OK;
}

// ---- inspection ------------------------------------------------------------------------------------------
int forma_hip_read_segments(forma_hip_ctx* ctx, int which, uint64_t* out, size_t capacity, size_t* out_n) {
    if (!ctx || !out_n) return FORMA_E_ARG;
    if (ctx->multi) return multi_read_segments(ctx, which, out, capacity, out_n);
    { const int rc = fd_drain(ctx); if (rc) return rc; }
    forma_hip_ctx* const owner = ctx;
    ctx = last_slot(ctx);                                 // the slot that rendered the most recent frame
    if (ctx != owner) ctx->err[0] = 0;
    struct CopyErr { forma_hip_ctx* o; forma_hip_ctx* s; ~CopyErr() { if (o != s && s->err[0]) memcpy(o->err, s->err, sizeof o->err); } } copy_err{owner, ctx};
    return fd_read_stream(ctx, which, out, capacity, out_n);
}

int forma_hip_read_image(forma_hip_ctx* ctx, uint8_t* dst, size_t stride_bytes) {
    if (!ctx || !dst) return FORMA_E_ARG;
    if (ctx->multi) return multi_read_image(ctx, dst, stride_bytes);
    { const int rc = fd_drain(ctx); if (rc) return rc; }
    forma_hip_ctx* const owner = ctx;
    ctx = last_slot(ctx);
    if (ctx != owner) ctx->err[0] = 0;
    struct CopyErr { forma_hip_ctx* o; forma_hip_ctx* s; ~CopyErr() { if (o != s && s->err[0]) memcpy(o->err, s->err, sizeof o->err); } } copy_err{owner, ctx};
    if (!ctx->img_w) return fail(ctx, FORMA_E_STATE, "no image on the device");
    if ((size_t)ctx->img_w * 4 > stride_bytes) return fail(ctx, FORMA_E_ARG, "width exceeds width stride");
    HIPCHECK(hipSetDevice(ctx->device));
    HIPCHECK(hipMemcpy2D(dst, stride_bytes, ctx->cur_image, (size_t)ctx->img_w * 4, (size_t)ctx->img_w * 4, ctx->img_h, hipMemcpyDeviceToHost));
    return FORMA_OK;
}

int forma_hip_tiles_written(forma_hip_ctx* ctx, uint8_t* flags, size_t n_tiles) {
    if (!ctx || !flags) return FORMA_E_ARG;
    if (ctx->multi) return multi_tiles_written(ctx, flags, n_tiles);
    { const int rc = fd_drain(ctx); if (rc) return rc; }
    forma_hip_ctx* const owner = ctx;
    ctx = last_slot(ctx);
    if (ctx != owner) ctx->err[0] = 0;
    struct CopyErr { forma_hip_ctx* o; forma_hip_ctx* s; ~CopyErr() { if (o != s && s->err[0]) memcpy(o->err, s->err, sizeof o->err); } } copy_err{owner, ctx};
    return fd_tiles_written(ctx, flags, n_tiles);
}

}  // extern "C"

// the stream / the written-tile flags of exactly this context's last frame (no slot resolution: multi.cpp names the slot)
int fd_read_stream(forma_hip_ctx* ctx, int which, uint64_t* out, size_t capacity, size_t* out_n) {
    *out_n = ctx->n_seg;
    if (which == 0 && !ctx->have_unsorted) return fd_fail(ctx, FORMA_E_STATE, "no unsorted stream on the device");
    if (ctx->n_seg > capacity) return fd_fail(ctx, FORMA_E_CAPACITY, "segment capacity too small");
    if (ctx->n_seg == 0) return FORMA_OK;
    if (!out) return FORMA_E_ARG;
    HIPCHECK(hipSetDevice(ctx->device));
    const void* src = which == 0 ? ctx->seg_u.p : (const void*)ctx->sorted;
    if (!src) return fd_fail(ctx, FORMA_E_STATE, "no segments on the device");
    HIPCHECK(hipMemcpy(out, src, ctx->n_seg * 8, hipMemcpyDeviceToHost));
    return FORMA_OK;
}
int fd_read_sorted(forma_hip_ctx* ctx, uint64_t* out, size_t capacity, size_t* out_n) { return fd_read_stream(ctx, 1, out, capacity, out_n); }

int fd_tiles_written(forma_hip_ctx* ctx, uint8_t* flags, size_t n_tiles) {
    if (!ctx->lw_valid) return fd_fail(ctx, FORMA_E_STATE, "no frame rendered yet");
    const size_t T = (size_t)ctx->lw_tiles_w * ctx->lw_tiles_h;
    if (n_tiles < T) return fd_fail(ctx, FORMA_E_CAPACITY, "tile flag capacity too small");
    memset(flags, 0, n_tiles);
    if (ctx->lw_cache && !ctx->lw_flags_on_host) {        // device-resident frame (dst == NULL): fetch the flags now
        HIPCHECK(hipSetDevice(ctx->device));
        if (ctx->h_written_cap < T) {
            if (ctx->h_written) (void)hipHostFree(ctx->h_written);
            ctx->h_written = nullptr; ctx->h_written_cap = 0;
            HIPCHECK(hipHostMalloc((void**)&ctx->h_written, T, hipHostMallocDefault));
            ctx->h_written_cap = T;
        }
        HIPCHECK(hipMemcpyAsync(ctx->h_written, ctx->cache_written.p, T, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        ctx->lw_flags_on_host = true;
    }
    for (uint32_t ty = ctx->lw_ty0; ty < ctx->lw_ty1; ty++)
        for (uint32_t tx = ctx->lw_tx0; tx < ctx->lw_tx1; tx++) {
            const size_t t = (size_t)ty * ctx->lw_tiles_w + tx;
            flags[t] = ctx->lw_cache ? (ctx->h_written[t] ? 1 : 0) : 1;
        }
    return FORMA_OK;
}

extern "C" {

// ---- multi-GPU -----------------------------------------------------------------------------------------------
int forma_hip_set_band(forma_hip_ctx* ctx, uint32_t row0, uint32_t row1) {
    ENTER_SINGLE(ctx);
    if (row1 != 0 && row0 >= row1) return fail(ctx, FORMA_E_ARG, "empty band");
    ctx->band_row0 = row1 ? row0 : 0; ctx->band_row1 = row1;
    invalidate_counts(ctx);
    share_scene(ctx);
    return FORMA_OK;
}

int forma_hip_segments_device(forma_hip_ctx* ctx, int which, uint64_t** dev_ptr, size_t* n) {
    if (!ctx || !dev_ptr || !n) return FORMA_E_ARG;
    ENTER_SINGLE(ctx);
    *dev_ptr = which == 0 ? ctx->seg_u.as<uint64_t>() : ctx->sorted;
    *n = ctx->n_seg;
    return FORMA_OK;
}

int forma_hip_rasterize_frame(forma_hip_ctx* ctx, uint32_t width, uint32_t height, forma_timings_t* timings) {
    ENTER_SINGLE(ctx);
    int rc = check_canvas(ctx, width, height);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(ctx->device));
    clear_stage_flags(ctx);
    if ((rc = poison_frame_buffers(ctx))) return rc;
    if ((rc = run_rasterize_frame(ctx, width, height, timings != nullptr))) return rc;
    ctx->n_passes = 0; ctx->last_runs = 0; ctx->last_entries = 0;
    return finish_frame(ctx, timings);
}

int forma_hip_reserve_segments(forma_hip_ctx* ctx, size_t n, uint64_t** dev_ptr) {
    if (!ctx || !dev_ptr) return FORMA_E_ARG;
    ENTER_SINGLE(ctx);
    HIPCHECK(hipSetDevice(ctx->device));
    // growing seg_u would drop the rasterized stream the caller may still be sending: grow seg_b (unused until the sort)
    HIPCHECK(ctx->seg_b.ensure((std::max<size_t>(n, 1) + SEG_PAD) * 8));
    *dev_ptr = ctx->seg_b.as<uint64_t>();
    return FORMA_OK;
}

int forma_hip_sort_paint_frame(forma_hip_ctx* ctx, size_t n, uint8_t* dst, uint32_t width, uint32_t height,
                               size_t stride_bytes, const uint8_t channels[4], const float clear_color[4],
                               const forma_rect_t* crop_or_null, forma_timings_t* timings) {
    ENTER_SINGLE(ctx);
    int rc = check_paint_args(ctx, dst, width, height, stride_bytes, channels, clear_color);
    if (rc) return rc;
    if (n * 8 > ctx->seg_b.cap) return fail(ctx, FORMA_E_STATE, "reserve_segments first");
    HIPCHECK(hipSetDevice(ctx->device));
    const bool timing = timings != nullptr;
    clear_stage_flags(ctx);
    ctx->pz = forma_hip_ctx::PreZero();
    if ((rc = reset_info(ctx))) return rc;
    // the received stream lives in seg_b: move it to seg_u (the sort's read-only input) so a/b can ping-pong
    HIPCHECK(ctx->seg_u.ensure((std::max<size_t>(n, 1) + SEG_PAD) * 8));
    if (n) HIPCHECK(hipMemcpyAsync(ctx->seg_u.p, ctx->seg_b.p, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->n_seg = n; ctx->have_unsorted = true;
    ctx->live44 = 0xFFFFFFFFFFFull;                   // no varying-bit mask for a received stream: sort every digit
    ctx->layer_sorted = false; ctx->speculated = false; // slices that are layer-sorted need not concatenate to a layer-sorted stream
    if ((rc = run_sort(ctx, ctx->seg_u.as<uint64_t>(), DevCount{nullptr, (uint32_t)n}, timing))) return rc;
    PaintArgs a{width, height, channels, clear_color, crop_or_null};
    if ((rc = run_paint(ctx, DevCount{nullptr, (uint32_t)n}, a, timing))) return rc;
    if ((rc = read_info(ctx)) || (rc = finish_paint(ctx))) return rc;
    if ((rc = copy_image_out(ctx, dst, stride_bytes, timing, a))) return rc;
    if (dst) HIPCHECK(hipStreamSynchronize(ctx->stream));
    return finish_frame(ctx, timings, true);
}


// ---- multi-GPU, exchange layout: every rank rasterizes 1/G of the LINES, pixel segments travel to the rank
//      that owns their tile row (ONE all-to-all of u64 payloads, driven by the host language on this context's stream),
//      the owner sorts and paints its band ---------------------------------------------------------------------------------
int forma_hip_stream(forma_hip_ctx* ctx, void** stream) {
    if (!ctx || !stream) return FORMA_E_ARG;
    if (ctx->multi) return fail(ctx, FORMA_E_STATE, "a multi-device context has one stream per device");
    *stream = (void*)ctx->stream;
    return FORMA_OK;
}

int forma_hip_exchange_plan(forma_hip_ctx* ctx, const uint32_t* row_edges, uint32_t n_ranks, uint32_t pair_capacity) {
    if (!ctx || !row_edges) return FORMA_E_ARG;
    ENTER_SINGLE(ctx);
    if (n_ranks < 1 || n_ranks > FORMA_MAX_RANKS) return fail(ctx, FORMA_E_ARG, "1 .. 8 ranks");
    // (a band may be empty — a canvas with fewer tile rows than ranks: its owner receives nothing and paints nothing)
    for (uint32_t g = 0; g < n_ranks; g++) if (row_edges[g] > row_edges[g + 1]) return fail(ctx, FORMA_E_ARG, "tile-row band edges must not decrease");
    if (row_edges[0] >= row_edges[n_ranks]) return fail(ctx, FORMA_E_ARG, "no tile rows");
    if (pair_capacity == 0 || (uint64_t)pair_capacity * n_ranks >= (1ull << 30)) return fail(ctx, FORMA_E_ARG, "pair capacity out of range");
    HIPCHECK(hipSetDevice(ctx->device));
    ctx->xbands.n = n_ranks;
    for (uint32_t g = 0; g <= n_ranks; g++) ctx->xbands.edge[g] = row_edges[g];
    ctx->xcap = pair_capacity;
    const size_t words = (size_t)n_ranks * ((size_t)pair_capacity + 1);    // a bucket = pair_capacity data words + its header
    HIPCHECK(ctx->xsend.ensure((words + SEG_PAD) * 8));
    HIPCHECK(ctx->xrecv.ensure((words + SEG_PAD) * 8));
    HIPCHECK(hipMemsetAsync(ctx->xsend.p, 0, (words + SEG_PAD) * 8, ctx->stream));      // (headers included: empty buckets until the first frame)
    HIPCHECK(hipMemsetAsync(ctx->xrecv.p, 0, (words + SEG_PAD) * 8, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    ctx->xplanned = true;
    ctx->pred_valid = false; ctx->pred_counts_valid = false; ctx->xpred_valid = false;
    return FORMA_OK;
}

int forma_hip_exchange_buffers(forma_hip_ctx* ctx, uint64_t** send, uint64_t** recv, size_t* words_per_pair) {
    if (!ctx || !send || !recv || !words_per_pair) return FORMA_E_ARG;
    ENTER_SINGLE(ctx);
    if (!ctx->xplanned) return fail(ctx, FORMA_E_STATE, "forma_hip_exchange_plan first");
    *send = ctx->xsend.as<uint64_t>(); *recv = ctx->xrecv.as<uint64_t>();
    *words_per_pair = (size_t)ctx->xcap + 1;
    return FORMA_OK;
}

int forma_hip_rasterize_bucket_frame(forma_hip_ctx* ctx, uint32_t width, uint32_t height, forma_timings_t* timings) {
    ENTER_SINGLE(ctx);
    if (!ctx->xplanned) return fail(ctx, FORMA_E_STATE, "forma_hip_exchange_plan first");
    int rc = check_canvas(ctx, width, height);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(ctx->device));
    const bool timing = timings != nullptr;
    clear_stage_flags(ctx);
    // read-back-free when the previous frame's local segment count is known (bound with slack), else synchronous.  A frame
    // whose true count exceeds the bound is flagged by k_owner_scan to every receiver (FORMA_E_CAPACITY -> the host re-plans).
    // The true count of every bucket frame is copied to a pinned word on the stream; the owner's half of the frame ends in a
    // stream synchronisation, so by the next bucket frame it has landed and refreshes the prediction (an animation drifts).
    if (width != ctx->xpred_w || height != ctx->xpred_h) { ctx->xpred_valid = false; ctx->xpred_w = width; ctx->xpred_h = height; }
    if (ctx->xpred_valid && ctx->h_xlocal[1]) {
        HIPCHECK(hipStreamSynchronize(ctx->stream));      // (a no-op after forma_hip_gather_sort_paint_frame)
        ctx->xpred_N = ctx->h_xlocal[0];
    }
    ctx->h_xlocal[1] = 0;
    const uint32_t bN = (ctx->xpred_valid && !ctx->no_async) ? ctx->xpred_N + ctx->xpred_N / 16 + 4096 : 0;
    if ((rc = poison_frame_buffers(ctx))) return rc;
    // read-back-free on both halves: this call's first kernel also clears what the owner's half of the frame
    // (forma_hip_gather_sort_paint_frame: sort scratch, tile tables, run chain for a band of n_ranks x capacity segments) expects
    ZeroJobs Z; forma_hip_ctx::PreZero cleared;
    const bool ahead = bN && ctx->pred_valid && ctx->pred_counts_valid && !ctx->no_async;
    if (ahead && (rc = plan_zero_jobs(ctx, width, height, ctx->xbands.n * ctx->xcap, ctx->xbands.n * ctx->xcap, &Z, &cleared))) return rc;
    if ((rc = run_rasterize_frame(ctx, width, height, timing, false, bN, ahead ? &Z : nullptr, ahead ? &cleared : nullptr))) return rc;
    FrameInfo* dinfo = ctx->info.as<FrameInfo>();
    DevCount nc = bN ? DevCount{&dinfo->n_segments, bN} : DevCount{nullptr, (uint32_t)ctx->n_seg};
    if (!bN) { ctx->xpred_N = (uint32_t)ctx->n_seg; ctx->xpred_valid = true; }
    HIPCHECK(ctx->seg_u.ensure(((size_t)std::max<uint32_t>(nc.bound, 1) + SEG_PAD) * 8));
    HIPCHECK(ctx->xscratch.ensure(owner_scratch_words(std::max<size_t>(nc.bound, 1)) * 4));
    stage_begin(ctx, ST_XCHG, timing);
    launch_owner_bucket(ctx->stream, ctx->seg_u.as<uint64_t>(), nc, ctx->xbands, ctx->xcap, ctx->xscratch.as<uint32_t>(),
                        ctx->xsend.as<uint64_t>(), dinfo);
    stage_end(ctx, ST_XCHG, timing);
    HIPCHECK(hipGetLastError());
    if (timing) {                                            // (the only host wait of this call, and only when timings are asked for)
        if (bN) { HIPCHECK(hipMemcpyAsync(&ctx->h_xlocal[0], &dinfo->n_segments, 4, hipMemcpyDeviceToHost, ctx->stream)); ctx->h_xlocal[1] = 1; }
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        ctx->n_passes = 0; ctx->last_runs = 0;
        HIPCHECK(hipMemcpy(ctx->h_info, ctx->info.p, sizeof(FrameInfo), hipMemcpyDeviceToHost));
        return finish_frame(ctx, timings, true);
    }
    if (bN) {
        // the true local count goes to a pinned word and FrameInfo returns to its pristine state (the owner's half starts
        // from one): one tiny kernel instead of a copy here and a reset there
        if ((rc = frame_tail(ctx, false, &ctx->h_xlocal[0]))) return rc;
        ctx->h_xlocal[1] = 1;
    }
    return FORMA_OK;
}

int forma_hip_gather_sort_paint_frame(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes,
                                      const uint8_t channels[4], const float clear_color[4], const forma_rect_t* crop_or_null,
                                      forma_timings_t* timings) {
    ENTER_SINGLE(ctx);
    return fd_gather_sort_paint(ctx, dst, width, height, stride_bytes, channels, clear_color, crop_or_null, -1, timings);
}

}  // extern "C"

// The owner's half of an exchange frame (forma_hip_gather_sort_paint_frame; a device of a multi-device context), in three
// pieces so that a multi-device context can keep frames in flight: ENQUEUE (read-back-free: everything goes onto the stream,
// nothing waits), COMPLETE (wait, verify the predictions, copy out) and the SYNCHRONOUS form (first frame of a plan, or a
// prediction failed: N, the key masks and J are read back).
namespace {

struct GspArgs {
    uint8_t* dst; uint32_t width, height; size_t stride_bytes; const uint8_t* channels; const float* clear; const forma_rect_t* crop;
    int cache_id; forma_timings_t* timings;
};

int gsp_overflow(forma_hip_ctx* ctx) {
    ctx->xoverflowed = true;                               // (state, not text: multi_render re-plans on it)
    return fail(ctx, FORMA_E_CAPACITY, "exchange: a bucket exceeds the pair capacity (re-plan)");
}

// *enqueued = false: the context has no predictions yet (or read-back-free frames are off) — nothing was enqueued
int gsp_enqueue(forma_hip_ctx* ctx, const GspArgs& g, bool* enqueued, uint32_t* bJ_out) {
    *enqueued = false;
    if (!(ctx->pred_valid && ctx->pred_counts_valid && !ctx->no_async)) return FORMA_OK;
    const bool timing = g.timings != nullptr;
    const uint32_t G = ctx->xbands.n, bound = G * ctx->xcap;
    const bool self = G == 1 && !ctx->xuse_recv;
    const uint64_t* recv = self ? ctx->xsend.as<uint64_t>() : ctx->xrecv.as<uint64_t>();
    FrameInfo* dinfo = ctx->info.as<FrameInfo>();
    PaintArgs a{g.width, g.height, g.channels, g.clear, g.crop, g.cache_id};
    int rc;
    const uint32_t bJ = ctx->pred_J + ctx->pred_J / 16 + 4096;
    *bJ_out = bJ;
    ctx->live44 = ctx->pred_live44; ctx->layer_sorted = ctx->pred_layer_sorted; ctx->speculated = true;
    uint64_t live = ctx->live44;
    if (ctx->layer_sorted) live &= ~0x1FFFFFull;
    // The received buckets are sorted where they lie: the histogram kernel and the first digit pass read the rank-major
    // concatenation through a logical -> physical index map, so nothing is gathered (k_gather_chunks: one more read and
    // write of the whole band).  Needs at least one digit pass; FORMA_HIP_DEBUG=xgather keeps the gather (A/B, tests).
    if (!ctx->xgather_always && live != 0 && bound > 1) {
        if ((rc = reset_info(ctx))) return rc;
        ctx->have_unsorted = false; ctx->n_lines = 0;
        const ChunkedSrc C{recv, G, ctx->xcap, ctx->xmask.as<uint32_t>()};
        if ((rc = run_sort(ctx, recv, DevCount{&dinfo->n_segments, bound}, timing, 0, &C))) return rc;
        ctx->pending_masks = PendingMasks{ctx->xmask.as<uint32_t>(), sort_hist_blocks(bound)};
    } else {
        if ((rc = reset_info(ctx))) return rc;
        stage_begin(ctx, ST_XCHG, timing);
        launch_gather_chunks(ctx->stream, recv, G, ctx->xcap, ctx->seg_u.as<uint64_t>(), dinfo, ctx->xmask.as<uint32_t>(), /*reduce_now=*/false);
        ctx->pending_masks = PendingMasks{ctx->xmask.as<uint32_t>(), (uint32_t)(gather_mask_words(G, ctx->xcap) / 8)};
        stage_end(ctx, ST_XCHG, timing);
        ctx->have_unsorted = true; ctx->n_lines = 0;
        if ((rc = run_sort(ctx, ctx->seg_u.as<uint64_t>(), DevCount{&dinfo->n_segments, bound}, timing))) return rc;
    }
    if ((rc = run_paint(ctx, DevCount{&dinfo->n_segments, bound}, a, timing, bJ))) return rc;
    if ((rc = frame_tail(ctx, true, nullptr))) return rc;
    *enqueued = true;
    return FORMA_OK;
}

// FORMA_RETRY: a prediction failed, nothing of the frame may be used (the caller runs gsp_sync)
int gsp_complete(forma_hip_ctx* ctx, const GspArgs& g, uint32_t bJ) {
    const bool timing = g.timings != nullptr;
    PaintArgs a{g.width, g.height, g.channels, g.clear, g.crop, g.cache_id};
    int rc;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->h_info->exchange_overflow) return gsp_overflow(ctx);
    const uint32_t N = ctx->h_info->n_segments, J = ctx->h_info->n_runs;
    ctx->n_seg = N; ctx->last_runs = J;
    if (ctx->h_info->plan_bad && ctx->small_tried) ctx->small_banned = true;
    if (ctx->h_info->plan_bad && ctx->covl_tried) ctx->covl_banned = true;
    if (ctx->h_info->plan_bad && ctx->plan_biased) ban_bias(ctx);
    if (!ctx->h_info->plan_bad && J <= bJ) {
        if (ctx->bias_banned) ctx->bias_banned--;
        ctx->pred_J = J; ctx->pred_max_row = ctx->h_info->max_row_runs;
        if ((rc = finish_paint(ctx))) return rc;
        if ((rc = copy_image_out(ctx, g.dst, g.stride_bytes, timing, a))) return rc;
        if (g.dst) HIPCHECK(hipStreamSynchronize(ctx->stream));
        rc = finish_frame(ctx, g.timings, true);
        frame_done(ctx, rc, a);
        return rc;
    }
    ctx->pred_counts_valid = false;
    clear_stage_flags(ctx);
    return FORMA_RETRY;
}

int gsp_sync(forma_hip_ctx* ctx, const GspArgs& g) {
    const bool timing = g.timings != nullptr;
    const uint32_t G = ctx->xbands.n;
    const bool self = G == 1 && !ctx->xuse_recv;
    const uint64_t* recv = self ? ctx->xsend.as<uint64_t>() : ctx->xrecv.as<uint64_t>();
    FrameInfo* dinfo = ctx->info.as<FrameInfo>();
    PaintArgs a{g.width, g.height, g.channels, g.clear, g.crop, g.cache_id};
    int rc;
    ctx->pz = forma_hip_ctx::PreZero();
    for (int attempt = 0; attempt < 2; attempt++) {                            // synchronous: N, key masks and J are read back
        if ((rc = reset_info(ctx))) return rc;
        stage_begin(ctx, ST_XCHG, timing);
        launch_gather_chunks(ctx->stream, recv, G, ctx->xcap, ctx->seg_u.as<uint64_t>(), dinfo, ctx->xmask.as<uint32_t>(), /*reduce_now=*/true);
        ctx->pending_masks = PendingMasks{nullptr, 0u};
        stage_end(ctx, ST_XCHG, timing);
        ctx->have_unsorted = true; ctx->n_lines = 0;
        if ((rc = read_info(ctx))) return rc;
        if (ctx->h_info->exchange_overflow) return gsp_overflow(ctx);
        ctx->n_seg = ctx->h_info->n_segments;
        const uint64_t k_or = (uint64_t)ctx->h_info->key_or | ((uint64_t)ctx->h_info->key_or_hi << 32);
        const uint64_t k_and = (uint64_t)ctx->h_info->key_and | ((uint64_t)ctx->h_info->key_and_hi << 32);
        ctx->live44 = ctx->n_seg ? ((k_or ^ k_and) & 0xFFFFFFFFFFFull) : 0;
        ctx->layer_sorted = ctx->h_info->layer_unsorted == 0; ctx->speculated = false;
        if ((rc = run_sort(ctx, ctx->seg_u.as<uint64_t>(), DevCount{nullptr, (uint32_t)ctx->n_seg}, timing))) return rc;
        rc = run_paint(ctx, DevCount{nullptr, (uint32_t)ctx->n_seg}, a, timing);
        if (rc == FORMA_RETRY) { clear_stage_flags(ctx); continue; }
        if (rc) return rc;
        if ((rc = read_info(ctx)) || (rc = finish_paint(ctx))) return rc;
        if ((rc = copy_image_out(ctx, g.dst, g.stride_bytes, timing, a))) return rc;
        if (g.dst) HIPCHECK(hipStreamSynchronize(ctx->stream));
        rc = finish_frame(ctx, g.timings, true);
        if (rc == FORMA_OK) { ctx->pred_J = ctx->last_runs; ctx->pred_counts_valid = true; }
        frame_done(ctx, rc, a);
        return rc;
    }
    return fail(ctx, FORMA_E_INTERNAL, "sort plan did not converge");
}

int gsp_prologue(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes, const uint8_t channels[4],
                 const float clear_color[4], int cache_id) {
    if (!ctx) return FORMA_E_ARG;
    if (!ctx->xplanned) return fail(ctx, FORMA_E_STATE, "forma_hip_exchange_plan first");
    int rc = check_paint_args(ctx, dst, width, height, stride_bytes, channels, clear_color);
    if (rc) return rc;
    if (cache_id >= 32) return fail(ctx, FORMA_E_ARG, "cache_id out of range");
    HIPCHECK(hipSetDevice(ctx->device));
    clear_stage_flags(ctx);
    ctx->xoverflowed = false;
    const uint32_t G = ctx->xbands.n, bound = G * ctx->xcap;
    HIPCHECK(ctx->seg_u.ensure(((size_t)bound + SEG_PAD) * 8));
    HIPCHECK(ctx->xmask.ensure(std::max<size_t>(gather_mask_words(G, ctx->xcap), (size_t)2048 * 8) * 4));
    return FORMA_OK;
}

}  // namespace

int fd_gather_sort_paint(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes,
                         const uint8_t channels[4], const float clear_color[4], const forma_rect_t* crop_or_null, int cache_id,
                         forma_timings_t* timings) {
    int rc = gsp_prologue(ctx, dst, width, height, stride_bytes, channels, clear_color, cache_id);
    if (rc) return rc;
    const GspArgs g{dst, width, height, stride_bytes, channels, clear_color, crop_or_null, cache_id, timings};
    bool enqueued = false; uint32_t bJ = 0;
    if ((rc = gsp_enqueue(ctx, g, &enqueued, &bJ))) return rc;
    if (enqueued) {
        rc = gsp_complete(ctx, g, bJ);
        if (rc != FORMA_RETRY) return rc;
    }
    return gsp_sync(ctx, g);
}

// A device-resident, cache-less frame of a multi-device context with frames in flight: the owner's half is ENQUEUED (or, when
// this slot has no predictions yet, run synchronously to its end) and the arguments are parked in the slot ...
int fd_gsp_defer(forma_hip_ctx* ctx, uint32_t width, uint32_t height, const uint8_t channels[4], const float clear_color[4],
                 const forma_rect_t* crop_or_null) {
    int rc = gsp_prologue(ctx, nullptr, width, height, 0, channels, clear_color, -1);
    if (rc) return rc;
    forma_hip_ctx::Deferred& d = ctx->def;
    d.width = width; d.height = height; memcpy(d.channels, channels, 4); memcpy(d.clear, clear_color, 16);
    d.has_crop = crop_or_null != nullptr; if (crop_or_null) d.crop = *crop_or_null;
    const GspArgs g{nullptr, width, height, 0, d.channels, d.clear, d.has_crop ? &d.crop : nullptr, -1, nullptr};
    bool enqueued = false;
    if ((rc = gsp_enqueue(ctx, g, &enqueued, &d.bJ))) return rc;
    if (enqueued) { ctx->xpending = true; return FORMA_OK; }
    return gsp_sync(ctx, g);
}
// ... and completed when the slot comes round again (or any call needs the result): FORMA_E_CAPACITY with ctx->xoverflowed when a
// bucket outgrew the plan (the caller re-plans and re-runs the frame), any other failed prediction is repaired here
int fd_gsp_settle(forma_hip_ctx* ctx) {
    if (!ctx->xpending) return FORMA_OK;
    ctx->xpending = false;
    HIPCHECK(hipSetDevice(ctx->device));
    const forma_hip_ctx::Deferred& d = ctx->def;
    const GspArgs g{nullptr, d.width, d.height, 0, d.channels, d.clear, d.has_crop ? &d.crop : nullptr, -1, nullptr};
    int rc = gsp_complete(ctx, g, d.bJ);
    if (rc == FORMA_RETRY) rc = gsp_sync(ctx, g);
    return rc;
}


// ---- helpers of the multi-device planner (multi.cpp) ---------------------------------------------------------------------
int fd_set_line_range(forma_hip_ctx* ctx, bool ranged, size_t lo, size_t hi) {
    if (!ctx) return FORMA_E_ARG;
    if (ranged && lo > hi) return fail(ctx, FORMA_E_ARG, "line range");
    int rc = fd_drain(ctx);
    if (rc) return rc;
    ctx->line_ranged = ranged; ctx->line_lo = ranged ? lo : 0; ctx->line_hi = ranged ? hi : 0;
    invalidate_counts(ctx);
    share_scene(ctx);
    return FORMA_OK;
}

int fd_line_sums(forma_hip_ctx* ctx, uint32_t width, uint32_t height, std::vector<uint32_t>& sums) {
    if (!ctx) return FORMA_E_ARG;
    int rc = fd_drain(ctx);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(ctx->device));
    const size_t n = ctx->n_points ? ctx->n_points - 1 : 0;
    sums.assign(n, 0u);
    if (n == 0) return FORMA_OK;
    // per-line lengths of ALL lines (the frame path's count kernel, segment.rs:298-383 restated in k_line_len), then the scan
    HIPCHECK(ctx->l_len.ensure(n * 4));
    HIPCHECK(ctx->prep_scratch.ensure(prepare_scratch_words(n) * 4));
    HIPCHECK(ctx->scan_tmp.ensure(scan_tmp_words(std::max<size_t>(n, 1 << 16)) * 4));
    const bool keep = ctx->line_ranged;
    const uint32_t keep_b0 = ctx->band_row0, keep_b1 = ctx->band_row1;     // (ALL lines, whole canvas: neither a line share nor a band)
    ctx->line_ranged = false; ctx->band_row0 = 0; ctx->band_row1 = 0;
    const LineSource S = geometry_source(ctx, width, height);
    ctx->line_ranged = keep; ctx->band_row0 = keep_b0; ctx->band_row1 = keep_b1;
    launch_line_lengths(ctx->stream, S, (uint32_t)n, ctx->l_len.as<uint32_t>(), ctx->prep_scratch.as<uint32_t>());
    launch_inclusive_scan_u32(ctx->stream, ctx->l_len.as<uint32_t>(), n, ctx->scan_tmp.as<uint32_t>(), nullptr);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(sums.data(), ctx->l_len.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return FORMA_OK;
}

int fd_row_histogram(forma_hip_ctx* ctx, uint32_t width, uint32_t height, uint32_t* hist, uint32_t* n_segments) {
    if (!ctx || !hist) return FORMA_E_ARG;
    int rc = fd_drain(ctx);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(ctx->device));
    clear_stage_flags(ctx);
    if ((rc = run_rasterize_frame(ctx, width, height, false))) return rc;          // synchronous: n_seg is known afterwards
    HIPCHECK(ctx->xscratch.ensure(std::max<size_t>(2048 * 4, ctx->xscratch.cap)));
    launch_row_histogram(ctx->stream, ctx->seg_u.as<uint64_t>(), DevCount{nullptr, (uint32_t)ctx->n_seg}, ctx->xscratch.as<uint32_t>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(hist, ctx->xscratch.p, 2048 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    if (n_segments) *n_segments = (uint32_t)ctx->n_seg;
    return FORMA_OK;
}

forma_hip_ctx* fd_last_slot(forma_hip_ctx* ctx) { return ctx->last ? ctx->last : ctx; }

int fd_copy_image_rows(forma_hip_ctx* ctx, uint8_t* dst, size_t stride_bytes, uint32_t y0, uint32_t y1) {
    if (!ctx || !dst) return FORMA_E_ARG;
    if (!ctx->img_w || !ctx->cur_image) return fail(ctx, FORMA_E_STATE, "no image on the device");
    y1 = std::min(y1, ctx->img_h);
    if (y0 >= y1) return FORMA_OK;
    HIPCHECK(hipSetDevice(ctx->device));
    const size_t pitch = (size_t)ctx->img_w * 4;
    HIPCHECK(hipMemcpy2D(dst + (size_t)y0 * stride_bytes, stride_bytes, ctx->cur_image + (size_t)y0 * pitch, pitch, pitch, y1 - y0,
                         hipMemcpyDeviceToHost));
    return FORMA_OK;
}
