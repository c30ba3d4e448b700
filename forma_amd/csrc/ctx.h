// ctx.h — the context behind the C ABI (include/forma_hip.h), shared by api.cpp (one device) and multi.cpp (several
// devices behind one context).  Private to libforma_hip.so.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "debug.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool borrowed = false;                  // a frame slot's view of its owner's scene buffer: never grown or freed here
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        if (borrowed) return hipErrorInvalidValue;
        size_t want = std::max(bytes, cap + cap / 2);
        want = std::max<size_t>(want, 256);
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) {
            cap = want;
            // FORMA_HIP_DEBUG=poison=<byte>: every fresh device allocation is filled with that byte — a kernel that reads what
            // nothing wrote this frame then misbehaves on every run instead of once per fresh box (tests/, tools/)
            static const int poison = forma_debug_parse().poison;
            if (poison >= 0) { e = hipMemset(p, poison, want); if (e == hipSuccess) e = hipDeviceSynchronize(); }
        }
        return e;
    }
    void release() { if (p && !borrowed) (void)hipFree(p); p = nullptr; cap = 0; borrowed = false; }
    void borrow(const DevBuf& o) { release(); p = o.p; cap = o.cap; borrowed = true; }
    template <class T> T* as() const { return (T*)p; }
};

constexpr uint32_t PAINT_WAVES_PER_CU = 24;     // wave slots of the general painter per CU (6 per SIMD): frames of at most n_cus x 24 painted
                                                // tiles — every tile gets a slot at once — are painted by strips (api.cpp paint_by_strips)
constexpr size_t SEG_PAD = 16;          // segment buffers are over-allocated: stream kernels read whole 64-byte lane pieces
enum { ST_PREPARE = 0, ST_RASTER, ST_SORT, ST_CARRY, ST_PAINT, ST_D2H, ST_XCHG, ST_COUNT };

struct forma_hip_ctx {
    int device = 0;
    uint32_t n_cus = 256;                   // hipDeviceProp_t::multiProcessorCount of `device` (MI355X: 256)
    hipStream_t stream = nullptr;
    char err[512] = {0};

    // scene
    DevBuf x, y, line_slot, geoms, style_off, style_words, unchanged, images, texels;
    DevBuf layer_sf, layer_col;             // per order: style summary for the carry pre-pass (set_styles)
    std::vector<uint32_t> h_layer_sf, h_layer_col;
    size_t n_points = 0, n_geoms = 0, n_orders = 0, n_words = 0, n_images = 0;
    uint32_t max_geom_order = 0;            // largest order any geom slot names (FORMA_NONE slots aside)
    uint32_t max_image_index = 0;           // largest image index a texture style names
    bool any_texture = false;
    bool scene_has_clips = false;
    DevBuf run_lt;                  // one word per run: layer16 | open | tile_x + 1 (RunStyle, common.h)
    DevBuf rec_sp, run_lt_sp, row_sp;       // launch_runs' BLOCKS numbering (BlkRuns, common.h): records and digests indexed like the segments (N entries)
    DevBuf grp_tab, grp_list;       // span group lists (SpanGroups, common.h): table per (row, slice, group) and the entry pool
    bool no_span_groups = false;    // FORMA_HIP_DEBUG=no_span_groups (A/B switch for tools/)
    bool force_span_groups = false; // FORMA_HIP_DEBUG=span_groups: on every frame and for every row, however few spans (tests)
    uint32_t pred_row_spans = 0;    // spans per painted tile row of the last verified frame: group lists pay above SPAN_GROUP_MIN_ROW
    uint32_t cur_rows_painted = 1;
    bool scene_simple = false;      // all layers solid + Over + unclipped: the painter's specialised kernel
    size_t costly_layers = 0;       // layers with a gradient / texture fill, a blend mode other than Over, or clipped (strip painters: api.cpp)
    bool have_unchanged = false;            // set_styles supplied per-order Layer::is_unchanged bytes
    // lines
    DevBuf l_order, l_x0, l_y0, l_dx, l_dy, l_a, l_b, l_c, l_d, l_len, scan_tmp;   // parity entry points only
    DevBuf cl_idx, cl_start, block_first, prep_scratch;                              // frame path: compacted line table
    bool pred_no_deep = false;              // the last verified frame sent no tile to k_paint_deep
    // carry pre-pass: slices per tile row and the LDS variant (api.cpp run_paint)
    uint32_t cur_slices = 1, pred_slice_n = 0, pred_max_slice = 0, force_slices = 0;
    // the painters' heaviest-first order (PaintParams::order_*): two sets of {counts, lists}; a read-back-free frame reads the set
    // the last verified frame of the same canvas / crop wrote and writes the other one
    DevBuf order_buf;
    struct OrderSig { uint32_t tiles_w = 0, tiles_h = 0, x0 = 0, x1 = 0, y0 = 0, y1 = 0;
                      bool operator==(const OrderSig& o) const { return tiles_w == o.tiles_w && tiles_h == o.tiles_h && x0 == o.x0 && x1 == o.x1 && y0 == o.y0 && y1 == o.y1; } };
    OrderSig order_sig, order_pending_sig;
    uint32_t order_flat = 0, order_off = 0;       // frames in a row without a tail / frames left with the order switched off
    uint32_t order_thr = 1u << 16, order_tiles = 0;  // shader clocks that make a tile heavy (steered per frame), tiles of the pending frame
    int order_cur = -1, order_pending = -1;       // set with valid lists (-1: none) / set this frame's painter writes (-1: none)
    bool order_enable = false;                    // set by the caller of run_paint for frames that end with k_frame_tail
    const uint32_t* order_cnt_dev = nullptr; uint32_t* order_keep_dev = nullptr;   // what that k_frame_tail copies
    bool cull_on = false;                         // this geometry has had tiles beyond the wave painter's lists: occlusion culling is on
    bool cur_half = false, pred_slice_half = false;   // the 512-lane variant of the small carry kernel (api.cpp run_paint)
    bool cur_small = false, pred_slice_small = false, small_tried = false, small_banned = false, no_small_carry = false;
    bool covl_tried = false, covl_banned = false;   // the COVL carry variant (rows' covers in LDS) was this read-back-free frame's guess / a frame that guessed it was void
    DevBuf ras_masks;                       // k_rasterize: key masks per workgroup (8 words), combined by k_reduce_masks
    PendingMasks pending_masks{nullptr, 0u}; // ... or, on read-back-free frames, by k_runs_count
    size_t n_lines = 0, n_compact = 0;
    // segments
    DevBuf seg_u, seg_a, seg_b, sort_counters;
    uint64_t* sorted = nullptr;
    size_t n_seg = 0;
    bool have_unsorted = false;
    uint64_t live44 = 0xFFFFFFFFFFFull;     // varying bits of (v >> 20)
    bool layer_sorted = false;              // rasterizer stream is non-decreasing in layer
    int digit_bits = 0;                     // radix digit width: 0 = 8, or 9 where that saves a pass (default); 4 / 8 / 9 forced (FORMA_HIP_DEBUG=digit_bits=)
    // paint
    DevBuf info_init;                       // pristine FrameInfo (reset template)
    // buffer-layer caches (reference cpu/buffer/mod.rs:113-197): per cache the CachedTile table, the device image the
    // cache's buffer shows (tiles the painter skips keep last frame's pixels), and the cached clear colour
    struct TileCache {
        DevBuf tiles, image;
        uint32_t w = 0, h = 0;
        bool has_clear = false;
        float clear[4] = {0, 0, 0, 0};
    };
    TileCache caches[32];
    DevBuf pack_list, pack_pix;             // cache frames: written tiles of the crop (list + count word in front), their pixels packed
    DevBuf cache_written;                   // one byte per tile: written this frame
    uint8_t* h_written = nullptr;           // pinned copy of cache_written
    size_t h_written_cap = 0;
    uint8_t* h_stage = nullptr;             // pinned staging image for tile-granular copy-out
    size_t h_stage_cap = 0;
    bool frame_has_dst = false;             // the frame being enqueued on this slot also copies its image out (forma_hip_render_enqueue)
    bool image_sent = false;                // a deferred frame into caller memory: its image left behind the kernels, before the frame was verified
    // A synchronous frame into caller memory (one frame in flight, no cache): the painter runs as two launches (tests: up to SPLIT_MAX) over
    // bands of tile rows, each followed by an event; the bands' copies go out on `copy_stream` behind those events, so the image
    // crosses PCIe while the rest of it is still being painted (api.cpp: run_paint, send_split_bands).
    static constexpr int SPLIT_MAX = 8;
    hipStream_t copy_stream = nullptr;      // created on first use
    hipEvent_t  split_ev[SPLIT_MAX] = {};
    bool     split_want = false;            // set by render_on around the frame's enqueue: this frame may split
    int      split_n = 0;                   // bands of the frame just enqueued (0: not split)
    uint32_t split_row[SPLIT_MAX + 1] = {}; // tile-row boundaries of the bands
    bool     split_sent = false;            // copies are on copy_stream: wait for it before `dst` is touched again
    std::vector<std::pair<void*, size_t>> registered;   // caller buffers pinned by forma_hip_register_buffer
    int cur_cache = -1;                     // cache of the frame in flight
    uint8_t* cur_image = nullptr;           // device image of the frame in flight / last frame
    // sort-plan speculation: the varying-bit mask and the layer-sortedness of a scene rarely change between frames, so
    // forma_hip_render plans the sort from the previous frame's values and verifies them when the frame is done
    bool pred_valid = false, pred_layer_sorted = false, speculated = false;
    bool global_runsort = false;           // FORMA_HIP_DEBUG=global_runsort: never order a row's runs in LDS (test switch)
    bool pred_counts_valid = false, no_async = false;    // N / J predictions for read-back-free frames (FORMA_HIP_DEBUG=sync disables)
    uint32_t pred_N = 0, pred_J = 0, pred_w = 0, pred_h = 0;
    uint64_t pred_live44 = 0;
    KeyRange pred_range{0, 0, 0, 0, false};  // what the tile fields spanned on the last verified frame (value-range digits, SortPlan::bias)
    bool plan_biased = false; uint32_t bias_banned = 0, bias_ban_len = 0;   // bias_banned: frames left of a ban (re-armed with back-off: an animated scene
                                                      // that left its span once gets the cheaper plan back), bias_ban_len: length of the last ban //   // this frame's plan leans on pred_range / a frame that did was void: plain digits for this geometry
    bool ras_hist_on = false; SortPlan ras_plan;     // the rasterizer of this frame counted the digits of ras_plan into the sort's histograms (RasHist)
    const uint32_t* sort_range = nullptr;   // the tile-field spans the frame's sort leaves behind (k_runs_count folds them into FrameInfo) ...
    uint32_t sort_range_n = 0;              // ... one record per k_sort_hist workgroup
    DevBuf info, records, rk_u, rk_a, rk_b, blk_edge, runs_scratch, row_tab, span_key, span_cov, image;
    uint32_t img_w = 0, img_h = 0;
    FrameInfo* h_info = nullptr;            // pinned
    uint32_t*  h_seq = nullptr;             // pinned (behind h_info): the number of the last read-back-free frame whose k_frame_tail has run
    uint32_t   tail_seq = 0;                // ... and of the last one enqueued
    bool info_clean = false;                // the device FrameInfo is pristine: the last frame ended with k_frame_tail (reset_info is then free)
    // what this frame's FIRST kernel cleared on behalf of later stages (ZeroJobs, common.h): consumed by run_sort / run_paint,
    // which clear the words themselves when the pointer or the size is not what they need
    struct PreZero { const void* sort_p = nullptr; size_t sort_words = 0; const void* tab_p = nullptr; size_t tab_words = 0;
                     const void* chain_p = nullptr; size_t chain_words = 0; } pz;
    ForMaDebug dbg;                         // FORMA_HIP_DEBUG as it stood when the context was created
    const uint32_t* chain_rows = nullptr;   // this frame's runs were numbered per tile row (launch_runs' chain): its row counts, for the
    uint32_t n_chain_rows = 0;              //   frame tail, which sums them into the host's n_runs
    uint32_t* h_rows = nullptr;             // pinned: runs per tile row (synchronous frames), 2049 words
    uint32_t pred_max_row = 0xFFFFFFFFu;    // most runs in one tile row of the last verified frame (unknown: no local sort)
    // tiles deeper than the painter's LDS lists (finish_paint): the launch arguments of the frame's painter and the scratch lists
    struct HugeArgs {
        PaintParams P; DevCount jc; TileCacheArgs tc;
        const uint32_t* tile_first_run; const uint32_t* row_span_lo; const uint32_t* row_span_cnt;
        uint32_t* over2_n; uint32_t* over2_list; uint32_t T;
    } huge{};
    DevBuf huge_offs, huge_key, huge_tmp, huge_flag;
    // band
    uint32_t band_row0 = 0, band_row1 = 0;
    // a sub-range [line_lo, line_hi) of the uploaded lines (line i joins points i and i + 1) when line_ranged.
    // A multi-device context uploads the whole geometry to every device and gives each its share of the LINES (multi.cpp).
    bool line_ranged = false;
    size_t line_lo = 0, line_hi = 0;
    // frames in flight inside ONE context (forma_hip_set_frames_in_flight): slots[0] is the context itself, the others are
    // full contexts (own stream, own per-frame buffers) that BORROW the scene buffers.  A device-resident, cache-less frame
    // is enqueued on the next slot and verified when that slot is needed again (or at any call that needs the result).
    forma_hip_ctx* owner = nullptr;         // extra slots: the context they belong to
    std::vector<forma_hip_ctx*> slots;      // of the owner (empty = one frame in flight)
    unsigned next_slot = 0;
    forma_hip_ctx* last = nullptr;          // the slot that holds the most recent frame (inspection calls read it)
    bool pending = false;                   // this slot holds an enqueued frame nobody has verified yet
    struct Deferred {
        uint8_t* dst = nullptr; size_t stride = 0;           // forma_hip_render_enqueue: the frame also travels to caller memory
        uint32_t width = 0, height = 0, bN = 0, bJ = 0;
        uint8_t channels[4] = {0, 1, 2, 3};
        float clear[4] = {0, 0, 0, 0};
        bool has_crop = false;
        forma_rect_t crop = {0, 0, 0, 0};
    } def;
    // several devices behind this context (forma_hip_create_multi): the context is then a shell, the work happens in
    // multi->kid[g] (one full context per device)
    struct MultiState* multi = nullptr;
    // multi-GPU exchange (forma_hip_exchange_plan): owner bands, per-pair capacity, send / receive buckets and their counts
    OwnerBands xbands{};
    uint32_t xcap = 0;
    bool xplanned = false;
    DevBuf xsend, xrecv, xscratch, xmask;         // buckets: n_ranks x (xcap data words + 1 header word {count | overflow << 32})
    bool xpending = false;                    // a deferred owner's half (fd_gsp_defer) nobody has settled yet
    bool xoverflowed = false;                 // the last owner's half failed because a bucket outgrew the plan (FORMA_E_CAPACITY: re-plan)
    bool xuse_recv = false;                   // one rank, but a collective DID run (RCCL rehearsal): the buckets are in xrecv
    bool xgather_always = false;              // FORMA_HIP_DEBUG=xgather: materialise the received stream before sorting it
    bool xpred_valid = false;               // the local rasterized count of the previous exchange frame is known
    uint32_t xpred_N = 0, xpred_w = 0, xpred_h = 0;
    uint32_t* h_xlocal = nullptr;           // pinned: [0] = local segment count of the last bucket frame (copied on the stream), [1] = 1 when pending
    // timing
    hipEvent_t ev0[ST_COUNT], ev1[ST_COUNT];
    KernelTimer kt;                          // per-kernel events of a timed frame (FORMA_LAUNCH, common.h) ...
    float kt_dur_us[KernelTimer::CAP] = {0}, kt_start_us[KernelTimer::CAP] = {0}; int kt_n_done = 0;   // ... resolved by finish_frame
    bool stage_used[ST_COUNT];
    int n_passes = 0;
    uint32_t last_runs = 0, last_entries = 0, last_written = 0;
    // what the last frame wrote, for forma_hip_tiles_written (host-side Flusher / generic Layout::write)
    uint32_t lw_tiles_w = 0, lw_tiles_h = 0, lw_tx0 = 0, lw_tx1 = 0, lw_ty0 = 0, lw_ty1 = 0;
    bool lw_valid = false, lw_cache = false, lw_flags_on_host = false;
};


inline uint32_t paint_strip_tiles(const forma_hip_ctx* c) { return c->n_cus * PAINT_WAVES_PER_CU; }

#define FORMA_RETRY 1     /* internal: a speculation of the read-back-free path was wrong, run the frame again synchronously */

// ---- internals shared by api.cpp and multi.cpp -------------------------------------------------------------------------
int fd_fail(forma_hip_ctx* c, int code, const char* what, hipError_t e = hipSuccess);
// every frame the context still owes (frames in flight) is finished; the first error of a deferred frame is returned
int fd_drain(forma_hip_ctx* ctx);
// restrict the context to lines [lo, hi) of the uploaded geometry (ranged == false: all of them)
int fd_set_line_range(forma_hip_ctx* ctx, bool ranged, size_t lo, size_t hi);
// inclusive prefix sums of the pixel-segment counts of ALL uploaded lines at this canvas size (the reference's `lengths`
// after prefix_sum, segment.rs:90-98,400) — what the line shares of a multi-device plan are cut from
int fd_line_sums(forma_hip_ctx* ctx, uint32_t width, uint32_t height, std::vector<uint32_t>& sums);
// stages 1-2 on the context's line range (synchronous), then pixel segments per tile row -> hist[0 .. 2048)
int fd_row_histogram(forma_hip_ctx* ctx, uint32_t width, uint32_t height, uint32_t* hist /* 2048 */, uint32_t* n_segments);
// forma_hip_gather_sort_paint_frame with a buffer-layer cache
int fd_gather_sort_paint(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes,
                         const uint8_t channels[4], const float clear_color[4], const forma_rect_t* crop_or_null, int cache_id,
                         forma_timings_t* timings);
// the same frame for a multi-device context with frames in flight (device-resident, no cache): enqueue now, settle later
int fd_gsp_defer(forma_hip_ctx* ctx, uint32_t width, uint32_t height, const uint8_t channels[4], const float clear_color[4],
                 const forma_rect_t* crop_or_null);
int fd_gsp_settle(forma_hip_ctx* ctx);
// rows [y0, y1) of the context's last image -> dst (row-major, stride bytes per row, dst addresses row 0)
int fd_copy_image_rows(forma_hip_ctx* ctx, uint8_t* dst, size_t stride_bytes, uint32_t y0, uint32_t y1);
// the frame slot that holds the context's most recent frame (the context itself without frame slots)
forma_hip_ctx* fd_last_slot(forma_hip_ctx* ctx);

// the stream (0 unsorted, 1 sorted) / the written-tile flags of exactly this context's last frame
int fd_read_stream(forma_hip_ctx* ctx, int which, uint64_t* out, size_t capacity, size_t* out_n);
int fd_read_sorted(forma_hip_ctx* ctx, uint64_t* out, size_t capacity, size_t* out_n);
int fd_tiles_written(forma_hip_ctx* ctx, uint8_t* flags, size_t n_tiles);

// multi.cpp: the entry points of a multi-device context (ctx->multi != nullptr)
int  multi_set_frames_in_flight(forma_hip_ctx* ctx, int n);
int  multi_set_layout(forma_hip_ctx* ctx, int layout);
int  multi_sync(forma_hip_ctx* ctx);
void multi_info(forma_hip_ctx* ctx, forma_context_info_t* out);
int  multi_create(forma_hip_ctx** out, const int* devices, int n);
void multi_destroy(forma_hip_ctx* ctx);
int  multi_set_geometry(forma_hip_ctx* ctx, const float* x, const float* y, const uint32_t* line_slot, size_t n_points);
int  multi_set_geoms(forma_hip_ctx* ctx, const forma_geom_t* geoms, size_t n_geoms);
int  multi_set_styles(forma_hip_ctx* ctx, const uint32_t* style_offsets, size_t n_orders, const uint32_t* style_words,
                      size_t n_words, const uint8_t* unchanged);
int  multi_set_images(forma_hip_ctx* ctx, const forma_image_t* images, size_t n_images, const uint16_t* texels, size_t n_texels);
int  multi_render(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes,
                  const uint8_t channels[4], const float clear_color[4], const forma_rect_t* crop_or_null, int cache_id,
                  forma_timings_t* timings);
int  multi_cache_clear(forma_hip_ctx* ctx, int cache_id);
int  multi_trim(forma_hip_ctx* ctx);
int  multi_read_segments(forma_hip_ctx* ctx, int which, uint64_t* out, size_t capacity, size_t* out_n);
int  multi_read_image(forma_hip_ctx* ctx, uint8_t* dst, size_t stride_bytes);
int  multi_tiles_written(forma_hip_ctx* ctx, uint8_t* flags, size_t n_tiles);
forma_hip_ctx* multi_first(forma_hip_ctx* ctx);      // the device-0 context (stage entry points run there)
