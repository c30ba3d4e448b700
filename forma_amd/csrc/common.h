// common.h — internal declarations shared by the HIP translation units of libforma_hip.so.
// Product code: never includes or links anything under oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/forma_hip.h"

// ---- packed pixel segment (reference forma/src/cpu/pixel_segment.rs:36-71, consts.rs:67-94) ----
//  bit 63..53 tile_y+1 (11) | 52..41 tile_x+1 (12) | 40..20 layer (21) | 19..16 local_x | 15..12 local_y
//  | 11..6 double-area multiplier (6) | 5..0 cover (6, two's complement).  Sort key = bits 63..20.
#define SEG_KEY_SHIFT 20
#define SEG_KEY_BITS  44

#if defined(__HIPCC__)
#define FD __device__ __forceinline__
#else
#define FD inline
#endif

FD int      seg_tile_y(uint64_t v) { return (int)(v >> 53) - 1; }
FD int      seg_tile_x(uint64_t v) { return (int)((v >> 41) & 0xFFFu) - 1; }
FD uint32_t seg_layer(uint64_t v) { return (uint32_t)(v >> 20) & 0x1FFFFFu; }
FD int      seg_lx(uint64_t v) { return (int)(v >> 16) & 0xF; }
FD int      seg_ly(uint64_t v) { return (int)(v >> 12) & 0xF; }
FD int      seg_dam(uint64_t v) { return (int)(v >> 6) & 0x3F; }
FD int      seg_cover(uint64_t v) { return ((int)(v & 0x3F) ^ 0x20) - 0x20; }
FD uint64_t seg_key(uint64_t v) { return v >> SEG_KEY_SHIFT; }            // Ord, pixel_segment.rs:161-171
FD uint32_t seg_tile_key(uint64_t v) { return (uint32_t)(v >> 41); }       // (tile_y+1, tile_x+1), 23 bits

// ---- device-side frame header: counters produced by one stage and consumed by the next ----------
struct FrameInfo {
    uint32_t n_segments;     // N (inclusive sum of the last line)
    uint32_t key_or;         // OR / AND of (v >> 20) over all segments, split in two words each:
    uint32_t key_or_hi;      //   used to find constant key digits (single-bin histograms)
    uint32_t key_and;
    uint32_t key_and_hi;
    uint32_t seg_begin;      // first sorted segment with tile_y >= 0
    uint32_t seg_end;        // first sorted segment with tile_y >= tiles_h
    uint32_t n_runs;         // J
    uint32_t n_spans;        // carry-only records
    uint32_t n_entries;      // E
    uint32_t layer_unsorted; // != 0 if the rasterizer stream is not non-decreasing in layer
    uint32_t error;          // device-side invariant violations (1 style, 2 tile depth [unused], 4 look-back spin); 8 = not an
                             // error: tiles deeper than the painter's LDS lists are pending (launch_paint_huge); 16 = not an error: some tile went to k_paint_deep
    uint32_t n_compact;      // lines with at least one pixel segment
    uint32_t plan_bad;       // asynchronous frames: the speculated sort plan does not match this frame's keys
    uint32_t max_row_runs;   // most runs in one tile row (the carry pre-pass sorts a row's runs in LDS when they fit)
    uint32_t exchange_overflow;   // multi-GPU exchange: a bucket did not fit the agreed pair capacity (here or at a sender)
    uint32_t max_slice_runs;      // most runs one workgroup of the carry pre-pass sorted in LDS (a slice of a tile row)
    uint32_t tile_range[4];       // of the sorted keys' tile fields, as maxima: ~min(tile_x + 1), max(tile_x + 1), ~min(tile_y + 1), max(tile_y + 1)
    uint32_t n_heavy;             // tiles the painters filed as heavy this frame (PaintParams::order_*; summed by k_frame_tail)
    uint32_t cost_sum, cost_n;    // ... and a 1-in-256 sample of what a tile cost: shader clocks >> 8, tiles sampled
};

// one run of the sorted stream = one painted (tile, layer) pair that owns pixel segments
struct TileRecord {          // 32 B
    uint32_t cover[4];       // 16 x i8 (little-endian bytes = local_y 0..15): the run's own cover sum out of k_runs,
                             // replaced by its carry-in cover in k_carry_rows
    uint32_t seg_start;      // first segment of the run in the sorted stream
    uint32_t seg_count;
    uint32_t layer;
    uint32_t tile;           // v >> 41 = ((tile_y + 1) << 12) | (tile_x + 1)
};

struct PaintParams {
    uint32_t width, height, tiles_w, tiles_h;
    uint32_t crop_x0, crop_x1, crop_y0, crop_y1;   // in tiles, [x0,x1) x [y0,y1)
    uint32_t channels;                             // 4 x u8 selectors
    float    clear[4];
    uint32_t stride_px;                            // device image row pitch in pixels
    uint32_t scene_has_clips;
    uint32_t scene_simple;          // every layer: solid fill, BlendMode::Over, neither a clip nor clipped
    uint32_t n_orders;
    uint32_t n_words;                              // length of the style word pool
    uint32_t clear_unchanged;                      // buffer-layer cache: this frame's clear colour == the cached one
    uint32_t n_slices;                             // span lists per tile row (slices of the carry pre-pass)
    uint32_t n_groups;                             // tile-column groups per row (SPAN_GROUP_TILES tiles each): see SpanGroups
    // Occlusion culling: a painter drops, while it builds its tile's layer list, every entry below the topmost OCCLUDER that
    // crosses the tile — a carry-only span with a FULL cover of an opaque solid colour, BlendMode::Over, unclipped.  That is
    // what skip_fully_covered_layers_pass (layer_workbench/passes/skip_fully_covered_layers.rs) skips anyway: the same pixels
    // from a list a fraction as long (the 1080p cubic scene: 69 entries per tile, 1.7 painted; lists beyond the wave painter's
    // 128 entries become rare), and k_carry_rows leaves out of a tile-column group's list what an occluder of the whole group
    // hides.  What changes is a tile's layer COUNT, which only a buffer-layer cache remembers (CachedTile), and a clip below the
    // occluder may govern layers above it: 0 on cache frames and for scenes with clips.
    uint32_t cull;
    // Heaviest tiles first (k_paint_wave, one wavefront per tile).  A launch ends with its slowest wavefront, and a tile with
    // many gradient / blend-mode layers lives three times the average: dispatched last it IS the tail of the launch.  A
    // wavefront whose tile took >= order_thr shader clocks therefore appends it to its XCD band's HEAVY list and sets the tile's
    // flag (order_*_out; the eight counts are zeroed with the frame's tile tables and kept by k_frame_tail).  The NEXT frame of
    // the same canvas and crop puts order_hcap workgroups per band in front of the grid: workgroup k of that section paints
    // entry k / PAINT_ORDER_SUBS of the band's list k % PAINT_ORDER_SUBS (or exits: the list is shorter), and the workgroup of
    // a flagged tile in the main section exits.  A
    // schedule only — every tile is painted by exactly one wavefront either way.  Only the heavy tiles (the host steers the
    // threshold towards ~4 % of the tiles, but never below twice the average tile: a flat scene has no tail to hide) pay an atomic: one per tile on a handful of addresses serialises in the L2 (measured:
    // 324 -> 1 570 us on the 8K scene).  order_cnt_out == nullptr: off; order_cnt_in == nullptr: nothing to read yet.
    const uint32_t* order_cnt_in;  const uint32_t* order_list_in;  const uint8_t* order_flag_in;
    uint32_t*       order_cnt_out; uint32_t*       order_list_out; uint8_t*       order_flag_out;
    uint32_t        order_hcap, order_thr;
    // launch_runs' chain numbering (row_base != nullptr): the runs of tile row y are records[row_base[y] .. + row_cnt[y]), and what
    // lies between two rows' ranges is stale — a tile's probes end with its ROW, not with the frame's run count
    const uint32_t* row_base; const uint32_t* row_cnt;
};
// the end of the records a tile of row `ty` may probe
#define PAINT_RUN_END(P, ty, n_runs) ((P).row_base ? (P).row_base[ty] + (P).row_cnt[ty] : (n_runs))

// The spans of a tile row, a second time, by TILE-COLUMN GROUP: k_carry_rows appends to the row's (layer, tile_x)-ordered span
// list one list per group of SPAN_GROUP_TILES tile columns holding ready-made painter entries of the spans that overlap the
// group (ascending layer, like the row list).  A wave painter scans its group's list (~65 entries on the 4K scene) instead
// of the row's (739).  tab[(row * n_slices + slice) * n_groups + g] = {first entry, count}; count = SPAN_GROUP_NONE: the pool
// was full, the painter scans the row list as before.
#ifndef SPAN_GROUP_SHIFT
#define SPAN_GROUP_SHIFT 4
#endif
#define SPAN_GROUP_TILES (1u << SPAN_GROUP_SHIFT)
#define SPAN_GROUP_NONE  0xFFFFFFFFu
#define SPAN_GROUP_MIN_ROW 256u
struct SpanGroups {
    uint2*   tab;
    uint4*   list;        // {key high word (layer | SF_* << 21), entry reference (REF_SPAN | REF_UNCH | span index), lo | hi << 16, 0}
    uint32_t cap;         // entries in `list`: a slice of a row owns [2 * its first run, + 2 * its runs)
    uint32_t min_row;     // a row with no more spans than this keeps only its row list (one round of painter loads either way)
};

// buffer-layer cache (reference cpu/buffer/mod.rs:113-197, painter/mod.rs:629-715 `CachedTile`), device-resident:
// per tile {x = tags (bit0 solid colour valid, bit1 layer count valid) | layer_count << 8, y = solid colour bytes}
struct TileCacheArgs {
    uint2*         tiles;        // nullptr: no cache attached to this frame
    uint8_t*       written;      // one byte per tile: 1 = the painter wrote the tile this frame (TileWriteOp != None)
};
// entry reference word of the painter's layer list (low 32 bits of a key)
#define REF_SPAN 0x80000000u     // index names a span record, else a run record
#define REF_UNCH 0x40000000u     // the layer is unchanged since the cache's previous frame (Layer::is_unchanged)
#define REF_IDX  0x3FFFFFFFu

// A count that lives on the device.  `ptr == nullptr`: the host knows it exactly (= bound).  Otherwise kernels read *ptr
// and clamp it to `bound`, the size their launch grid / buffers were provisioned for: a frame can then be enqueued
// without reading N or J back; the host checks afterwards that nothing exceeded its bound (else it re-runs the frame).
struct DevCount {
    const uint32_t* ptr;
    uint32_t bound;
};
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t dev_count(DevCount c) { return c.ptr ? min(*c.ptr, c.bound) : c.bound; }
#endif

// ---- per-kernel timing of a `timings` frame -------------------------------------------------------
// Every launch of the library goes through FORMA_LAUNCH.  While a timed frame is being enqueued (forma_hip_render with a
// forma_timings_t) the calling thread's g_ktimer points at the context's KernelTimer and the launch carries a pair of events
// (hipExtLaunchKernelGGL: the dispatch's own start / end timestamps — what rocprofv3 reports as the kernel's duration, with no
// marker packets in front of or behind it); otherwise it is a plain launch.  forma_hip_kernel_times returns the list.
// The timer names its context's stream: a frame that failed between stage_begin and stage_end leaves g_ktimer set, and a launch
// of ANOTHER context on the same thread (another stream, maybe another device) must not pick up this context's events.
struct KernelTimer {
    static constexpr int CAP = 96;
    hipEvent_t e0[CAP], e1[CAP];
    const char* name[CAP];
    int stage[CAP];
    int n = 0, made = 0, cur_stage = 0;
    int dropped = 0;                    // launches of the timed frame beyond CAP (they ran untimed: forma_hip_kernel_times says how many)
    hipStream_t stream = nullptr;       // the stream of the context whose frame is being timed: only ITS launches carry events
};
extern thread_local KernelTimer* g_ktimer;
inline bool ktimer_slot(KernelTimer* kt, const char* name, hipEvent_t* e0, hipEvent_t* e1) {
    if (!kt) return false;
    if (kt->n >= KernelTimer::CAP) { kt->dropped++; return false; }
    if (kt->n >= kt->made) {
        if (hipEventCreate(&kt->e0[kt->made]) != hipSuccess) return false;
        if (hipEventCreate(&kt->e1[kt->made]) != hipSuccess) { (void)hipEventDestroy(kt->e0[kt->made]); return false; }
        kt->made++;
    }
    const int i = kt->n++;
    kt->name[i] = name; kt->stage[i] = kt->cur_stage;
    *e0 = kt->e0[i]; *e1 = kt->e1[i];
    return true;
}
#define FORMA_LAUNCH(kern, grid, block, shm, s, ...)                                                        \
    do {                                                                                                    \
        hipEvent_t fl_e0_, fl_e1_;                                                                          \
        if (g_ktimer && g_ktimer->stream == (s) && ktimer_slot(g_ktimer, #kern, &fl_e0_, &fl_e1_))          \
            hipExtLaunchKernelGGL(kern, grid, block, shm, s, fl_e0_, fl_e1_, 0, __VA_ARGS__);               \
        else                                                                                                \
            hipLaunchKernelGGL(kern, grid, block, shm, s, __VA_ARGS__);                                     \
    } while (0)

// ---- kernel launch wrappers (defined in the .hip files) ------------------------------------------
// lines.hip
void launch_prepare_lines(hipStream_t s, const float* x, const float* y, const uint32_t* line_slot, uint32_t n_lines,
                          const forma_geom_t* geoms, uint32_t n_geoms, float width, float height, float band_lo,
                          float band_hi, uint32_t* orders, float* x0, float* y0, float* dx, float* dy, float* a,
                          float* b, float* c, float* d, uint32_t* lengths);
// inclusive scan of u32 (in place) using `tmp` (>= scan_tmp_words(n) u32); total written to *d_total if non-null
size_t scan_tmp_words(size_t n);
void launch_inclusive_scan_u32(hipStream_t s, uint32_t* data, size_t n, uint32_t* tmp, uint32_t* d_total);
void launch_exclusive_scan_u32(hipStream_t s, uint32_t* data, size_t n, uint32_t* tmp, uint32_t* d_total);
// exclusive scan in place of ceil(n / per) values by one workgroup; total -> *d_total
void launch_scan_small_u32(hipStream_t s, uint32_t* data, DevCount n, uint32_t per, uint32_t* d_total);

// The frame path: line lengths -> single-pass scan -> compacted table of the lines that own pixel segments
// (cl_idx = line index, cl_start = index of its first pixel segment) + block_first[b] = compacted line that owns
// pixel segment b * RAS_TILE.  Totals land in info->n_segments / n_compact.
#define RAS_TILE 2048u
// Key masks of a stream as per-workgroup records (8 words: or, or_hi, and, and_hi, layer_unsorted, pad) that nobody has
// combined yet: on read-back-free frames k_runs_count does it (rec == nullptr: info already holds them).
// n_fixed == 0: one record per RAS_TILE block of the info->n_segments segments (k_rasterize); else exactly n_fixed (k_gather_chunks).
// has_range: words 5, 6 of a record hold what the workgroup's keys spanned in tile_x / tile_y (min | max << 16): k_rasterize with
// the sort's histograms fused in (RasHist) — k_runs_count folds them into FrameInfo::tile_range like k_sort_hist's records.
struct PendingMasks { const uint32_t* rec; uint32_t n_fixed; uint32_t has_range; };
struct LineSource {               // either the uploaded geometry (sums == nullptr) or caller-supplied line parameters
    const float* x; const float* y; const uint32_t* line_slot; const forma_geom_t* geoms; uint32_t n_geoms;
    float width, height, band_lo, band_hi;
    const uint32_t* sums;         // inclusive sums of caller-supplied lengths (parity entry point), else nullptr
    const uint32_t* orders; const float *x0, *y0, *dx, *dy, *a, *b, *c, *d;
};
size_t prepare_scratch_words(size_t n_lines);
struct ZeroJobs;
void launch_prepare_compact(hipStream_t s, const LineSource& src, uint32_t n_lines, uint32_t* cl_idx, uint32_t* cl_start,
                            uint32_t* block_first, uint32_t bf_cap, uint32_t* scratch, FrameInfo* info,
                            const ZeroJobs* zero /* nullable: words the first kernel clears for later stages */);
void launch_line_lengths(hipStream_t s, const LineSource& src, uint32_t n_lines, uint32_t* lens, uint32_t* scratch /* prepare_scratch_words */);
// rebuilds block_first when the buffer launch_prepare_compact saw was too small for N
void launch_block_first(hipStream_t s, const uint32_t* cl_start, uint32_t n_compact, uint32_t n_segments,
                        uint32_t* block_first);
// The sort's digit histograms taken where the keys are made (read-back-free frames: the plan is the previous frame's): the
// rasterizer counts the digits of the RH_MAX_PASSES passes of `plan` into the sort's histogram copies (cleared by the frame's
// first kernel), leaves the tile-field span of its keys in its mask record and voids the frame if a biased digit leaves the
// planned span — everything k_sort_hist does, without its read of the whole stream.  hist == nullptr: off.
#define HS_COPIES  16                               // private copies of the sort's digit histograms (added up by k_onesweep)
#define SORT_BINS  512                              // histogram words per pass (a pass has 16, 256 or 512 bins)
#define RH_MAX_PASSES 3
struct RasHist { uint32_t* hist; uint32_t n_passes; uint32_t shift[RH_MAX_PASSES], mask[RH_MAX_PASSES], bias[RH_MAX_PASSES], fmask[RH_MAX_PASSES];
                 uint32_t track_range; /* also measure what the keys' tile fields span (KeyRange for the next frame's plan): only
                                          where a field taken relative to its minimum can save a digit pass — else the records say "unknown" */ };
void launch_rasterize(hipStream_t s, const LineSource& src, DevCount n_compact, DevCount n_segments,
                      const uint32_t* cl_idx, const uint32_t* cl_start, const uint32_t* block_first, uint64_t* out,
                      FrameInfo* info, int band_row0, int band_row1, uint32_t* wg_masks /* 8 words per RAS_TILE block */,
                      bool reduce_now /* false: the caller hands the records on as PendingMasks */,
                      const RasHist* hist = nullptr);
void launch_flatten(hipStream_t s, const forma_flatten_tables_t* dev_tables, float* out_x, float* out_y);

// sort.hip — stable LSB radix sort of u64 (chained-scan "onesweep" passes over the live key bits).
#define SORT_MAX_PASSES 12
struct SortPlan {                 // digit p = ((key >> shift[p]) - bias[p]) & mask[p]
    int      n_passes;
    int      shift[SORT_MAX_PASSES];
    uint32_t mask[SORT_MAX_PASSES];
    uint32_t bias[SORT_MAX_PASSES];   // != 0: the digit is a whole tile field minus its smallest value (KeyRange) ...
    uint32_t fmask[SORT_MAX_PASSES];  // ... and this is the field's mask at `shift` (0: a plain bit window)
};
// What the keys' tile fields spanned on the previous frame.  A canvas of 2^k tiles uses the values 1 .. 2^k of a field that
// stores tile + 1 (0 = left of / above the canvas): k + 1 live bits for one value, and on a 4096- or 8192-pixel canvas that
// bit costs a whole digit pass.  Sorting by (field - min) is the same order in k bits.
struct KeyRange { uint32_t min_x, max_x, min_y, max_y; bool valid; };
// digits packed greedily over the live bits of [lo_bit, hi_bit); digit_bits = 4, 8 or 9 bits per digit, 0 = 8, or 9 where
// that saves a whole pass
SortPlan make_sort_plan(uint64_t live_mask, int lo_bit, int hi_bit, int digit_bits);
// the same for a frame's pixel segments (key = bits 20..63), with the tile fields taken relative to their minima when `range`
// is known and that saves a pass; *biased tells whether it did
SortPlan make_segment_sort_plan(uint64_t live44, bool layer_sorted, int digit_bits, const KeyRange* range, bool* biased);
size_t sort_scratch_words(size_t n);
size_t sort_zero_words(size_t n, const SortPlan& plan);
const uint32_t* sort_range_words(const uint32_t* scratch);  // per k_sort_hist workgroup: {~min tile_x+1, max tile_x+1, ~min tile_y+1, max tile_y+1} of its keys     // leading words of the scratch that must be zero when the sort starts
// `in` is read-only (preserved), a/b are ping-pong buffers; returns the buffer holding the result (== in when the
// plan is empty).  scratch: >= sort_scratch_words(n) u32.  err: device word, bit 2 set if a look-back spin expired.
// Multi-GPU exchange: the stream to sort is NOT contiguous — it is the rank-major concatenation of the n_chunks received
// buckets, bucket q at in[q * (capacity + 1) .. + count_q), its header {count_q | sender overflowed << 32} at + capacity.
// k_sort_hist and the first digit pass read it in place (logical index -> bucket by <= 7 compares), which removes the
// gather kernel; k_sort_hist also publishes the total (info->n_segments), the overflow flag and, per workgroup, the key
// masks / layer-order bit the sort plan is verified with (mask_records, 8 words each: PendingMasks).  n_chunks == 0: plain.
struct ChunkedSrc { const uint64_t* buckets; uint32_t n_chunks; uint32_t capacity; uint32_t* mask_records; };
uint32_t sort_hist_blocks(size_t n);           // grid of k_sort_hist for n keys = number of mask records it writes
const uint64_t* launch_radix_sort(hipStream_t s, const uint64_t* in, uint64_t* a, uint64_t* b, DevCount n,
                                  const SortPlan& plan, int digit_bits, uint32_t* scratch, uint32_t* err,
                                  const ChunkedSrc* chunked = nullptr, FrameInfo* info = nullptr,
                                  bool scratch_is_zero = false /* an earlier kernel of the frame cleared sort_zero_words() */,
                                  bool hist_ready = false /* ... and the producer of the keys counted the digits (RasHist) */,
                                  uint32_t max_workgroups = 0 /* 0: one persistent workgroup per CU; else at most this many */);
RasHist make_ras_hist(const SortPlan& plan, uint32_t* sort_scratch);      // hist == nullptr if the plan does not qualify

// exchange.hip — multi-GPU: bucket a rank's pixel segments by tile-row owner, gather what the owner received
#define FORMA_MAX_RANKS 8
struct OwnerBands { uint32_t n; uint32_t edge[FORMA_MAX_RANKS + 1]; };   // rank g owns tile rows [edge[g], edge[g + 1])
size_t owner_scratch_words(size_t n);
// stable partition of seg[0 .. n) into send[g * (capacity + 1) + ...]; bucket header at + capacity: segments for rank g | overflow << 32
void launch_owner_bucket(hipStream_t s, const uint64_t* seg, DevCount n, const OwnerBands& B, uint32_t capacity,
                         uint32_t* scratch, uint64_t* send, FrameInfo* info);
void launch_row_histogram(hipStream_t s, const uint64_t* seg, DevCount n, uint32_t* hist /* 2048 words */);
// recv[s * (capacity + 1) + ...] (count in the bucket's header) -> out, info->{n_segments, key masks, layer_unsorted}
size_t gather_mask_words(uint32_t n_ranks, uint32_t capacity);
void launch_gather_chunks(hipStream_t s, const uint64_t* recv, uint32_t n_ranks, uint32_t capacity,
                          uint64_t* out, FrameInfo* info, uint32_t* mask_records /* gather_mask_words() */, bool reduce_now);

// paint.hip
struct BlkEdge {             // what a k_runs tile contributes to a run that started before it
    uint32_t cov[4];         // cover sum (16 x i8) of the segments before the tile's first key change
    uint32_t cnt;            // their number
    uint32_t has_boundary;   // 0: the whole tile is one run's interior
    uint32_t pad[2];
};
size_t runs_scratch_words(size_t n);
size_t runs_blocks(size_t n);
// run detection + per-run cover sums; row_tab = [row_count | row_span_lo | row_span_cnt], (tiles_h + 1) words each
// What the run kernel attaches to a record beyond the geometry — it runs on the whole chip, k_carry_rows on one CU per tile
// row, where every scattered access per run is a cycle of that CU's address unit: the layer's style bits (SF_*, bits 21.. of
// the record's layer word) and "unchanged" flag (bit 31 of its tile word), and a 32-bit digest per run, in stream order,
// that lets the carry pre-pass order a row and find the tile columns WITHOUT touching the records:
//   run_lt[j] = (layer & 0xFFFF) << 16 | open << 15 | tile_x + 1        (open: the run continues past its chunk)
struct RunStyle {
    const uint32_t* layer_sf;     // per order: SF_* | LSF_VALID
    uint32_t        n_orders;
    const uint8_t*  unchanged;    // per order, nullable (Layer::is_unchanged(cache_id) of this frame's cache)
    uint32_t*       run_lt;       // one word per run (rec_cap)
};
#define RUN_LT_OPEN 0x8000u
#define RUN_LT_NEWTILE 0x4000u          // BLOCKS numbering only: the run is the first of its tile
// launch_runs' BLOCKS numbering (round 6; paint.hip): the run kernel wrote SPARSE records / digests — tile b of 2 048 segments owns
// [2048 b, 2048 b + heads[b]) — and k_carry_rows copies every row's runs into the dense arrays.  rec_sp == nullptr: not this frame.
struct BlkRuns {
    const TileRecord* rec_sp;       // n.bound records, sparse
    const uint32_t*   run_lt_sp;    // n.bound digests, sparse
    const uint32_t*   heads;        // paintable run heads per 2 048-segment tile (the head of launch_runs' `scratch`)
    const uint32_t*   row_sp;       // index of each row's first run in the sparse numbering (launch_runs' chain_row_base of that frame)
    uint32_t*         row_base_out; // ... in the dense one, written by the row's workgroup (PaintParams::row_base)
    uint32_t*         tile_first_run;
    uint32_t*         run_lt_out;   // (unused: the dense digests)
    uint32_t          round_tiles;  // tiles of the run kernel a round of k_carry_rows' table takes: 256 (tests: a power of two below)
};
// tables_are_zero: the frame's tile tables (row_tab_zero_words() words of row_tab) were cleared by an earlier kernel of this
// frame; else k_runs_count clears them
void launch_runs(hipStream_t s, const uint64_t* sorted, DevCount n, uint32_t tiles_w, uint32_t tiles_h, TileRecord* records,
                 uint32_t rec_cap, uint64_t* run_keys, uint32_t* tile_first_run, BlkEdge* blk_edge,
                 uint32_t* row_tab, uint32_t* scratch, FrameInfo* info, bool verify_plan, uint64_t spec_live44,
                 bool spec_layer_sorted, PendingMasks pm, RunStyle rs, bool tables_are_zero,
                 const uint32_t* range_records /* nullable: sort_range_words() of the sort that produced `sorted` ... */,
                 uint32_t n_range_records /* ... and sort_hist_blocks() of its key count */,
                 int what = 3 /* bit 0: k_runs_count (per-tile head counts into `scratch`), bit 1: k_runs_wave (the records) — a
                                 synchronous frame reads the count back in between and sizes the records for it */,
                 uint32_t* chain_row_base = nullptr /* tiles_h + 1 words.  Not null: ONE kernel, no counting pass — runs are numbered per
                                 tile row from the index of the row's first segment (here: where each row begins), `records` and
                                 the arrays indexed like it hold n.bound entries, the run count is the sum of the row counts */,
                 bool chain_status_is_zero = false /* the first runs_chain_words(n.bound) words of `scratch` were cleared by an earlier kernel */,
                 bool blocks = false /* with chain_row_base: numbered per 2 048-segment tile instead (BlkRuns): no look-back, `records` and
                                 rs.run_lt are the sparse arrays, the head of `scratch` takes the tiles' head counts */);
size_t runs_chain_words(size_t n);
// the per-tile counts k_runs_count leaves in `scratch`: how many (host sum = J), or — big frames — already scanned (J = info->n_runs)
uint32_t runs_count_tiles(size_t n, bool* scanned);
uint32_t runs_edge_segments();            // segments per BlkEdge entry
// The end of a read-back-free frame: the device-side FrameInfo goes to pinned host memory (`host_info`, nullable) and/or its
// segment count to a pinned word (`host_count`, nullable), and the device copy returns to its pristine state for the next
// frame of the stream — one tiny kernel instead of a device-to-host copy here and a device-to-device reset there.
void launch_frame_tail(hipStream_t s, FrameInfo* info, FrameInfo* host_info, uint32_t* host_count,
                       const uint32_t* order_cnt = nullptr, uint32_t* order_keep = nullptr /* PAINT_ORDER_WORDS words copied (PaintParams::order_*) */,
                       const uint32_t* chain_rows = nullptr, uint32_t n_chain_rows = 0 /* launch_runs' chain numbering: the frame's row
                       counts, summed into the host copy's n_runs */,
                       uint32_t* host_seq = nullptr, uint32_t seq = 0 /* pinned word that takes `seq` when everything else has landed */);
// Words the frame's FIRST kernel clears on behalf of later stages (sort scratch, tile tables): a few hundred
// KB spread over a grid that exists anyway, instead of two or three memset operations on the stream.
#define FORMA_ZERO_JOBS 4
struct ZeroJobs { uint32_t* p[FORMA_ZERO_JOBS]; uint32_t words[FORMA_ZERO_JOBS]; uint32_t n; };
// cache frames: list of the written tiles of the crop (row-major) + their pixels packed into 1 KB slots, <= max_pack tiles
void launch_pack_written(hipStream_t s, const uint8_t* written, uint32_t tiles_w, uint32_t tx0, uint32_t tx1, uint32_t ty0, uint32_t ty1,
                         uint32_t* list, uint32_t* count, uint32_t max_pack, const uint8_t* image, uint32_t width, uint32_t height,
                         uint32_t* packed);   // segments per BlkEdge entry of the kernel launch_runs picks
// style flags of a layer as the carry pre-pass and the painter pass them around (bits 21.. of a record's layer word)
#define SF_FULL        0x001u     // spans only: Cover::is_full (painter/mod.rs:200-215)
#define SF_IS_CLIP     0x002u
#define SF_CLIPPED     0x004u
#define SF_OPAQUE      0x008u     // solid fill with alpha == 1
#define SF_EVENODD     0x010u
#define SF_BLEND_SHIFT 5          // 4 bits: ordinal of BlendMode
#define SF_FILL_SHIFT  9          // 2 bits: fill type
#define LSF_VALID      0x80000000u // layer_sf[] entry: the order has a style (else FORMA_NONE)

uint32_t carry_rows_local_cap();      // most runs a workgroup of launch_carry_rows(local_sort = true) sorts in LDS: large variant ...
uint32_t carry_rows_small_cap();      // ... small variant (several workgroups per CU)
uint32_t carry_rows_half_cap();       // ... its 512-lane form
uint32_t carry_rows_covl_cap();       // ... the large variant with the covers in LDS (one slice per row)
// Tiles per XCD "band" of the wave painters' grid (workgroup b runs on XCD b % 8).  PAINT_ROW_XCD = 1: band x = the tile rows x,
// x + 8, ... of the crop: the rows of a frame are spread over all eight XCDs from the first workgroup on (1/8 band of the 4K
// scene: painter 55.5 -> 48.1 us, whole frame 106.4 -> 103.9; dealing k_runs_wave's tiles to the same XCD as their row — so that the
// records would be read from the L2 they were written to — changed nothing: the gain is the dispatch order, not cache affinity);
// 0: eight contiguous bands of tiles (rounds 1-4).
#ifndef PAINT_ROW_XCD
#define PAINT_ROW_XCD 1
#endif
static inline uint32_t paint_band_tiles(uint32_t rows, uint32_t tiles_w) {
#if PAINT_ROW_XCD
    return ((rows + 7u) / 8u) * tiles_w;
#else
    return (rows * tiles_w + 7u) / 8u;
#endif
}
#define PAINT_ORDER_SUBS  64u          // heavy lists per XCD band of the painters' order (PaintParams::order_*): appends spread over
#define PAINT_ORDER_WORDS (8u * PAINT_ORDER_SUBS)   // 512 counters — one address takes ~150 ns per returning atomic, one after the other
#define CR_MAX_SLICES_HOST 8u         // workgroups that may share one tile row
// the frame's tile tables, one buffer: [row_count: tiles_h + 1][row_span_lo: 8 tiles_h + 1][row_span_cnt: 8 tiles_h + 1]
// [painter overflow counters: 2][first-run table: T][painter order counts: PAINT_ORDER_WORDS] — zeroed every frame by launch_runs —
// then [overflow list: T][{tile, entries}: 2 T]
// ... then [where each row's runs begin (launch_runs' chain numbering): tiles_h + 1]
#ifndef RUNS_BLK_DEFAULT
#define RUNS_BLK_DEFAULT 1              // read-back-free frames with one carry workgroup per tile row number their runs per 2 048-segment tile (BlkRuns; debug.h: runs_blk)
#endif
#ifndef RUNS_CHAIN_DEFAULT
#define RUNS_CHAIN_DEFAULT 1            // read-back-free frames number their runs per tile row, without the counting pass (debug.h: runs_chain) ...
#endif
#ifndef RUNS_CHAIN_MAX_TILES
#define RUNS_CHAIN_MAX_TILES 1536u      // ... when the stream is at most this many 2 048-segment tiles, i.e. about one round of the run kernel's workgroups
// MEMORY: a chained frame indexes its run arrays like its segments, so records (32 B), run_lt (4), span_key (8), span_cov (16) and
// the group pool (2 x 16) are provisioned for the segment bound N instead of the run bound J: 92 B per segment, ~290 MB per
// frame slot at this limit (3.1 M segments), kept — like every per-frame buffer — until forma_hip_trim; a context that
// alternates between chained and counted frames keeps the larger set.
#endif
static inline size_t row_tab_total_words(uint32_t tiles_w, uint32_t tiles_h);
static inline uint32_t row_tab_zero_words(uint32_t tiles_w, uint32_t tiles_h) { return (tiles_h + 1) + 2 * (CR_MAX_SLICES_HOST * tiles_h + 1) + 3 + tiles_w * tiles_h + PAINT_ORDER_WORDS; }
static inline size_t row_tab_total_words(uint32_t tiles_w, uint32_t tiles_h) { return (size_t)row_tab_zero_words(tiles_w, tiles_h) + 5 * (size_t)tiles_w * tiles_h + tiles_h + 1; }
// n_slices workgroups per tile row (each a range of layers, 256 bins of layer >> bin_shift); small: the CR_CAP_S variant
void launch_carry_rows(hipStream_t s, bool local_sort, bool small, bool half /* with small: 512-lane workgroups, slices of <= 2048 runs */,
                       uint32_t n_slices, uint32_t bin_shift,
                       const uint64_t* sorted_run_keys, TileRecord* records,
                       const BlkEdge* blk_edge, DevCount n_segments, DevCount n_runs,
                       const uint32_t* layer_sf /* per order: SF_* | LSF_VALID */, uint32_t n_orders, uint32_t tiles_w,
                       uint32_t tiles_h, const uint32_t* row_count, uint32_t* row_span_lo, uint32_t* row_span_cnt,
                       uint64_t* span_key, uint4* span_cov, const uint8_t* unchanged /* per order, nullable */, FrameInfo* info,
                       uint32_t edge_segs,
                       uint32_t vis_last /* visible pixel rows of the last tile row (height % 16, 16 if 0) */,
                       uint32_t row0, uint32_t row1 /* the tile rows that are painted (the crop): only those get workgroups */,
                       SpanGroups groups /* tab == nullptr: no group lists */, const uint32_t* run_lt /* RunStyle::run_lt */,
                       bool cull /* PaintParams::cull */,
                       uint32_t left_start /* cache frames: the first painted tile column, whose tiles list every layer with segments to
                                              their left (painter/mod.rs:500-522); 0xFFFFFFFF: no cache and a channel order under which a folded tile and a painted one
                                              are the same bytes — nothing can observe the entry */,
                       const uint32_t* row_base = nullptr /* launch_runs' chain numbering: where each row's runs begin */,
                       bool covl = false /* one slice per row, rows of <= carry_rows_covl_cap() runs, neither small nor half: the variant that
                                            brings the row's cover sums and style summaries into LDS before the walk */,
                       BlkRuns bk = BlkRuns{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 256u} /* a COVL variant: local_sort, ONE slice per row */);
void launch_paint(hipStream_t s, const PaintParams& p, const uint64_t* sorted, const TileRecord* records, DevCount n_runs,
                  const uint32_t* tile_first_run, const uint32_t* row_span_lo, const uint32_t* row_span_cnt,
                  const uint64_t* span_key, const uint4* span_cov,
                  const uint4* layer_col /* per order: style words 2..5 (clip: word 1) */,
                  const uint32_t* style_offsets, const uint32_t* style_words, const forma_image_t* images,
                  const uint16_t* texels, uint8_t* image, TileCacheArgs cache, FrameInfo* info, uint32_t* overflow_n /* zeroed by launch_runs */,
                  uint32_t* overflow_list /* tiles_w * tiles_h words */, uint32_t* over2_n /* zeroed by launch_runs */,
                  uint32_t* over2_list /* {tile, entries} pairs: 2 * tiles_w * tiles_h words */,
                  bool launch_deep /* false: k_paint_deep is not launched; a tile that needs it voids the frame (plan_bad) */,
                  SpanGroups groups /* tab == nullptr: the painters scan the row lists (p.n_groups is ignored) */,
                  bool strips = false /* four wavefronts per tile, each a 16 x 4 strip (k_paint_wave<.., NPX = 1>): frames that do not
                                         fill the chip with one wavefront per tile; ignored with a buffer-layer cache */,
                  bool quads = false /* four tiles per wavefront (k_paint_quad): all-solid scenes with shallow tiles; ignored otherwise */,
                  uint32_t* mid_n = nullptr /* zeroed by launch_runs */, uint32_t* mid_list = nullptr /* {tile, entries} pairs of the tiles beyond
                                         k_paint_deep's 1024-entry tier: 2 * tiles_w * tiles_h words (both required with launch_deep) */,
                  uint32_t n_cus = 256);
// tiles whose layer list exceeds the painter's LDS lists (info->error bit 3 after launch_paint): lists in global memory,
// offs[i] = first entry slot of tile over2_list[2 i]; g_key holds 4 entries per slot, g_tmp / g_flag one
void launch_paint_huge(hipStream_t s, const PaintParams& p, const uint64_t* sorted, const TileRecord* records, DevCount n_runs,
                       const uint32_t* tile_first_run, const uint32_t* row_span_lo, const uint32_t* row_span_cnt,
                       const uint64_t* span_key, const uint4* span_cov, const uint4* layer_col,
                       const uint32_t* style_offsets, const uint32_t* style_words, const forma_image_t* images,
                       const uint16_t* texels, uint8_t* image, TileCacheArgs cache, FrameInfo* info, const uint32_t* over2_list,
                       uint32_t n_tiles, const uint64_t* offs, uint64_t* g_key, uint64_t* g_tmp, uint32_t* g_flag);
