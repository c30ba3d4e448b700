/*
 * forma_hip.h — C ABI of libforma_hip.so, the MI355X (gfx950) backend for forma's
 * 4-stage raster pipeline (flatten -> pixel-grid rasterize -> sort -> per-tile paint).
 *
 * This is the drop-in boundary.  It replaces what
 * `forma::cpu::Renderer::render` (reference forma/src/cpu/renderer.rs:75-224) does between
 * "the composition's geometry store + layer table" and "the caller's RGBA8 buffer":
 *
 *   data in  = SegmentBuffer {x, y, ids}            reference forma/src/segment.rs:529-555
 *              geom_id_to_order + InnerLayer         reference forma/src/composition/state.rs:30,
 *                                                    forma/src/composition/layer.rs:26-41
 *              Props per Order                       reference forma/src/styling.rs:438-442
 *   data out = caller-owned `&mut [u8]` + LinearLayout{width, width_stride, height}
 *                                                    reference forma/src/cpu/buffer/layout/mod.rs:167-222
 *
 * Conventions
 *   - every entry point returns 0 on success and a negative FORMA_E_* code on failure; nothing
 *     ever unwinds across the ABI (the reference panics; a Rust shim maps !=0 to panic!).
 *   - all host pointers are borrowed for the duration of the call only.
 *   - a context is single-threaded (the reference takes `&mut self`, renderer.rs:75); distinct
 *     contexts may be used concurrently.  One HIP stream per context.
 *   - plain pointers and sizes only; no torch / C++ types.
 */
#ifndef FORMA_HIP_H
#define FORMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- limits (reference forma/src/consts.rs:25-26,107-109) ---------------------------- */
#define FORMA_MAX_WIDTH   65536u
#define FORMA_MAX_HEIGHT  32768u
#define FORMA_LAYER_LIMIT 0x1FFFFFu          /* 2^21 - 1 */
#define FORMA_TILE        16u                /* CPU tile 16x16, consts.rs:31-43 */
#define FORMA_NONE        0xFFFFFFFFu

/* ---- error codes ---------------------------------------------------------------------- */
#define FORMA_OK             0
#define FORMA_E_ARG         -1   /* invalid argument (size limits, null pointer, stride)    */
#define FORMA_E_HIP         -2   /* a HIP runtime call failed; see forma_hip_last_error    */
#define FORMA_E_NO_DEVICE   -3   /* no gfx950 device visible                                */
#define FORMA_E_CAPACITY    -4   /* caller-provided output capacity too small               */
#define FORMA_E_STATE       -5   /* call sequence error (e.g. render before set_geometry)   */
#define FORMA_E_INTERNAL    -6   /* device-side invariant violated (bounded spin expired…)  */
#define FORMA_E_COMM        -7   /* multi-device context: an RCCL call failed / librccl could not be loaded */

typedef struct forma_hip_ctx forma_hip_ctx;

/* ---- scene tables ---------------------------------------------------------------------- */

/* One entry per geometry slot.  A slot is the device-side stand-in of a reference `GeomId`
 * (segment.rs:100-131): `line_slot[i]` names the slot of the line point i -> point i+1, or
 * FORMA_NONE for the gap between two polygonal chains (reference `ids[i] == None`).
 * The entry folds `geom_id_to_order` and `InnerLayer` (segment.rs:141-149, layer.rs:26-41). */
typedef struct forma_geom_t {
    uint32_t order;    /* FORMA_NONE: geom not in the composition, layer disabled, order None */
    uint32_t flags;    /* bit0: has_transform                                                  */
    float    xf[6];    /* ux, uy, vx, vy, tx, ty  (AffineTransform::to_array, transform.rs:50) */
} forma_geom_t;
#define FORMA_GEOM_HAS_XF 1u

/* Style table: `style_offsets[order]` indexes `style_words` (FORMA_NONE = order unused).
 *
 *  word0 header  bits 0..3   blend mode, ordinal of reference `BlendMode` (styling.rs:390-408)
 *                bits 4..5   fill type: 0 solid, 1 linear gradient, 2 radial gradient, 3 texture
 *                bit  6      fill rule: 0 NonZero, 1 EvenOdd                 (styling.rs:63-67)
 *                bit  7      is_clipped                                      (styling.rs:416)
 *                bit  8      func: 0 Draw, 1 Clip                            (styling.rs:421-428)
 *                bits 16..31 gradient stop count
 *  word1         n of Func::Clip(n), 0 for Draw
 *  solid:        word2..5   r g b a (f32 bits)
 *  gradient:     word2..5   start.x start.y end.x end.y, then 5 words per stop: r g b a stop
 *  texture:      word2..7   ux uy vx vy tx ty (screen -> texture space), word8 image index
 */
#define FORMA_STYLE_BLEND(h)      ((h) & 0xFu)
#define FORMA_STYLE_FILL(h)       (((h) >> 4) & 0x3u)
#define FORMA_STYLE_EVENODD(h)    (((h) >> 6) & 0x1u)
#define FORMA_STYLE_CLIPPED(h)    (((h) >> 7) & 0x1u)
#define FORMA_STYLE_IS_CLIP(h)    (((h) >> 8) & 0x1u)
#define FORMA_STYLE_STOPS(h)      ((h) >> 16)
#define FORMA_FILL_SOLID   0u
#define FORMA_FILL_LINEAR  1u
#define FORMA_FILL_RADIAL  2u
#define FORMA_FILL_TEXTURE 3u

/* Image table entry; texels are 4 x u16 in the reference's bias-shifted half format
 * (`f16`, styling.rs:224-259), row-major, in one pool.                                  */
typedef struct forma_image_t {
    uint64_t texel_offset;   /* first texel of this image in the pool (in texels) */
    uint32_t width;
    uint32_t height;
} forma_image_t;

/* Output channel selectors, reference forma/src/cpu/channel.rs:34-62. */
enum { FORMA_CH_RED = 0, FORMA_CH_GREEN = 1, FORMA_CH_BLUE = 2, FORMA_CH_ALPHA = 3,
       FORMA_CH_ZERO = 4, FORMA_CH_ONE = 5 };

/* Crop rectangle in PIXELS; rounded out to the tile grid exactly like `Rect::new`
 * (renderer.rs:43-52). */
typedef struct forma_rect_t { uint32_t x0, x1, y0, y1; } forma_rect_t;

/* Per-stage device times of the last render, in microseconds (HIP events on the context's
 * stream).  Same split the reference exposes: trace spans cpu/renderer.rs:171-199 and
 * gpu `Timings`, gpu/renderer/mod.rs:24-36.                                              */
typedef struct forma_timings_t {
    float    prepare_us;      /* prepare_lines + prefix sum            */
    float    rasterize_us;    /* pixel-grid intersector                */
    float    sort_us;         /* all radix passes                      */
    float    sort_pass_us;    /* average of one scatter pass           */
    float    carry_us;        /* runs + cover-carry pre-pass + tile lists */
    float    paint_us;        /* per-tile painter                      */
    float    total_us;        /* first kernel start -> last kernel end */
    float    d2h_us;          /* image copy into caller memory (0 if dst == NULL) */
    uint32_t n_lines;
    uint32_t n_segments;      /* N = number of pixel segments          */
    uint32_t n_sort_passes;   /* digit passes actually executed        */
    uint32_t n_runs;          /* (tile, layer) runs in the sorted stream */
    uint32_t n_tile_entries;  /* carry-only span records of the frame  */
    uint32_t n_tiles_written; /* tiles copied into dst (all tiles of the crop without a cache; the damaged ones with one) */
    float    exchange_us;     /* multi-GPU: bucketing by tile-row owner (sender) + gathering the received buckets (owner) */
} forma_timings_t;

/* ---- lifetime ----------------------------------------------------------------------------- */
/* `device` is a HIP device ordinal (one process per GPU: pass LOCAL_RANK). */
int  forma_hip_create(forma_hip_ctx** out, int device);
/* ONE context over `n` devices of this process: what `Renderer::new()` (cpu/renderer.rs:63-65)
 * becomes when the renderer owns several GPUs.  The scene calls (set_geometry / set_geoms / set_styles / set_images) and
 * forma_hip_render keep their signatures and their contract — `dst` is fully written when render returns — and the
 * whole multi-GPU frame happens inside the library, one host thread per device:
 *   every device holds the scene and rasterizes ITS share of the lines (equal pixel-segment counts, cut from the prefix
 *   sums of the line lengths) -> HIP kernels bucket the pixel segments by the device that owns their tile row -> ONE
 *   all-to-all over xGMI (RCCL: ncclCommInitAll over the devices, grouped ncclAllToAll of the padded buckets, each with its
 *   {count, overflow} header as last word, on the devices' streams, no host synchronisation) -> every device sorts and paints its band of tile rows and
 *   copies its rows straight into `dst` (disjoint row ranges, no gather collective).
 * Bands (equal pixel-segment counts per device), line shares and bucket capacities are planned on the first frame of a
 * geometry / canvas size and re-planned when a bucket outgrows its capacity.  Buffer-layer caches, crops, channel orders
 * and forma_hip_read_image / _read_segments(1) / _tiles_written work as on one device; forma_hip_read_segments(1) returns
 * the sorted stream of the PAINTED tile rows (segments above or below the canvas are dropped before the exchange, they
 * are never painted: painter/mod.rs:731-734).  The stage entry points (flatten, prepare_lines, rasterize, sort, paint)
 * run on devices[0]; the single-device plumbing of the process-per-GPU layout (set_band, *_frame, exchange_*) returns
 * FORMA_E_STATE.  librccl is loaded on first use (dlopen), so single-device users never map it.
 * n == 1 is forma_hip_create(devices[0]).  A device may be listed more than once: the "devices" are then contexts on the
 * same GPU and the all-to-all is done with device-to-device copies instead of RCCL — a rehearsal mode for single-GPU
 * machines (tests), not a way to go faster.  n <= FORMA_MAX_DEVICES. */
#define FORMA_MAX_DEVICES 8
int  forma_hip_create_multi(forma_hip_ctx** out, const int* devices, int n);
void forma_hip_destroy(forma_hip_ctx* ctx);
const char* forma_hip_last_error(const forma_hip_ctx* ctx);
/* Library self-description: "forma_hip <version> gfx950". */
const char* forma_hip_version(void);

/* ---- scene upload (reference data crossing: SURVEY §8b row 3) ------------------------------ */
/* Replace the device-resident geometry store.  n_points >= 0; line_slot has max(n_points-1,0)
 * entries.  Replaces SegmentBufferView.{x,y,ids} (segment.rs:529-555).                        */
int forma_hip_set_geometry(forma_hip_ctx* ctx, const float* x, const float* y,
                           const uint32_t* line_slot, size_t n_points);
int forma_hip_set_geoms(forma_hip_ctx* ctx, const forma_geom_t* geoms, size_t n_geoms);
/* `unchanged` may be NULL (= every layer changed); else one byte per order, the reference's
 * `Layer::is_unchanged(cache_id)` (layer.rs:179-189).                                         */
int forma_hip_set_styles(forma_hip_ctx* ctx, const uint32_t* style_offsets, size_t n_orders,
                         const uint32_t* style_words, size_t n_words,
                         const uint8_t* unchanged);
int forma_hip_set_images(forma_hip_ctx* ctx, const forma_image_t* images, size_t n_images,
                         const uint16_t* texels, size_t n_texels);

/* ---- stage 1: curve flattening (reference path.rs:473-538 `Primitives::into_segments`) ----- */
/* Work items are produced by the host-side sequential pass (path.rs:252-445); all index
 * arrays are absolute into the quad / spline tables.  Appends nothing to the context: pure
 * map, results copied back to `out_x/out_y` (n_points each).                                  */
typedef struct forma_flatten_tables_t {
    /* per output point */
    const uint32_t* point_commands;  /* PointCommand bits, path.rs:137-168                    */
    const uint32_t* point_indices;   /* pi                                                    */
    const uint32_t* quad_indices;    /* qi                                                    */
    size_t          n_points;
    /* per quad (3 control points each in qx/qy/qw) */
    const float* qx; const float* qy; const float* qw;
    const float* x0; const float* dx_recip; const float* k0; const float* dk;
    const float* curvatures_recip;
    const uint32_t* partial_spline;  /* partial_curvatures[i].0                               */
    const float*    partial_curv;    /* partial_curvatures[i].1                               */
    size_t          n_quads;
    /* per spline */
    const float* sp0x; const float* sp0y; const float* sp2x; const float* sp2y;
    size_t       n_splines;
} forma_flatten_tables_t;
int forma_hip_flatten(forma_hip_ctx* ctx, const forma_flatten_tables_t* t,
                      float* out_x, float* out_y);

/* ---- stage entry points for parity tests (host arrays in, host arrays out) ----------------- */
/* prepare_lines = SegmentBuffer::fill_cpu_view (segment.rs:275-402) on the uploaded geometry.
 * Each output array has n_points-1 entries; `lengths` holds the INCLUSIVE prefix sums exactly
 * like the reference (segment.rs:90-98,400).                                                   */
int forma_hip_prepare_lines(forma_hip_ctx* ctx, uint32_t width, uint32_t height,
                            uint32_t* orders, float* x0, float* y0, float* dx, float* dy,
                            float* a, float* b, float* c, float* d, uint32_t* lengths);
/* rasterize = Rasterizer::rasterize (cpu/rasterizer.rs:92-159) from caller-supplied line
 * parameters; writes the unsorted u64 stream in reference order (line, then i).              */
int forma_hip_rasterize(forma_hip_ctx* ctx, size_t n_lines,
                        const uint32_t* orders, const float* x0, const float* y0,
                        const float* dx, const float* dy, const float* a, const float* b,
                        const float* c, const float* d, const uint32_t* lengths,
                        uint64_t* out_segments, size_t capacity, size_t* out_n);
/* sort = Rasterizer::sort (cpu/rasterizer.rs:161-164): stable LSB radix on key bits 20..63
 * (`PixelSegment::cmp`, pixel_segment.rs:161-171).  In place on the host array.
 * digit_bits: 4 or 8; 0 = library default.                                                    */
int forma_hip_sort(forma_hip_ctx* ctx, uint64_t* segments, size_t n, int digit_bits);
/* paint = painter::for_each_row (cpu/painter/mod.rs:717-778) over a caller-supplied SORTED
 * stream with the uploaded styles.  dst/stride/channels/clear/crop as forma_hip_render.        */
int forma_hip_paint(forma_hip_ctx* ctx, const uint64_t* sorted_segments, size_t n,
                    uint8_t* dst, uint32_t width, uint32_t height, size_t stride_bytes,
                    const uint8_t channels[4], const float clear_color[4],
                    const forma_rect_t* crop_or_null);

/* ---- the frame: cpu::Renderer::render (renderer.rs:75-224) ---------------------------------- */
/* dst == NULL leaves the image device-resident (read it with forma_hip_read_image).
 * cache_id >= 0 selects one of 32 buffer-layer caches (renderer.rs:68-73, buffer/mod.rs:113-197): the
 * device keeps the per-tile CachedTile state, the cached clear colour and the image the cache's
 * buffer shows; tiles the optimizer passes skip (TileWriteOp::None) are NOT written into `dst`.
 * -1 = no cache.  Only the crop rectangle (tile-rounded) is ever written.  `timings` may be NULL.  */
int forma_hip_render(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height,
                     size_t stride_bytes, const uint8_t channels[4],
                     const float clear_color[4], const forma_rect_t* crop_or_null,
                     int cache_id, forma_timings_t* timings);
/* forma_hip_render into caller memory WITHOUT waiting for the frame.  With n > 1 frame slots (forma_hip_set_frames_in_flight)
 * the frame — kernels AND the copy of its rows into `dst`, band by band under the painters — is enqueued on the next slot and the
 * call returns: the 33 MB of a 4K frame cross PCIe while the next frames are rasterized, sorted and painted.  `dst` is complete
 * (and must not be touched before) once forma_hip_sync has returned or n further frames have been enqueued; a host that shows
 * frames hands over n + 1 buffers in turn.  Register the buffers (below): a copy into pageable memory holds the calling thread
 * until it is done and nothing overlaps.  With one frame slot, on a multi-device context, the call is forma_hip_render.
 * (The reference's render is synchronous, cpu/renderer.rs:75-84: this is the one place where the drop-in offers more, and the
 * shim keeps `render` itself on forma_hip_render.) */
int forma_hip_render_enqueue(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height,
                             size_t stride_bytes, const uint8_t channels[4],
                             const float clear_color[4], const forma_rect_t* crop_or_null);
/* Page-lock a caller buffer the renderer writes often (hipHostRegister): copies into it are asynchronous and run at the
 * link's rate.  The caller keeps the memory alive until forma_hip_unregister_buffer (which waits for frames in flight) or
 * forma_hip_destroy. */
int forma_hip_register_buffer(forma_hip_ctx* ctx, void* ptr, size_t bytes);
int forma_hip_unregister_buffer(forma_hip_ctx* ctx, void* ptr);
/* Drop a cache's tile state (BufferLayerCache::clear, buffer/mod.rs:189-196). */
int forma_hip_cache_clear(forma_hip_ctx* ctx, int cache_id);

/* Frames in flight inside ONE context.  The kernels of a frame are bound by their own dependent round trips, not by a
 * chip-wide resource, so a second frame on another HIP stream fills the gaps (+25-30 % frames/s).  With n > 1 the
 * context keeps n frame slots (own stream and per-frame buffers, ONE shared scene); a forma_hip_render call that leaves
 * the image on the device (dst == NULL), uses no buffer-layer cache and asks for no timings ENQUEUES its frame on the
 * next slot and returns.  The frame is verified — and re-run if a speculation of the read-back-free path failed — when
 * its slot comes round again or when any call needs its result: forma_hip_read_image / _read_segments / _tiles_written
 * refer to the most recent frame, every scene upload and forma_hip_sync wait for all of them.  An error of a deferred
 * frame is returned by the call that completes it.  Frames with dst != NULL keep the reference's contract (the caller's
 * buffer is fully written when render returns, cpu/buffer/mod.rs:43-49); cache frames stay in order (frame k + 1 reads
 * what frame k left in the cache).  Default 1: every render call is complete when it returns.  Three slots are the
 * measured optimum on one device.  On a multi-device context (forma_hip_create_multi) every device gets n slots and every
 * slot its own communicators; a frame whose exchange buckets outgrew the plan is re-planned and re-run when it is settled. */
#define FORMA_MAX_FRAMES_IN_FLIGHT 8
int forma_hip_set_frames_in_flight(forma_hip_ctx* ctx, int n);
/* Wait for every enqueued frame; returns the first error any of them produced. */
int forma_hip_sync(forma_hip_ctx* ctx);
/* What a context is made of: its devices, its frame slots and — for a multi-device context — how pixel segments travel
 * between the devices (RCCL all-to-all over xGMI, or peer copies when librccl cannot be loaded / initialised; a context over
 * one device exchanges nothing).  A renderer has no counterpart in the reference (its parallelism is a Rayon pool,
 * cpu/renderer.rs:61-73); hosts and benchmarks report it. */
#define FORMA_TRANSPORT_NONE 0u
#define FORMA_TRANSPORT_RCCL 1u
#define FORMA_TRANSPORT_COPY 2u
typedef struct forma_context_info {
    uint32_t n_devices;
    uint32_t frames_in_flight;
    uint32_t transport;                 /* FORMA_TRANSPORT_* (NONE: one device, or the BANDS layout) */
    uint32_t layout;                    /* FORMA_LAYOUT_EXCHANGE / _BANDS: what the last plan of a multi-device context was made for */
    int32_t  devices[FORMA_MAX_DEVICES];
} forma_context_info_t;
int forma_hip_context_info(forma_hip_ctx* ctx, forma_context_info_t* out);
/* How a multi-device context (forma_hip_create_multi) splits a frame.  The tile rows are always cut into one band per device
 * (the reference's own unit of parallelism: painter/mod.rs:741-776 paints tile rows in parallel, and the cover carry never
 * crosses rows, :518-522); the layouts differ in how a band's pixel segments reach its device:
 *   FORMA_LAYOUT_EXCHANGE  every device rasterizes 1/G of the LINES; HIP kernels bucket the pixel segments by the device that
 *                          owns their tile row and ONE all-to-all (RCCL over xGMI) delivers them;
 *   FORMA_LAYOUT_BANDS     no exchange: every device holds the whole scene, prepares all lines but culls them to its band in
 *                          the frame's first kernel, and renders its band like a single device (section 8e's "replicate stages
 *                          1-2, keep only own-band segments: zero communication");
 *   FORMA_LAYOUT_AUTO      (default) the cheaper of the two for the scene, decided whenever a plan is made: BANDS unless the
 *                          scene has more lines than pixel segments.
 * Both paint bit-identical images.  Frames in flight are settled first; the next frame plans anew.  FORMA_E_STATE on a
 * single-device context. */
#define FORMA_LAYOUT_AUTO     0
#define FORMA_LAYOUT_EXCHANGE 1
#define FORMA_LAYOUT_BANDS    2
int forma_hip_multi_layout(forma_hip_ctx* ctx, int layout);
/* Introspection, host logic only (no device is touched): the digit plan the library makes for a frame's pixel-segment sort
 * (reference: one `par_sort_unstable` on the 44 key bits, cpu/rasterizer.rs:161-164, pixel_segment.rs:161-171).  Pass p is a
 * stable counting pass on digit = ((segment >> shift[p]) - bias[p]) & mask[p], least significant pass first.
 * live_key_bits: bit i set = bit 20 + i of the segments varies in the stream (OR ^ AND of the keys); layer_sorted: the stream
 * is non-decreasing in layer (the layer digits are dropped); digit_bits: 0 (8, or 9 where that saves a pass), 4, 8 or 9;
 * field_range = {min tile_x + 1, max tile_x + 1, min tile_y + 1, max tile_y + 1} as stored in the keys, or NULL: with it a tile
 * field may be sorted relative to its minimum (one digit of bit_length(max - min) bits) where that saves a pass.
 * The CPU test suite checks the property that matters: the passes, applied as stable sorts, order any keys with those live bits
 * and that range exactly like a stable sort on bits 20..63. */
#define FORMA_SORT_MAX_PASSES 12
typedef struct forma_sort_plan {
    uint32_t n_passes;
    uint32_t biased;                    /* some pass uses a non-zero bias (needs field_range to hold for every key) */
    uint32_t shift[FORMA_SORT_MAX_PASSES], mask[FORMA_SORT_MAX_PASSES], bias[FORMA_SORT_MAX_PASSES];
} forma_sort_plan_t;
int forma_hip_sort_plan(uint64_t live_key_bits, int layer_sorted, int digit_bits, const uint32_t* field_range, forma_sort_plan_t* out);
/* Give per-frame device memory back (it is grown to the largest frame seen and otherwise kept until destroy): streams,
 * records, tables and the scratch image of the context and of its frame slots.  The scene and the buffer-layer caches stay.
 * forma_hip_read_image returns FORMA_E_STATE and forma_hip_read_segments an empty stream until the next render.  (The reference's renderer owns Vecs
 * that keep their capacity the same way, cpu/renderer.rs:61-73; this is the shrink_to_fit it never needed on the host.) */
int forma_hip_trim(forma_hip_ctx* ctx);

/* ---- inspection of the last render (parity tests at full size, bench) ---------------------- */
/* The kernels of the last forma_hip_render call that asked for timings, in launch order: each launch carried its own pair of
 * events (the dispatch's start / end timestamps, i.e. what `rocprofv3 --kernel-trace` reports), so `us` holds no marker or
 * launch overhead; forma_timings_t's stage times are sums of these.  `stage`: 0 prepare, 1 rasterize, 2 sort, 3 carry (runs +
 * cover carry), 4 paint, 6 exchange.  `start_us` is relative to the first kernel's start.  Writes min(count, capacity) entries,
 * *out_n = count.  At most 96 launches of a frame are timed; a frame with more still reports its full count (the launches
 * beyond ran untimed: *out_n > 96 says the stage sums are short by them).  Single-device contexts (a multi-device context returns FORMA_E_STATE).  No reference counterpart: forma
 * times its stages on the host (`duration!`, cpu/renderer.rs:106-223). */
typedef struct forma_kernel_time_t {
    char     name[48];        /* kernel name without template arguments, NUL-terminated */
    float    start_us;
    float    us;
    uint32_t stage;
    uint32_t reserved;
} forma_kernel_time_t;
int forma_hip_kernel_times(forma_hip_ctx* ctx, forma_kernel_time_t* out, size_t capacity, size_t* out_n);
/* which: 0 = unsorted stream (rasterizer order), 1 = sorted stream. */
int forma_hip_read_segments(forma_hip_ctx* ctx, int which, uint64_t* out, size_t capacity,
                            size_t* out_n);
int forma_hip_read_image(forma_hip_ctx* ctx, uint8_t* dst, size_t stride_bytes);

/* Which tiles the last forma_hip_render / _paint / _sort_paint_frame call wrote (TileWriteOp != None inside the crop):
 * one byte per tile, row-major [tiles_h][tiles_w], tiles = ceil(size / 16).  This is what lets the host side run the
 * reference's per-tile hooks after the one strided copy: `Flusher::flush` on every row slice of every written tile
 * and a user `Layout::write` (cpu/buffer/layout/mod.rs:29-34, 51-163, 264-295; painter/mod.rs:537-548).          */
int forma_hip_tiles_written(forma_hip_ctx* ctx, uint8_t* flags, size_t n_tiles);

/* ---- multi-GPU: tile-row band ownership ---------------------------------------- */
/* Restrict this context to tile rows [row0, row1): prepare_lines additionally culls lines
 * entirely outside the band and the rasterizer drops pixel segments of other bands, so the
 * sort and the painter only see the owner's segments.  row1 == 0 resets to the full canvas.    */
int forma_hip_set_band(forma_hip_ctx* ctx, uint32_t row0, uint32_t row1);
/* Device pointers for the exchange step (RCCL all-to-all is driven from the host language
 * through torch.distributed on these buffers).  Valid until the next render/ingest call.        */
int forma_hip_segments_device(forma_hip_ctx* ctx, int which, uint64_t** dev_ptr, size_t* n);
/* Run stages 1-2 only (prepare + rasterize), leaving the unsorted stream on the device. */
int forma_hip_rasterize_frame(forma_hip_ctx* ctx, uint32_t width, uint32_t height,
                              forma_timings_t* timings);
/* Sort + paint a device-resident unsorted stream of n segments (after an exchange the caller
 * wrote into the buffer returned by forma_hip_reserve_segments).                                */
int forma_hip_reserve_segments(forma_hip_ctx* ctx, size_t n, uint64_t** dev_ptr);
int forma_hip_sort_paint_frame(forma_hip_ctx* ctx, size_t n, uint8_t* dst, uint32_t width,
                               uint32_t height, size_t stride_bytes,
                               const uint8_t channels[4], const float clear_color[4],
                               const forma_rect_t* crop_or_null, forma_timings_t* timings);


/* ---- multi-GPU, exchange layout: rasterize 1/G of the lines per GPU, ONE all-to-all of pixel segments over xGMI --------
 * One process per GPU, every rank holds the layer / style tables and ITS contiguous share of the lines
 * (forma_hip_set_geometry with a slice).  Rank g owns the tile rows [row_edges[g], row_edges[g + 1]).  Per frame:
 *   1. forma_hip_rasterize_bucket_frame: stages 1-2 on the local lines, then a stable partition of the local pixel
 *      segments by owner into `send`.  A bucket is words_per_pair = pair_capacity + 1 u64: bucket g's segments at
 *      send[g * words_per_pair ...], its header at send[g * words_per_pair + pair_capacity] = count | overflow << 32
 *      (overflow: this sender exceeded the capacity, or rasterized more than a read-back-free frame provisioned — every
 *      receiver then fails the frame with FORMA_E_CAPACITY and the host re-plans with a larger capacity);
 *   2. the host language runs ONE collective on forma_hip_stream's stream: an equal-split all-to-all of words_per_pair
 *      u64 per pair from send into recv (the headers travel with the data) — RCCL through torch.distributed in this
 *      repository; with one rank nothing is exchanged;
 *   3. forma_hip_gather_sort_paint_frame: the received buckets, rank-major (= global line order, which keeps the sort
 *      bit-exact), become one stream; sort + paint of the band as in forma_hip_render (crop = the band).
 * No call waits for the device except the end of step 3 (and step 1 when timings are requested).                        */
int forma_hip_stream(forma_hip_ctx* ctx, void** hip_stream);
int forma_hip_exchange_plan(forma_hip_ctx* ctx, const uint32_t* row_edges /* n_ranks + 1 */, uint32_t n_ranks /* <= 8 */,
                            uint32_t pair_capacity);
int forma_hip_exchange_buffers(forma_hip_ctx* ctx, uint64_t** send, uint64_t** recv, size_t* words_per_pair);
int forma_hip_rasterize_bucket_frame(forma_hip_ctx* ctx, uint32_t width, uint32_t height, forma_timings_t* timings);
int forma_hip_gather_sort_paint_frame(forma_hip_ctx* ctx, uint8_t* dst, uint32_t width, uint32_t height,
                                      size_t stride_bytes, const uint8_t channels[4], const float clear_color[4],
                                      const forma_rect_t* crop_or_null, forma_timings_t* timings);

#ifdef __cplusplus
}
#endif
#endif /* FORMA_HIP_H */
