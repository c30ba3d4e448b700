// paint.hip — stage 4 for gfx950: run detection, the cover-carry pre-pass that makes tiles
// independent, per-tile layer lists and the one-workgroup-per-16x16-tile painter.
//
// Reference behaviour restated: painter::for_each_row / paint_tile_row / LayerWorkbench::
// drive_tile_painting / Painter::paint_layer (reference forma/src/cpu/painter/mod.rs:290-347,
// 485-568, 717-778; layer_workbench/mod.rs:213-342; passes/*.rs; cpu/painter/styling.rs).
// The reference serialises a tile row because every tile needs, per layer, the cover accumulated
// by all tiles to its left (the "cover carry").  Here the carry is a data-parallel pre-pass over the
// sorted stream, so the painter runs one workgroup per tile: a 256-lane workgroup owns the 256
// pixels, accumulates the tile's segments into LDS cells with ds_add, forms the signed cover prefix
// along x with DPP row shifts (a DPP row is exactly one 16-pixel tile row) and blends in fp32.
// Built with -ffp-contract=off; every fused op is an explicit fmaf where the reference has mul_add.
#include "common.h"
#include "lookback.h"
#include "radix_rank.h"
#include "debug.h"

// ================================================================================================
// helpers
// ================================================================================================
__device__ __forceinline__ uint64_t swar_add8(uint64_t a, uint64_t b) {      // 8 wrapping i8 adds
    return ((a & 0x7F7F7F7F7F7F7F7Full) + (b & 0x7F7F7F7F7F7F7F7Full)) ^ ((a ^ b) & 0x8080808080808080ull);
}
// add one segment's cover to the 16 x i8 accumulator (lo = local_y 0..7, hi = 8..15): acc_cover semantics of
// LayerWorkbench::cover_carry (layer_workbench/mod.rs:213-234)
__device__ __forceinline__ void acc_seg_cover(uint64_t v, uint64_t& lo, uint64_t& hi) {
    const int ly = seg_ly(v);
    const uint64_t add = (uint64_t)((uint32_t)seg_cover(v) & 0xFFu) << ((ly & 7) * 8);
    if (ly < 8) lo = swar_add8(lo, add); else hi = swar_add8(hi, add);
}
__device__ __forceinline__ bool seg_paintable(uint64_t v, uint32_t tiles_w, uint32_t tiles_h) {
    // painter/mod.rs:731-734 drops tile_y < 0; rows >= tiles_h and tiles right of the canvas are never visited
    // (:524-538); tile_x == -1 (stored 0) is the left-of-canvas bucket that only feeds the carry (:500-522)
    const uint32_t tyb = (uint32_t)(v >> 53), txb = (uint32_t)(v >> 41) & 0xFFFu;
    return tyb >= 1u && tyb <= tiles_h && txb <= tiles_w;
}
__device__ __forceinline__ bool cover_is_empty(uint64_t lo, uint64_t hi, bool even_odd) {   // painter/mod.rs:187-198
    if (!even_odd) return (lo | hi) == 0;
    return ((lo | hi) & 0x1F1F1F1F1F1F1F1Full) == 0;      // all (|c| & 31) == 0  <=>  c mod 32 == 0 for every byte
}

// style facts carried in bits 21..31 of a run record's `layer` word and of a span key's high word (= bits 53..63 of
// the painter's entry keys): enough for LayerWorkbench's optimizer passes to classify a layer without the style table
#define LAYER_MASK     0x1FFFFFu

// An OCCLUDER: a carry-only span whose cover is full, of an opaque solid colour, blended Over, neither a clip nor clipped —
// what skip_fully_covered_layers_pass looks for (layer_workbench/passes/skip_fully_covered_layers.rs).  `khi` = the high word of
// a span key / group-list entry: layer | SF_* << 21.
__device__ __forceinline__ bool span_is_occluder(uint32_t khi) {
    // FULL and OPAQUE set; IS_CLIP, CLIPPED, the blend ordinal and the fill type (FORMA_FILL_SOLID = 0) all zero; EVENODD free
    constexpr uint32_t care = (SF_FULL | SF_IS_CLIP | SF_CLIPPED | SF_OPAQUE | (15u << SF_BLEND_SHIFT) | (3u << SF_FILL_SHIFT)) << 21;
    static_assert(FORMA_FILL_SOLID == 0, "the occluder test reads the fill type as a zero field");
    return (khi & care) == ((SF_FULL | SF_OPAQUE) << 21);
}

__device__ __forceinline__ bool cover_full(const uint32_t* c, bool even_odd) {     // Cover::is_full painter/mod.rs:200-215
    uint32_t ok = 1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t w = c[i];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int8_t v = (int8_t)(w >> (8 * b));
            int8_t a = (int8_t)(v < 0 ? -v : v);          // abs(-128) = -128 like _mm_abs_epi8
            ok &= even_odd ? ((a & 31) == 16) : (a == 16);
        }
    }
    return ok != 0;
}

// ================================================================================================
// runs: maximal runs of equal 44-bit key (tile_y, tile_x, layer) in the sorted stream, found in ONE
// read of the stream.  Per run: record {first segment, count, layer, tile}, the wrapping i8 cover sum per
// pixel row (what LayerWorkbench::cover_carry accumulates), the (tile_y, layer | run) key the global run
// sort is ordered by and a 32-bit digest for the in-LDS one.
//
// A wavefront owns RW_CHUNK = 512 consecutive segments, lane l the 8 consecutive segments [8 l, 8 l + 8), and
// works in its own slice of LDS with no workgroup barrier in its loop:
//   * key changes ("breaks") are numbered along the wave (one DPP scan of the lanes' break counts): segment i
//     belongs to slot = number of breaks at or before i; slot 0 is the run an earlier chunk opened;
//   * every segment does ONE ds_add of its cover into bins[slot][local_y] of the wave's slice; the lane that
//     owns a break parks the run's start next to the bins;
//   * then one lane per slot packs the 16 bins to 16 x i8 and writes the record — all runs of the chunk
//     in one full-width step, whatever their length;
//   * slot 0 goes to the chunk's BlkEdge, and a run still open at the end of a full chunk is flagged RUN_OPEN:
//     the consumer (k_carry_rows) completes it from the following chunks' edges — no wave waits for a later one.
// A chunk with more than RW_SLOTS runs sweeps again.
//
// Runs are numbered DENSELY in stream order (the painters read a tile's runs as consecutive records, the carry
// pre-pass a tile row's as one range): k_runs_count reads the stream once to count the paintable run heads per chunk, every
// workgroup of k_runs_wave sums the counts in front of it.  (Round 4 built the single-pass form — every workgroup publishes
// its head count right after the break walk and looks back through its predecessors' status words — and measured it: 118 us
// against 22 + 51 on the 4K scene.  A workgroup lives ~15 us and 2048 of them are resident: the window of predecessors that
// have published an aggregate but no prefix grows until the walk is the kernel.  The chain needs few, long-lived tiles —
// the radix sort's 840 x 16384 keys — not 6720 short ones.)
// ================================================================================================
#define RN_TILE    2048                 // segments per workgroup of the run kernel (4 waves x RW_CHUNK)
#ifndef BLK_STRIDE
#define BLK_STRIDE RN_TILE              // BLOCKS numbering: index range a tile of the run kernel owns (experiments: a smaller stride = a smaller footprint)
#endif
#define RN_STRIDE  17                   // words per slot: 16 bins + 1 pad (bank spread)
#define RUN_OPEN   0x80000000u          // seg_count flag: the run continues past its wave's chunk
#define RN_ROWS    64                   // tile rows a workgroup aggregates in LDS before touching row_count[]

__device__ __forceinline__ uint4 pack_bins(const int* b) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        w[k] = ((uint32_t)b[4 * k] & 0xFFu) | (((uint32_t)b[4 * k + 1] & 0xFFu) << 8) | (((uint32_t)b[4 * k + 2] & 0xFFu) << 16) |
               ((uint32_t)b[4 * k + 3] << 24);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// what a run kernel adds to the record of a run of `layer` (RunStyle, common.h): the words stored as the record's layer and tile
__device__ __forceinline__ uint2 run_style_words(const RunStyle& rs, uint32_t layer, uint32_t tile) {
    const uint32_t lsf = layer < rs.n_orders ? rs.layer_sf[layer] : 0u;
    const uint32_t sfl = (lsf & LSF_VALID) ? (lsf & ~LSF_VALID) : 0u;   // (a layer without a style: k_carry_rows reports it)
    const uint32_t unch = (rs.unchanged && layer < rs.n_orders && rs.unchanged[layer]) ? 0x80000000u : 0u;
    return make_uint2(layer | (sfl << 21), tile | unch);
}

#define RW_SEGS    8
#define RW_CHUNK   (64 * RW_SEGS)
#define RW_WAVES   4
#define RW_THREADS (64 * RW_WAVES)
#define RW_SLOTS   64                   // runs per sweep = lanes
static_assert(RW_WAVES * RW_CHUNK == RN_TILE, "a workgroup of the run kernel owns RN_TILE segments");

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {            // lane l gets lane l - 1's value, lane 0 gets 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ void wave_lds_fence() {                      // LDS operations of one wave retire in order: compiler fence
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct RunWaveLds {                       // one wave's slice
    int      bins[RW_SLOTS * RN_STRIDE];  // [slot][local_y], 17-word rows (bank spread)
    uint32_t start[RW_SLOTS + 2];         // chunk-local index of the first segment of the sweep's slots (+ the end of the last one)
};

// What one workgroup does per frame besides its share of the stream — k_runs_count's workgroup 0, or the extra workgroup of the
// chained k_runs_wave: the guard on the provisioned count, the tile-field spans (next frame's sort plan), the key masks of the
// stream (read-back-free frames: one record per producer workgroup, combined here instead of by a launch) and the check of the
// speculated sort plan.  `first` = this is that workgroup.
#define RC_THREADS 256
__device__ __forceinline__ void runs_housekeeping(const bool first, const uint32_t n, const DevCount nc, FrameInfo* __restrict__ info,
                                                  const uint64_t spec_live44, const uint32_t spec_flags, const PendingMasks pm,
                                                  const uint32_t* __restrict__ range_records, const uint32_t n_range_records,
                                                  uint32_t (*s_red)[RC_THREADS / 64]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    {
        if (first && tid == 0 && nc.ptr && *nc.ptr > nc.bound) info->plan_bad = 1u;   // more segments than provisioned
        if (first && range_records) {                         // what the tile fields spanned (next frame's sort plan)
            uint4 m = make_uint4(0u, 0u, 0u, 0u);
            for (uint32_t b = tid; b < n_range_records; b += RC_THREADS) {
                const uint4 r = *reinterpret_cast<const uint4*>(range_records + (size_t)b * 4);
                m.x = max(m.x, r.x); m.y = max(m.y, r.y); m.z = max(m.z, r.z); m.w = max(m.w, r.w);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                m.x = max(m.x, (uint32_t)__shfl_xor(m.x, d, 64)); m.y = max(m.y, (uint32_t)__shfl_xor(m.y, d, 64));
                m.z = max(m.z, (uint32_t)__shfl_xor(m.z, d, 64)); m.w = max(m.w, (uint32_t)__shfl_xor(m.w, d, 64));
            }
            if (lane == 0) { atomicMax(&info->tile_range[0], m.x); atomicMax(&info->tile_range[1], m.y); atomicMax(&info->tile_range[2], m.z); atomicMax(&info->tile_range[3], m.w); }
        }
        if (first) {
            // the key masks of the stream: on read-back-free frames the producer (k_rasterize / k_gather_chunks) left one
            // record per workgroup and nobody needed them combined until now — this workgroup does it instead of a launch
            uint32_t o = 0, oh = 0, a = 0xFFFFFFFFu, ah = 0xFFFFFFFFu, u = 0;
            uint32_t r_nmin_x = 0, r_max_x = 0, r_nmin_y = 0, r_max_y = 0;     // pm.has_range: the producer's tile-field spans
            if (pm.rec) {
                // (the CLAMPED count: the producer wrote one record per block of the provisioned bound, and a frame whose true
                //  count exceeds it is voided by plan_bad above — reading info->n_segments here walked past the buffer)
                const uint32_t nrec = pm.n_fixed ? pm.n_fixed : (n + RAS_TILE - 1) / RAS_TILE;
                // four records (two 16-byte loads each) in flight per lane: the registers of this loop must stay below what the
                // streaming loop needs — at eight records the kernel lost a wave per SIMD and 16 us
                for (uint32_t b0 = 0; b0 < nrec; b0 += 4 * RC_THREADS) {
                    uint4 m[4], m2[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t b = b0 + k * RC_THREADS + tid;
                        m[k] = b < nrec ? *reinterpret_cast<const uint4*>(pm.rec + (size_t)b * 8) : make_uint4(0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu);
                        m2[k] = b < nrec ? *reinterpret_cast<const uint4*>(pm.rec + (size_t)b * 8 + 4) : make_uint4(0u, 0x0000FFFFu, 0x0000FFFFu, 0u);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        o |= m[k].x; oh |= m[k].y; a &= m[k].z; ah &= m[k].w; u |= m2[k].x;
                        if (pm.has_range) {                                        // words 5, 6: min | max << 16 of tile_x, tile_y
                            r_nmin_x = max(r_nmin_x, ~(m2[k].y & 0xFFFFu)); r_max_x = max(r_max_x, m2[k].y >> 16);
                            r_nmin_y = max(r_nmin_y, ~(m2[k].z & 0xFFFFu)); r_max_y = max(r_max_y, m2[k].z >> 16);
                        }
                    }
                }
                if (pm.has_range) {
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) {
                        r_nmin_x = max(r_nmin_x, (uint32_t)__shfl_xor(r_nmin_x, d, 64)); r_max_x = max(r_max_x, (uint32_t)__shfl_xor(r_max_x, d, 64));
                        r_nmin_y = max(r_nmin_y, (uint32_t)__shfl_xor(r_nmin_y, d, 64)); r_max_y = max(r_max_y, (uint32_t)__shfl_xor(r_max_y, d, 64));
                    }
                    if (lane == 0 && nrec) {
                        atomicMax(&info->tile_range[0], r_nmin_x); atomicMax(&info->tile_range[1], r_max_x);
                        atomicMax(&info->tile_range[2], r_nmin_y); atomicMax(&info->tile_range[3], r_max_y);
                    }
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    o |= __shfl_xor(o, d, 64); oh |= __shfl_xor(oh, d, 64); a &= __shfl_xor(a, d, 64); ah &= __shfl_xor(ah, d, 64);
                    u |= __shfl_xor(u, d, 64);
                }
                if (lane == 0) { s_red[0][w] = o; s_red[1][w] = oh; s_red[2][w] = a; s_red[3][w] = ah; s_red[4][w] = u; }
                __syncthreads();
                if (tid == 0 && info->n_segments) {
                    for (int i = 1; i < RC_THREADS / 64; i++) { o |= s_red[0][i]; oh |= s_red[1][i]; a &= s_red[2][i]; ah &= s_red[3][i]; u |= s_red[4][i]; }
                    info->key_or = o; info->key_or_hi = oh; info->key_and = a; info->key_and_hi = ah; info->layer_unsorted = u;
                }
            } else if (tid == 0) {
                o = info->key_or; oh = info->key_or_hi; a = info->key_and; ah = info->key_and_hi; u = info->layer_unsorted;
            }
            if (tid == 0 && (spec_flags & 1u) && info->n_segments) {
                const uint64_t k_or = (uint64_t)o | ((uint64_t)oh << 32), k_and = (uint64_t)a | ((uint64_t)ah << 32);
                const uint32_t sorted_now = u == 0 ? 2u : 0u;
                if (((k_or ^ k_and) & 0xFFFFFFFFFFFull) != spec_live44 || sorted_now != (spec_flags & 2u)) info->plan_bad = 1u;
            }
        }
    }
}

// pass A: paintable run heads per tile (the run index of a tile's first head is the exclusive scan of these)
__global__ __launch_bounds__(RC_THREADS) void k_runs_count(const uint64_t* __restrict__ sorted, DevCount nc, uint32_t tiles_w,
                                                           uint32_t tiles_h, uint32_t* __restrict__ counts,
                                                           uint32_t* __restrict__ chunk_counts /* per 512 segments */,
                                                           uint32_t* __restrict__ zero_base, uint32_t zero_words,
                                                           FrameInfo* __restrict__ info, uint64_t spec_live44,
                                                           uint32_t spec_flags /* bit0 verify, bit1 layer_sorted */,
                                                           PendingMasks pm, const uint32_t* __restrict__ range_records /* nullable: the
                                                           sort's tile-field spans, one 4-word record per k_sort_hist workgroup */,
                                                           uint32_t n_range_records) {
    __shared__ uint32_t s_c[RC_THREADS / 64];
    __shared__ uint32_t s_red[5][RC_THREADS / 64];
    const uint32_t n = dev_count(nc);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // housekeeping that would otherwise be separate launches: zero this frame's tile tables (row counts, span tables,
    // painter overflow counter, first-run table: consumed by k_runs and later), verify the speculated sort plan
    {
        const uint32_t per = (zero_words + gridDim.x - 1) / gridDim.x;
        const uint32_t z0 = blockIdx.x * per, z1 = min(zero_words, z0 + per);
        for (uint32_t i = z0 + tid; i < z1; i += RC_THREADS) zero_base[i] = 0;
    }
    runs_housekeeping(blockIdx.x == 0, n, nc, info, spec_live44, spec_flags, pm, range_records, n_range_records, s_red);
    // A pure streaming read, shaped like the copy kernels that reach 6 TB/s on this chip (tools/ubench_bw.hip): 256-lane
    // workgroups in a grid-stride loop, eight 16-byte loads (two consecutive segments each) in flight per lane.  One
    // iteration covers two k_runs tiles (waves 0-1 and 2-3).
    // Workgroup 0 of a grid of several only keeps house: its round trips (records -> reductions -> FrameInfo) would otherwise
    // sit in front of its share of the stream, and the kernel ends with its slowest workgroup.
    if (gridDim.x > 1 && blockIdx.x == 0) return;
    const uint32_t sblocks = gridDim.x > 1 ? gridDim.x - 1 : 1u, sb = gridDim.x > 1 ? blockIdx.x - 1 : 0u;
    const uint4* s4 = (const uint4*)sorted;                              // hipMalloc'd: 16-byte aligned
    const uint32_t ntiles = (n + RN_TILE - 1) / RN_TILE;
    for (uint32_t t0 = sb * 2; t0 < ntiles; t0 += sblocks * 2) {
        const uint32_t wbase = t0 * RN_TILE + w * 1024;                   // this wave's 1024 consecutive segments
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i0 = wbase + q * 128 + lane * 2;
            v[q] = i0 + 1 < n ? s4[i0 >> 1] : make_uint4(0, 0, 0, 0);
            if (i0 + 1 == n) { const uint64_t k = sorted[i0]; v[q].x = (uint32_t)k; v[q].y = (uint32_t)(k >> 32); }
        }
        uint64_t before = (wbase > 0 && wbase < n) ? sorted[wbase - 1] : 0ull;   // the segment in front of the wave's piece
        uint32_t c = 0, c_first = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i0 = wbase + q * 128 + lane * 2;
            const uint64_t k0 = (uint64_t)v[q].x | ((uint64_t)v[q].y << 32), k1 = (uint64_t)v[q].z | ((uint64_t)v[q].w << 32);
            uint64_t pk = __shfl_up(k1, 1, 64);
            if (lane == 0) pk = before;
            before = __shfl(k1, 63, 64);
            const bool h0 = i0 < n && ((((k0 ^ pk) >> SEG_KEY_SHIFT) != 0) || i0 == 0) && seg_paintable(k0, tiles_w, tiles_h);
            const bool h1 = i0 + 1 < n && (((k1 ^ k0) >> SEG_KEY_SHIFT) != 0) && seg_paintable(k1, tiles_w, tiles_h);
            c += (uint32_t)__popcll(__ballot(h0)) + (uint32_t)__popcll(__ballot(h1));
            if (q == 3) { c_first = c; }                                 // the wave's first 512 segments = one k_runs_wave chunk
        }
        if (lane == 0) {
            s_c[w] = c;
            if (wbase < n) { chunk_counts[wbase >> 9] = c_first; chunk_counts[(wbase >> 9) + 1] = c - c_first; }
        }
        __syncthreads();
        if (tid < 2 && t0 + tid < ntiles) counts[t0 + tid] = s_c[2 * tid] + s_c[2 * tid + 1];
        __syncthreads();
    }
}

// CHAIN (read-back-free frames whose rows are ordered inside k_carry_rows): there is no counting pass.  Runs are numbered per
// TILE ROW: the row's first run takes the index of the row's first segment (a run owns a segment, so the rows' ranges cannot
// overlap; the record arrays are provisioned for N), the others follow densely.  A workgroup counts its own heads during the break
// walk, publishes them as one status word — the aggregate of its 2 048 segments, or at once the next index of the row its last
// segment belongs to when that row begins inside the tile — and takes the heads of its predecessors from theirs: a look-back that
// ends at the tile where the row began, i.e. after N / rows / 2 048 tiles (50 on the 4K scene, 10 on the 8K one), one wave-wide
// probe, with nothing to wait for but aggregates that are published a few microseconds into a workgroup's life.  (The single chain
// over all 6 720 tiles of round 4 — see above — needed every predecessor's PREFIX.)  The probe is made LATE, right in front of
// the first store that needs the index: a wave's predecessors were dispatched less than a microsecond before it, so a probe
// made when the wave itself has just published finds half of them unpublished and costs a second round trip (measured: the
// kernel at 112 us instead of 55); by the time the covers are summed and the style words requested they all have.  `status`
// starts from zero; row_base[row] = where the row begins; the first workgroup of the grid only keeps house (runs_housekeeping).
struct RunChain {
    uint32_t*       status;       // one word per 2 048-segment tile, lookback.h's u32 layout (BLOCKS: the tile's paintable run heads)
    uint32_t*       row_base;     // tiles_h + 1 words
    uint64_t        spec_live44;  // runs_housekeeping's arguments
    uint32_t        spec_flags;
    PendingMasks    pm;
    const uint32_t* range_records;
    uint32_t        n_range_records;
};
// BLOCKS (MODE 2, round 6): no counting pass and no look-back either.  Runs are numbered per 2 048-SEGMENT TILE of this kernel:
// the tile's first paintable head takes the index of the tile's first segment, the others follow densely (a run owns a segment:
// the tiles' ranges cannot overlap; `records` and `run_lt` are then SPARSE arrays provisioned for N).  A tile leaves its head count in
// `status[tile]` and the first head of a tile row its index in `row_base[row]`; k_carry_rows — one workgroup per row, which reads
// every run of its row anyway — walks the row's tiles, copies the runs into the DENSE stream-order numbering the painters expect
// (records[row_lo + e], row_lo = the runs of the rows above) and fills the first-run table.  What the counting pass cost — a second
// read of the sorted stream, 20 us on the 4K scene — is gone; the price is one 32-byte copy per run inside the carry kernel.
template <int MODE>
__global__ __launch_bounds__(RW_THREADS) void k_runs_wave(const uint64_t* __restrict__ sorted, DevCount nc, uint32_t tiles_w,
                                                          uint32_t tiles_h, TileRecord* __restrict__ records, uint32_t rec_cap,
                                                          uint64_t* __restrict__ run_keys, uint32_t* __restrict__ tile_first_run,
                                                          BlkEdge* __restrict__ blk_edge /* one per RW_CHUNK segments */,
                                                          uint32_t* __restrict__ row_count,
                                                          const uint32_t* __restrict__ run_counts, int counts_scanned,
                                                          const uint32_t* __restrict__ chunk_counts,
                                                          FrameInfo* __restrict__ info, RunStyle rs, RunChain ch) {
    constexpr bool CHAIN = MODE == 1, BLOCKS = MODE == 2, OWN_HOUSE = MODE != 0;   // (OWN_HOUSE: no k_runs_count in front: the first workgroup keeps house)
    __shared__ RunWaveLds s_w[RW_WAVES];
    __shared__ uint32_t s_rows[RN_ROWS];
    __shared__ uint32_t s_jb[RW_WAVES];
    __shared__ uint32_t s_ws[OWN_HOUSE ? RW_WAVES : 1][3];              // CHAIN, per wave: heads | heads from its last row start on | that row start + 1 (0: none); BLOCKS: heads
    __shared__ uint32_t s_hk[OWN_HOUSE ? 5 : 1][RC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = dev_count(nc);
    if (OWN_HOUSE && blockIdx.x == 0) {                                 // (the FIRST workgroup: its round trips run beside the others' work)
        runs_housekeeping(true, n, nc, info, ch.spec_live44, ch.spec_flags, ch.pm, ch.range_records, ch.n_range_records, s_hk);
        return;
    }
    const uint32_t tb = OWN_HOUSE ? blockIdx.x - 1u : blockIdx.x;       // this workgroup's 2 048-segment tile
    if ((uint64_t)tb * RN_TILE >= n) return;                           // whole workgroup: the grid was sized for the bound
    RunWaveLds& L = s_w[w];
    if (tid < RN_ROWS) s_rows[tid] = 0;

    // ---- the lane's 8 segments (64 contiguous bytes: four independent 16-byte loads) and the key in front of them.
    //      A lane past the end reads the stream's head instead; one that straddles the end over-reads by less than 64
    //      bytes, which the segment buffers are padded for.  Both are patched below.
    const uint32_t cbase = tb * RN_TILE + w * RW_CHUNK;        // first segment of the wave's chunk
    const uint32_t g0 = cbase + lane * RW_SEGS;
    const uint32_t chunk_n = cbase < n ? min((uint32_t)RW_CHUNK, n - cbase) : 0u;
    uint32_t lo[RW_SEGS], hi[RW_SEGS];
    {
        const uint4* p4 = reinterpret_cast<const uint4*>(sorted + (g0 < n ? g0 : 0u));   // 16-byte aligned (g0 multiple of 8)
        uint4 t[RW_SEGS / 2];
#pragma unroll
        for (int q = 0; q < RW_SEGS / 2; q++) t[q] = p4[q];
#pragma unroll
        for (int q = 0; q < RW_SEGS / 2; q++) { lo[2 * q] = t[q].x; hi[2 * q] = t[q].y; lo[2 * q + 1] = t[q].z; hi[2 * q + 1] = t[q].w; }
    }
    const uint64_t k_before = (lane == 0 && cbase > 0 && cbase < n) ? sorted[cbase - 1] : 0ull;
    const uint32_t row0_tyb = (uint32_t)(sorted[tb * RN_TILE] >> 53);
    // run index of this tile's first head = sum of the head counts of the tiles before it (the count array is L2-resident:
    // every workgroup adds it up itself, which saves a scan launch; big frames get it pre-scanned).  Issued AFTER the segment
    // loads and eight loads at a time: a one-load-per-round-trip loop here was most of a workgroup's life.
    uint32_t jnext = 0;                                                 // dense index of the chunk's next paintable head
    if (OWN_HOUSE) {}
    else if (counts_scanned) jnext = run_counts[tb];
    else {
        uint32_t acc = 0;
        for (uint32_t i0 = 0; i0 < tb; i0 += 8 * RW_THREADS) {
            uint32_t c[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const uint32_t i = i0 + k * RW_THREADS + tid; c[k] = i < tb ? run_counts[i] : 0u; }
#pragma unroll
            for (int k = 0; k < 8; k++) acc += c[k];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (lane == 0) s_jb[w] = acc;
    }
    if (!OWN_HOUSE) {                                                   // + the heads of the tile's earlier chunks
        uint32_t c = (lane < w) ? chunk_counts[(tb * RN_TILE >> 9) + lane] : 0u;           // w <= 3 loads
        c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64);
        jnext += (uint32_t)__builtin_amdgcn_readlane((int)c, 0);
    }
    if (chunk_n < RW_CHUNK && chunk_n) {
        // the stream ends inside this chunk (one wave per frame): what lies past the end becomes a copy of the last
        // segment with cover 0 — same key (no break), adds nothing — so that the walks need no bounds checks
        const uint64_t last = sorted[cbase + chunk_n - 1];
#pragma unroll
        for (int q = 0; q < RW_SEGS; q++)
            if (lane * RW_SEGS + q >= chunk_n) { lo[q] = (uint32_t)last & ~0x3Fu; hi[q] = (uint32_t)(last >> 32); }
    }
    uint32_t plo = wave_shr1(lo[RW_SEGS - 1]), phi = wave_shr1(hi[RW_SEGS - 1]);
    if (lane == 0) { plo = (uint32_t)k_before; phi = cbase ? (uint32_t)(k_before >> 32) : ~hi[0]; }   // the stream's first segment is a head
    // ---- walk 1: breaks (key changes) ---------------------------------------------------------------------------------------
    uint32_t bm = 0;                                                    // bit q of the lane's 8 segments
    {
        uint32_t qlo = plo, qhi = phi;
#pragma unroll
        for (int q = 0; q < RW_SEGS; q++) {
            if (((hi[q] ^ qhi) | ((lo[q] ^ qlo) >> SEG_KEY_SHIFT)) != 0u) bm |= 1u << q;
            qlo = lo[q]; qhi = hi[q];
        }
    }
    const uint32_t nb = (uint32_t)__popc(bm);
    const uint32_t nb_incl = wave_incl_scan_u32(nb);
    const uint32_t slot_lane = nb_incl - nb;                            // breaks in the lanes before this one
    const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)nb_incl, 63);   // breaks in the chunk = its last slot
    const uint32_t first_hi = (uint32_t)__builtin_amdgcn_readlane((int)hi[0], 0);
    const uint32_t before_hi = (uint32_t)__builtin_amdgcn_readlane((int)phi, 0);
    if (CHAIN) {
        // the wave's paintable heads and row starts (both are breaks), from the registers of walk 1
        uint32_t hm = 0, rm = 0;
        if (chunk_n) {
            uint32_t qhi = phi;
#pragma unroll
            for (int q = 0; q < RW_SEGS; q++) {
                if ((bm >> q) & 1u) {
                    const uint32_t tybq = hi[q] >> 21, txbq = (hi[q] >> 9) & 0xFFFu;
                    if (tybq - 1u < tiles_h && txbq <= tiles_w) hm |= 1u << q;      // seg_paintable
                    if ((qhi >> 21) != tybq) rm |= 1u << q;                          // (the stream's first segment: phi = ~hi[0])
                }
                qhi = hi[q];
            }
        }
        const uint32_t nh = (uint32_t)__popc(hm);
        const uint32_t tot = lb_wave_sum(nh);
        const uint64_t rsl = __ballot(rm != 0u);
        uint32_t aft = tot, rsp1 = 0;
        if (rsl) {                                                      // (wave-uniform)
            const int Ls = 63 - __builtin_clzll(rsl);                   // the lane of the wave's last row start, its segment there
            const uint32_t rmL = (uint32_t)__shfl((int)rm, Ls, 64);
            const int ql = 31 - __builtin_clz(rmL);
            aft = lb_wave_sum(lane > Ls ? nh : (lane == Ls ? (uint32_t)__popc(hm >> ql) : 0u));
            rsp1 = cbase + (uint32_t)(Ls * RW_SEGS + ql) + 1u;
        }
        if (lane == 0) { s_ws[w][0] = tot; s_ws[w][1] = aft; s_ws[w][2] = rsp1; }
    }
    if (BLOCKS) {
        // the wave's paintable heads (breaks whose tile fields lie on the canvas: seg_paintable), from the registers of walk 1
        uint32_t nh = 0;
        if (chunk_n) {
#pragma unroll
            for (int q = 0; q < RW_SEGS; q++) {
                const uint32_t tybq = hi[q] >> 21, txbq = (hi[q] >> 9) & 0xFFFu;
                if (((bm >> q) & 1u) && tybq - 1u < tiles_h && txbq <= tiles_w) nh++;
            }
        }
        const uint32_t tot = lb_wave_sum(nh);
        if (lane == 0) s_ws[w][0] = tot;
    }
    __syncthreads();
    uint32_t jrow = 0;                                                  // CHAIN: the next run index of the row the wave's next slot continues ...
    bool lb_pending = false, pfx_pending = false;                       // ... still without the heads of the tiles in front / the tile's PREFIX is this wave's to publish
    uint32_t pfx_add = 0;
    if (CHAIN) {
        uint32_t wt[RW_WAVES], wa[RW_WAVES], wr[RW_WAVES];
#pragma unroll
        for (int v = 0; v < RW_WAVES; v++) { wt[v] = s_ws[v][0]; wa[v] = s_ws[v][1]; wr[v] = s_ws[v][2]; }
        int ul = -1;                                                    // the tile's last wave with a row start
#pragma unroll
        for (int v = 0; v < RW_WAVES; v++) if (wr[v]) ul = v;
        if (w == 0 && lane == 0) {                                      // the tile's status word, before anybody waits for anything
            uint32_t val = 0, flag = LB_AGG;
#pragma unroll
            for (int v = 0; v < RW_WAVES; v++) {
                if (v == ul) { val = (wr[v] - 1u) + wa[v]; flag = LB_PREFIX; }
                else if (v > ul) val += wt[v];
            }
            lb_st32(&ch.status[tb], (flag << 30) | (val & 0x3FFFFFFFu));
        }
        int u = -1;                                                     // the last wave in front of this one with a row start
#pragma unroll
        for (int v = 0; v < RW_WAVES; v++) if (v < w && wr[v]) u = v;
        if (u >= 0) {
#pragma unroll
            for (int v = 0; v < RW_WAVES; v++) {
                if (v == u) jrow = (wr[v] - 1u) + wa[v];
                else if (v > u && v < w) jrow += wt[v];
            }
        } else {
            // the row continues from the tiles in front: their heads are looked up where the first index is needed (below)
            lb_pending = true;
#pragma unroll
            for (int v = 0; v < RW_WAVES; v++) if (v < w) jrow += wt[v];
            if (w == RW_WAVES - 1 && ul < 0) { pfx_pending = true; pfx_add = jrow + wt[RW_WAVES - 1]; }   // (this wave completes the tile's status)
        }
    } else if (BLOCKS) {
        // the tile's first head takes the index of the tile's first segment; this wave's first one follows the heads of the waves in front
        uint32_t heads = 0;
        jnext = tb * BLK_STRIDE;
#pragma unroll
        for (int v = 0; v < RW_WAVES; v++) { const uint32_t t = s_ws[v][0]; if (v < w) jnext += t; heads += t; }
        if (tid == 0) ch.status[tb] = heads;
    } else if (!counts_scanned) {
#pragma unroll
        for (int q = 0; q < RW_WAVES; q++) jnext += s_jb[q];
        if (tid == 0 && tb == (n + RN_TILE - 1) / RN_TILE - 1) info->n_runs = jnext + run_counts[tb];
    }
    // first tile row any head of this tile can have (rows are non-decreasing along the stream)
    const uint32_t row0 = row0_tyb ? row0_tyb - 1u : 0u;
    uint32_t prev_tile = 0;                                             // tile of the run in front of the sweep's first slot

    for (uint32_t s0 = 0; s0 <= R && chunk_n; s0 += RW_SLOTS) {
        const bool one_sweep = R < RW_SLOTS;                            // wave-uniform: every slot of the chunk fits this sweep
        // ---- clear the sweep's bins (the previous sweep's readers are done: same wave, LDS in order) --------------------
        {
            int4* z = reinterpret_cast<int4*>(L.bins);
            for (int i = lane; i < RW_SLOTS * RN_STRIDE / 4; i += 64) z[i] = make_int4(0, 0, 0, 0);
        }
        wave_lds_fence();
        // ---- walk 2: one ds_add per segment into bins[slot][local_y] ---------------------------------------------------------
        {
            uint32_t slot = slot_lane - s0;
#pragma unroll
            for (int q = 0; q < RW_SEGS; q++) {
                slot += (bm >> q) & 1u;
                if (one_sweep || slot < RW_SLOTS)                       // (slots before s0 wrap to huge values)
                    atomicAdd(&L.bins[slot * RN_STRIDE + ((lo[q] >> 12) & 15u)], ((int)(lo[q] << 26)) >> 26);
            }
            // where the sweep's slots begin: a lane parks its first break, then (few lanes) the others
            uint32_t m = bm, k = 1;
            while (__any(m != 0u)) {
                if (m) {
                    const uint32_t sl = slot_lane + k - s0;
                    if (sl <= RW_SLOTS) L.start[sl] = (uint32_t)(lane * RW_SEGS) + (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                }
                k++;
            }
            if (lane == 0 && R + 1u - s0 <= RW_SLOTS) L.start[R + 1u - s0] = chunk_n;    // the last run ends with the chunk
        }
        wave_lds_fence();
        // ---- one lane per slot: slot s0 + lane ---------------------------------------------------------------------------------
        const uint32_t sg = s0 + (uint32_t)lane;
        const bool mine = sg >= 1u && sg <= R;
        uint32_t st = 0, st_next = 0;
        if (mine) { st = L.start[lane]; st_next = L.start[lane + 1]; }
        const uint64_t key = mine ? sorted[cbase + st] : 0ull;          // the run's first segment (just read by this wave: a cache hit)
        const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
        const uint32_t tyb = khi >> 21, txb = (khi >> 9) & 0xFFFu, tile = khi >> 9;
        const bool val = mine && tyb - 1u < tiles_h && txb <= tiles_w;  // seg_paintable
        // tile of the segment in front of the run: the previous slot's run; for slot 1 the chunk's leading run (or, when the
        // chunk starts with a break, the segment in front of the chunk)
        uint32_t ptile = wave_shr1(tile);
        if (lane == 0) ptile = prev_tile;
        if (sg == 1u) ptile = (st ? first_hi : before_hi) >> 9;
        const bool new_tile = tile != ptile || (cbase + st) == 0u;
        prev_tile = (uint32_t)__builtin_amdgcn_readlane((int)tile, 63);
        const uint64_t bv = __ballot(val);
        int b[16];
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = L.bins[lane * RN_STRIDE + k];
        const uint4 pb = pack_bins(b);                                  // the slot's cover sum, 16 x i8
        if (!CHAIN) {                                                   // (dense numbering, or BLOCKS: dense from the tile's first segment index)
            const uint32_t j = jnext + __builtin_amdgcn_mbcnt_hi((uint32_t)(bv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bv, 0u));
            if (val) {
                const uint32_t open = (sg == R && chunk_n == RW_CHUNK) ? RUN_OPEN : 0u;
                const uint32_t layer = ((khi & 0x1FFu) << 12) | (klo >> 20);
                const uint2 sw = run_style_words(rs, layer, tile);          // (one gather per run, here on all 256 CUs)
                if (j < rec_cap) {                                          // asynchronous frames provision for a predicted run count
                    uint4* rp = reinterpret_cast<uint4*>(&records[j]);
                    rp[0] = pb;                                             // the run's own cover sum; k_carry_rows turns it into the carry-in
                    rp[1] = make_uint4(cbase + st, (st_next - st) | open, sw.x, sw.y);
                    if (!BLOCKS) run_keys[j] = ((uint64_t)tyb << 53) | ((uint64_t)layer << 32) | j;
                    rs.run_lt[j] = ((layer & 0xFFFFu) << 16) | (open ? RUN_LT_OPEN : 0u) | ((BLOCKS && new_tile) ? RUN_LT_NEWTILE : 0u) | txb;
                }
                if (BLOCKS) {
                    // (the first-run table is k_carry_rows' to fill, with dense indices) the first head of a tile row says where the row begins
                    if ((ptile >> 12) != tyb || (cbase + st) == 0u) ch.row_base[tyb - 1u] = j;
                } else
                if (txb >= 1u && new_tile) tile_first_run[(tyb - 1u) * tiles_w + (txb - 1u)] = j + 1u;   // 0 = the tile has no run
                const uint32_t rr = (tyb - 1u) - row0;
                if (rr < RN_ROWS) atomicAdd(&s_rows[rr], 1u); else atomicAdd(&row_count[tyb - 1u], 1u);
            }
        } else {
            // a slot that begins a tile row restarts the numbering at its own segment index
            const bool rs_slot = mine && (((ptile >> 12) != tyb) || (cbase + st) == 0u);
            const uint64_t rsb = __ballot(rs_slot);
            const uint64_t below = (1ull << lane) - 1ull;
            const uint64_t rs_le = rsb & ((below << 1) | 1ull);          // row starts at or before this slot
            const uint32_t pos = cbase + st;
            const int r = rs_le ? 63 - __builtin_clzll(rs_le) : 0;
            const uint32_t pos_r = (uint32_t)__shfl((int)pos, r, 64);
            const bool jabs = rs_le != 0ull;                             // the slot's row began in this sweep: its index needs nobody
            const uint32_t jpart = jabs ? pos_r + (uint32_t)__popcll(bv & below & ~((1ull << r) - 1ull)) : (uint32_t)__popcll(bv & below);
            if (rs_slot && tyb - 1u < tiles_h) ch.row_base[tyb - 1u] = pos;
            const uint32_t open = (sg == R && chunk_n == RW_CHUNK) ? RUN_OPEN : 0u;
            const uint32_t layer = ((khi & 0x1FFu) << 12) | (klo >> 20);
            uint2 sw = make_uint2(0u, 0u);
            if (val) sw = run_style_words(rs, layer, tile);              // (one gather per run, here on all 256 CUs)
            if (lb_pending && (__ballot(val && !jabs) != 0ull || pfx_pending)) {       // (wave-uniform)
                __builtin_amdgcn_sched_barrier(0);                       // (the probe goes out behind the gathers, not in front of the wave's loads)
                const uint32_t lb = lb_lookback_u32(ch.status, tb, &info->plan_bad);   // (a predecessor that never shows up voids the frame)
                __builtin_amdgcn_sched_barrier(0);
                jrow += lb; lb_pending = false;
                if (pfx_pending && lane == 0) lb_st32(&ch.status[tb], (LB_PREFIX << 30) | ((lb + pfx_add) & 0x3FFFFFFFu));
                pfx_pending = false;
            }
            if (val) {
                const uint32_t j = jabs ? jpart : jrow + jpart;
                if (j < rec_cap) {
                    uint4* rp = reinterpret_cast<uint4*>(&records[j]);
                    rp[0] = pb;
                    rp[1] = make_uint4(cbase + st, (st_next - st) | open, sw.x, sw.y);
                    rs.run_lt[j] = ((layer & 0xFFFFu) << 16) | (open ? RUN_LT_OPEN : 0u) | txb;
                }
                if (txb >= 1u && new_tile) tile_first_run[(tyb - 1u) * tiles_w + (txb - 1u)] = j + 1u;
                const uint32_t rr = (tyb - 1u) - row0;
                if (rr < RN_ROWS) atomicAdd(&s_rows[rr], 1u); else atomicAdd(&row_count[tyb - 1u], 1u);
            }
            if (rsb) {                                                   // (wave-uniform) what the next sweep continues
                const int rl = 63 - __builtin_clzll(rsb);
                jrow = (uint32_t)__shfl((int)pos, rl, 64) + (uint32_t)__popcll(bv >> rl);
                lb_pending = false;                                      // (absolute from here on)
            } else jrow += (uint32_t)__popcll(bv);
        }
        jnext += (uint32_t)__popcll(bv);
        if (sg == 0) {                                                  // slot 0: what this chunk adds to a run that began before it
            uint4* ep = reinterpret_cast<uint4*>(&blk_edge[cbase / RW_CHUNK]);
            ep[0] = pb;
            ep[1] = make_uint4(R ? L.start[1] : chunk_n, R ? 1u : 0u, 0u, 0u);
        }
        wave_lds_fence();
    }
    __syncthreads();
    if (tid < RN_ROWS && s_rows[tid]) atomicAdd(&row_count[row0 + tid], s_rows[tid]);
}

size_t runs_scratch_words(size_t n) { return 5 * ((n + RN_TILE - 1) / RN_TILE + 2) + 16; }   // [tile counts | chunk counts]
size_t runs_chain_words(size_t n) { return (n + RN_TILE - 1) / RN_TILE + 1; }                  // (chain: the tiles' status words)
size_t runs_blocks(size_t n) { return (n + RW_CHUNK - 1) / RW_CHUNK + 4; }      // BlkEdge entries: one per wave chunk

void launch_runs(hipStream_t s, const uint64_t* sorted, DevCount nc, uint32_t tiles_w, uint32_t tiles_h, TileRecord* records,
                 uint32_t rec_cap, uint64_t* run_keys, uint32_t* tile_first_run, BlkEdge* blk_edge, uint32_t* row_tab,
                 uint32_t* scratch, FrameInfo* info, bool verify_plan, uint64_t spec_live44, bool spec_layer_sorted,
                 PendingMasks pm, RunStyle rs, bool tables_are_zero, const uint32_t* range_records, uint32_t n_range_records, int what,
                 uint32_t* chain_row_base, bool chain_status_is_zero, bool blocks) {
    // per-frame tile tables: [row_count | row_span_lo | row_span_cnt | painter overflow counters (2) | first-run table] are
    // contiguous (api.cpp lays them out so) and start from zero — cleared by k_runs_count, unless an earlier kernel of the frame
    // already did (api.cpp folds that into the frame's first kernel); 0 in the first-run table = the tile has no run
    const uint32_t zero_words = tables_are_zero ? 0u : row_tab_zero_words(tiles_w, tiles_h);
    if (nc.bound == 0) {
        if (what & 1) {
            if (zero_words) (void)hipMemsetAsync(row_tab, 0, (size_t)zero_words * 4, s);
            (void)hipMemsetAsync(&info->n_runs, 0, 4, s);
        }
        return;
    }
    const uint32_t ntiles = (nc.bound + RN_TILE - 1) / RN_TILE;
    const uint32_t flags = (verify_plan ? 1u : 0u) | (spec_layer_sorted ? 2u : 0u);
    const uint32_t cgrid = std::min<uint32_t>((ntiles + 1) / 2, 4096u);
    uint32_t* chunk_counts = scratch + ntiles + 8;                      // 4 per tile (+ slack for the last wave's second half)
    const int scanned = ntiles > 16384 ? 1 : 0;
    if (chain_row_base && blocks) {
        // no counting pass and no look-back (k_runs_wave<2>): runs numbered per 2 048-segment tile, `records` / rs.run_lt are the SPARSE
        // arrays (n.bound entries), the head of `scratch` takes the tiles' head counts (every tile of the stream writes its own)
        if (zero_words) (void)hipMemsetAsync(row_tab, 0, (size_t)zero_words * 4, s);
        FORMA_LAUNCH(k_runs_wave<2>, dim3(ntiles + 1), dim3(RW_THREADS), 0, s, sorted, nc, tiles_w, tiles_h, records, rec_cap,
                           run_keys, tile_first_run, blk_edge, row_tab, (const uint32_t*)nullptr, 0, (const uint32_t*)nullptr, info, rs,
                           RunChain{scratch, chain_row_base, spec_live44, flags, pm, range_records, n_range_records});
        return;
    }
    if (chain_row_base) {
        // no counting pass (k_runs_wave<1>): the tables and the tiles' status words (the head of `scratch`) are cleared by an
        // earlier kernel of the frame or by two memsets here; one more workgroup (the first) keeps house
        if (zero_words) (void)hipMemsetAsync(row_tab, 0, (size_t)zero_words * 4, s);
        if (!chain_status_is_zero) (void)hipMemsetAsync(scratch, 0, (size_t)ntiles * 4, s);
        FORMA_LAUNCH(k_runs_wave<1>, dim3(ntiles + 1), dim3(RW_THREADS), 0, s, sorted, nc, tiles_w, tiles_h, records, rec_cap,
                           run_keys, tile_first_run, blk_edge, row_tab, (const uint32_t*)nullptr, 0, (const uint32_t*)nullptr, info, rs,
                           RunChain{scratch, chain_row_base, spec_live44, flags, pm, range_records, n_range_records});
        return;
    }
    if (what & 1) {
        FORMA_LAUNCH(k_runs_count, dim3(cgrid + 1), dim3(RC_THREADS), 0, s, sorted, nc, tiles_w, tiles_h, scratch, chunk_counts, row_tab,
                           zero_words, info, spec_live44, flags, pm, range_records, n_range_records);
        if (scanned) launch_scan_small_u32(s, scratch, nc, RN_TILE, &info->n_runs);   // exclusive, in place; total -> n_runs
    }
    if (what & 2)
        FORMA_LAUNCH(k_runs_wave<0>, dim3(ntiles), dim3(RW_THREADS), 0, s, sorted, nc, tiles_w, tiles_h, records, rec_cap,
                           run_keys, tile_first_run, blk_edge, row_tab, (const uint32_t*)scratch, scanned,
                           (const uint32_t*)chunk_counts, info, rs, RunChain{nullptr, nullptr, 0ull, 0u, PendingMasks{nullptr, 0u}, nullptr, 0u});
}
uint32_t runs_edge_segments() { return RW_CHUNK; }
uint32_t runs_count_tiles(size_t n, bool* scanned) { const uint32_t t = (uint32_t)((n + RN_TILE - 1) / RN_TILE); *scanned = t > 16384; return t; }

// ================================================================================================
// carry pre-pass: one 1024-lane workgroup per tile row.  The row's runs are brought into (layer, tile_x) order (LOCAL:
// by the workgroup itself in LDS; else run_keys after a stable radix sort on the (tile_y, layer) bits).  A segmented scan of
// the 16 x i8 cover sums over each (row, layer) group gives every run its carry-in:
//   carry-in(run) = wrapping sum of the covers of all runs of the group with smaller tile_x
//                   (painter/mod.rs:500-522 for the left-of-canvas bucket; layer_workbench/mod.rs:325-333)
// and, where the carry-out is not empty (Cover::is_empty, painter/mod.rs:187-198), a SPAN record for the
// tiles strictly between this run and the next one of the group ("carry-only" tiles: interior of shapes).
// Dropping an empty carry (mod.rs:335-339) equals carrying it: NonZero drops only all-zero covers and
// EvenOdd coverage only sees covers modulo 32.  Spans of a row are written in (layer, tile_x) order into
// the row's own slice [row_first_run, row_first_run + n) of the span arrays, so no counter is shared.
// ================================================================================================
#define CR_THREADS 1024
#define CR_WAVES   (CR_THREADS / 64)
#define CR_CAP     16384                // runs one workgroup's in-LDS sort holds, large variant (2 x 64 KiB of 32-bit keys: one per CU)
#define CR_CAP_S   4096                 // ... small variant (2 x 16 KiB of keys, 66 KB in all.  NOT two per CU: its 81 VGPRs x 16 waves
                                        // leave room for one workgroup per CU like the large one — it is the shorter sort that pays)
#ifndef CR_SMALL_OCC
#define CR_SMALL_OCC 4
#endif
#ifndef CR_HALF_OCC
#define CR_HALF_OCC 6
#endif
#define CR_THREADS_H 512                // the HALF variant: 512 lanes, one piece of 2048 runs — three per CU (LDS 33 KB, 24 of the CU's
#define CR_CAP_H   2048                 // waves), so that a whole 4K frame (135 rows x 3 slices) is resident at once on all 256 CUs
#define CR_MAX_SLICES 8                 // workgroups that share one tile row (each takes a range of layers)
// COVL (round 6): rows of <= CR_CAP_L runs in ONE slice — the 4K scene's heaviest has 5 512 — bring what the row walk gathered per
// run into LDS first: the runs' own cover sums (16 B each, ONE round of coalesced loads over the row's contiguous records once the
// in-LDS sort is done: 88 KB, in the space the sort's wave counters used) and their layers' style summaries (one gather per run,
// issued before the sort and parked in 16 bits per run next to the run digests in the sort's idle key buffer).  The walk then
// reads LDS only.  tools/cr_prof.py: "waiting for the gathers" was 28 k of a row's 80 k clocks — 8 scattered requests per lane
// through the one address unit of the row's CU.
#define CR_CAP_L   5632

// workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef CR_PROF
// -DCR_PROF (tools only): shader-clock stamps at the phase boundaries of k_carry_rows (thread 0 of every row's workgroup)
__device__ unsigned long long g_cr_prof[8];
#define CRP_STAMP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_cr_prof[i], _t - crp_t); crp_t = __builtin_readcyclecounter(); } while (0)
extern "C" int forma_hip_debug_cr_prof(unsigned long long* out8, int reset) {
    if (reset) { unsigned long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cr_prof), z, sizeof z); }
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cr_prof), 8 * 8);
}
#else
#define CRP_STAMP(i) do { } while (0)
#endif

// what the carry scan gathers per run, requested one piece ahead
struct CarryLoad {
    bool active;
    uint32_t group, jrun, layer, tile, sc, seg_start, lsf;
    uint4 oc;
};
template <bool LOCAL, bool COVL = false>
__device__ __forceinline__ CarryLoad carry_load(uint32_t c0, int tid, uint32_t cnt /* runs of the slice */, uint32_t row_lo /* first run of the row */,
                                                uint32_t kbase /* !LOCAL: first sorted key of the slice */, uint32_t n_runs, uint32_t ty,
                                                const uint32_t* lkeys, const uint64_t* __restrict__ sorted_keys,
                                                const TileRecord* __restrict__ records,
                                                const uint32_t* __restrict__ layer_sf, uint32_t n_orders,
                                                const uint16_t* s_txo /* LOCAL, one slice: the row's run_lt low halves in LDS */,
                                                const uint32_t* __restrict__ run_lt,
                                                const uint4* s_cov = nullptr /* COVL: the row's cover sums by run index, in LDS */,
                                                const uint16_t* s_sf16 = nullptr /* COVL: sfl | valid << 15 by run index, in LDS */) {
    CarryLoad L;
    const uint32_t k = (LOCAL ? row_lo : kbase) + c0 + tid;
    L.active = c0 + tid < cnt && k < n_runs;
    L.group = 0xFFFFFFFEu; L.jrun = 0; L.layer = 0; L.tile = 0; L.sc = 0; L.seg_start = 0; L.lsf = 0;
    L.oc = make_uint4(0, 0, 0, 0);
    if (L.active) {
        // Every scattered access is a cycle of this CU's address unit (a row's workgroup is alone on its CU): LOCAL takes the
        // tile column and the "open" flag from the run digests (k_runs_wave's run_lt: in LDS when the row has one slice) and
        // gathers ONLY the run's cover sum; the record's second half is read for the rare run that crosses its chunk.
        const uint4* rp;
        if (LOCAL) {
            const uint32_t pk = lkeys[c0 + tid];
            const uint32_t e = pk & 0xFFFFu;
            L.layer = pk >> 16; L.jrun = row_lo + e; L.group = ((ty + 1u) << 21) | L.layer;
            rp = reinterpret_cast<const uint4*>(&records[L.jrun]);
            const uint32_t txo = s_txo ? (uint32_t)s_txo[e] : (run_lt[L.jrun] & 0xFFFFu);
            L.tile = ((ty + 1u) << 12) | (txo & 0xFFFu);
            if (txo & RUN_LT_OPEN) { const uint4 tail = rp[1]; L.seg_start = tail.x; L.sc = tail.y; }
        } else {
            const uint64_t key = sorted_keys[k];
            L.group = (uint32_t)(key >> 32); L.jrun = (uint32_t)key; L.layer = L.group & 0x1FFFFFu;
            // the record in two 16-byte gathers (one cache line): own cover sum | seg_start, seg_count, layer, tile
            rp = reinterpret_cast<const uint4*>(&records[L.jrun]);
            const uint4 tail = rp[1];
            L.seg_start = tail.x; L.sc = tail.y; L.tile = tail.w & 0x7FFFFFFFu;
        }
        if (COVL) {
            const uint32_t e = L.jrun - row_lo;
            L.oc = s_cov[e];
            const uint32_t p = s_sf16[e];
            L.lsf = (p & 0x7FFFu) | ((p & 0x8000u) ? LSF_VALID : 0u);
        } else {
            L.oc = rp[0];
            if (L.layer < n_orders) L.lsf = layer_sf[L.layer];
        }
    }
    return L;
}

// LOCAL = true: `sorted_keys` are the run keys as k_runs wrote them (stream order).  The runs of a tile row are contiguous
// there ([row_lo, row_lo + cnt), tile_x-major), so the workgroup of the row orders them by (layer, tile_x) itself: a stable
// LSB radix sort of `layer << 16 | index` in LDS, one or two 8-bit passes over the layer bits, wave-ranked like a sort
// tile.  That replaces a global histogram + three chained radix passes over all J run keys (launch- and latency-bound:
// ~60 us per frame at J = 1.3 M) by a few microseconds inside a kernel that is launched anyway.  Needs n_orders <= 65536
// and a slice of <= CAP runs (checked on the device; the host falls back to the global sort, LOCAL = false).
//
// SLICES.  A tile row is shared by n_slices workgroups (blockIdx.x = row * n_slices + slice), each taking a contiguous RANGE
// OF LAYERS: carries never cross layers, so the slices are independent.  LOCAL: every workgroup of the row histograms the
// row's run keys over 256 layer bins (layer >> bin_shift), cuts the bins into n_slices ranges of about equal run counts and
// keeps (stable compaction) only the runs of its range — then sorts, gathers and scans a 1 / n_slices share.  !LOCAL: the
// globally sorted keys of the row are cut at layer boundaries.  Slice h writes its spans into its own sub-range of the row's
// span slots (row_lo + runs in the bins below it), and the painters walk a row's n_slices span lists one after the other —
// ascending layers, as before.  One workgroup per row was 135 workgroups on 256 CUs for the 4K frame, each ~50 us of pure
// latency (scattered record gathers through one CU's address pipe); on a multi-GPU band of 17 rows it was the frame's floor.
template <bool LOCAL, int CAP, int RPT, int TH, bool COVL = false>
__global__ __launch_bounds__(TH, TH == 512 ? (COVL ? 4 : CR_HALF_OCC) : (CAP == CR_CAP_S ? CR_SMALL_OCC : 1)) void k_carry_rows(   // (512 lanes: three workgroups per CU = six waves per SIMD; with COVL's 32 KB of covers two = four)
    const uint64_t* __restrict__ sorted_keys,
                                                           TileRecord* __restrict__ records,
                                                           const BlkEdge* __restrict__ blk_edge, DevCount nc_segments,
                                                           DevCount nc_runs,
                                                           const uint32_t* __restrict__ layer_sf, uint32_t n_orders,
                                                           uint32_t tiles_w, uint32_t tiles_h,
                                                           const uint32_t* __restrict__ row_count,
                                                           uint32_t* __restrict__ row_span_lo,
                                                           uint32_t* __restrict__ row_span_cnt,
                                                           uint64_t* __restrict__ span_key, uint4* __restrict__ span_cov,
                                                           const uint8_t* __restrict__ unchanged,
                                                           FrameInfo* __restrict__ info, uint32_t edge_segs,
                                                           uint32_t vis_last /* visible pixel rows of the last tile row, 16 = all */,
                                                           uint32_t n_slices, uint32_t bin_shift,
                                                           uint32_t row0 /* first tile row that is painted: blockIdx.x = 0 */,
                                                           SpanGroups groups, const uint32_t* __restrict__ run_lt,
                                                           uint32_t cull /* PaintParams::cull: the group lists leave out what an occluder of the whole group hides */,
                                                           uint32_t left_start /* see below; 0xFFFFFFFF: off */,
                                                           const uint32_t* __restrict__ row_base /* nullable: where each row's runs begin
                                                           (launch_runs' chain numbering); else the rows' runs are dense in row order */,
                                                           BlkRuns bk /* rec_sp != nullptr: the run kernel numbered per 2 048-segment tile (below) */) {
    constexpr int CR_RPT = RPT;                        // consecutive runs of the (layer, tile_x) order per lane in the row walk
    constexpr int CR_PIECE = TH * RPT;         // runs per piece
    constexpr bool NB_IN_IDLE = LOCAL && !COVL && 2 * (CR_PIECE + 1) <= CAP;   // group / tile_x of a piece's runs live in the sort's idle buffer
    static_assert(!COVL || LOCAL, "COVL is a LOCAL variant");
    __shared__ uint32_t s_red[(TH / 64)];
    __shared__ uint64_t s_wlo[(TH / 64)], s_whi[(TH / 64)];
    __shared__ uint32_t s_wflag[(TH / 64)], s_wspan[(TH / 64)];
    __shared__ uint32_t s_group[256];                  // scratch of the in-LDS sort; before it, the layer-bin histogram
    __shared__ uint32_t s_nb[NB_IN_IDLE ? 1 : (COVL ? (CR_PIECE + 1) + (CR_PIECE + 4) / 2 : 2 * (CR_PIECE + 1))];   // a_group (words) + a_txb (halves)
    __shared__ uint64_t s_clo, s_chi;                  // carry across chunks: inclusive acc of the last element
    __shared__ uint32_t s_cgroup, s_spans;
    __shared__ uint32_t s_ka[LOCAL ? CAP : 1], s_kb[LOCAL ? CAP : 1];
    constexpr int KPL = (CAP + TH - 1) / TH;           // keys per lane of the in-LDS sort (CAP need not be a multiple of TH)
    __shared__ uint32_t s_wh_own[LOCAL && !COVL ? (TH / 64) * 256 : 1];
    __shared__ uint4 s_cov[COVL ? CAP : 1];            // COVL: the row's cover sums by run index — loaded AFTER the sort, whose wave counters live here until then
    static_assert(!COVL || (size_t)CAP * 16 >= (size_t)(TH / 64) * 256 * 4, "the sort's wave counters fit the cover array");
    uint32_t* s_wh = COVL ? reinterpret_cast<uint32_t*>(s_cov) : s_wh_own;
    uint32_t lsf_reg[COVL ? KPL : 1];                   // COVL: the style summaries of this lane's runs (stream order), in flight across the sort
    __shared__ uint32_t s_cut[4];                      // this slice: first bin, end bin | first key, end key (!LOCAL)
    __shared__ uint32_t s_gcnt[256], s_gpre[256];      // span group lists: entries per group, their exclusive prefix
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t ty = row0 + blockIdx.x / n_slices, slice = blockIdx.x % n_slices;
    // A canvas whose height is not a multiple of 16: lines entirely below it are culled (segment.rs:41-52), so a layer that
    // crosses the bottom edge keeps a non-zero cover on the INVISIBLE pixel rows of the last tile row.  The reference carries
    // it (Cover::is_empty looks at all 16 rows) and paints the layer with zero visible coverage in every tile to the right;
    // here such a carry produces no span: its pixels are never written (painter/mod.rs:537-548 writes the canvas rows only),
    // and a busy scene would otherwise put thousands of invisible layers into every tile of that row.  Only the emptiness
    // test of carry-only spans is masked; covers, runs and the `full` flag are untouched.  The host asks for this
    // (vis_last < 16) only on frames WITHOUT a buffer-layer cache: with one, a tile's layer count and the passes' verdicts
    // are remembered across frames (passes/tile_unchanged.rs, CachedTile), the invisible layers are part of both, and they
    // are carried exactly like the reference's.
    uint64_t vis_lo = ~0ull, vis_hi = ~0ull;
    if (ty + 1 == tiles_h && vis_last < 16u) {
        vis_lo = vis_last >= 8u ? ~0ull : ((1ull << (8u * vis_last)) - 1ull);
        vis_hi = vis_last <= 8u ? 0ull : ((1ull << (8u * (vis_last - 8u))) - 1ull);
    }
#ifdef CR_PROF
    unsigned long long crp_t = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicAdd(&g_cr_prof[7], 1ull);
#endif
    // the guard word, the two device-side counts and the row counts are independent loads: all of them are issued before
    // the first is tested (one global round trip instead of four in front of a kernel whose 135 workgroups are all latency)
    const uint32_t plan_bad = info->plan_bad;                           // mis-sorted stream (async frame): the host re-runs
    const uint32_t runs_dev = nc_runs.ptr ? *nc_runs.ptr : nc_runs.bound;
    const uint32_t n_blk = (dev_count(nc_segments) + edge_segs - 1) / edge_segs;   // BlkEdge entries (one per edge_segs segments)
    // first run of this row = sum of the run counts of the rows above
    uint32_t part = 0;
    const bool blocks = COVL && bk.rec_sp != nullptr;                   // (uniform; a COVL variant, one slice per row: the host's condition)
    if (row_base && !blocks) { if (tid == 0) part = row_count[ty] ? row_base[ty] : 0u; }   // (a row without runs has no entry)
    else for (uint32_t r = tid; r < ty; r += TH) part += row_count[r];
    const uint32_t cnt = row_count[ty];
    // BLOCKS: the row's first run in the sparse numbering (an entry nobody wrote — a row without runs — is not used), and the head
    // counts of the 256 tiles from there on: requested HERE, behind the first round of loads, not behind the barriers below
    const uint32_t row_sp = blocks ? bk.row_sp[ty] : 0u;
    uint32_t blk_avail0 = 0;
    const uint32_t blk_round = bk.round_tiles;                          // (256; tests: fewer, so that small rows take several rounds)
    if (COVL && blocks && (uint32_t)tid < blk_round) {
        const uint32_t nt2k = (dev_count(nc_segments) + RN_TILE - 1u) / RN_TILE, b = row_sp / BLK_STRIDE + (uint32_t)tid;
        if (b < nt2k) { const uint32_t h = bk.heads[b]; const uint32_t skip = tid == 0 ? row_sp % BLK_STRIDE : 0u; blk_avail0 = h > skip ? h - skip : 0u; }
    }
    if (plan_bad) return;
    if (runs_dev > nc_runs.bound) {                                     // more runs than provisioned: the frame is void, and
        if (threadIdx.x == 0) info->plan_bad = 1u;                      // the painters (next launches) must not touch anything
        return;
    }
    const uint32_t n_runs = runs_dev < nc_runs.bound ? runs_dev : nc_runs.bound;    // = dev_count(nc_runs)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if (lane == 0) s_red[w] = part;
    if (tid == 0) { s_clo = 0; s_chi = 0; s_cgroup = 0xFFFFFFFFu; s_spans = 0; }
    __syncthreads();
    uint32_t row_lo = 0;
#pragma unroll
    for (int i = 0; i < (TH / 64); i++) row_lo += s_red[i];
    const uint32_t* lkeys = s_ka;                        // LOCAL: the slice's runs, ordered by (layer, tile_x)
    if (tid == 0 && cnt && slice == 0) atomicMax(&info->max_row_runs, cnt);
    // ---- BLOCKS: the run kernel numbered the runs per 2 048-segment tile b — [2048 b, 2048 b + heads[b]) — without knowing how many
    //      heads lie in front of a tile.  This row's runs are cnt consecutive ones of that sparse order from row_sp on; run e of the
    //      row (stream order) has the dense index row_lo + e every later step and the painters use.  A lane owns the runs
    //      e = r TH + tid like the loads below always did; where run e lies in the sparse arrays — blk_j[r] — comes from the head
    //      counts of the row's tiles: a lane per tile scans them (256 tiles per round: a 4K row spans ~50), every lane then finds its
    //      runs' tiles by binary search.  The digests (in front of the sort) and the cover sums (behind it) are loaded from there in
    //      one round of independent loads each, as before; the records' second halves go to their dense places on the way, with the
    //      segment count of a run that crosses its chunk made whole, and the first run of a tile enters the first-run table.
    //      (First form: whole records copied in front of everything else, a wave per tile — each of its steps a round trip of its
    //      own: k_carry_rows 44.6 -> 58.5 us on the 4K scene.  Second: the copies by a helper workgroup per row, the loads still a
    //      wave per tile: 56 us, and the 8K scene's 512 helpers made a second round of workgroups: 28.5 -> 46.  This form: 50-51 us
    //      on the 4K scene — the head counts are one more dependent round trip in front of everything, 13 k clocks per row with the
    //      scan and the search — for the 20.6 us of k_runs_count and 4 of k_runs_wave.  On the 8K triangle scene, 512 rows of light
    //      work on two workgroups per CU, the same round trip costs 30 k clocks at the start of a launch whose every workgroup asks at
    //      once: 28 -> 46 us, all that the counting pass cost — the host keeps the counting pass for frames of more LIGHT rows (this 512-lane variant) than CUs.)
    uint32_t blk_j[COVL ? KPL : 1];
    if (COVL && blocks) {
        if (row_lo + cnt > n_runs || cnt > (uint32_t)CAP) { if (tid == 0) info->plan_bad = 1u; return; }   // (more runs than provisioned / than this variant holds: the host re-runs)
        if (tid == 0) bk.row_base_out[ty] = row_lo;                     // what the painters bound a row's probes by (PaintParams::row_base)
        const uint32_t n_tiles2k = (dev_count(nc_segments) + RN_TILE - 1u) / RN_TILE;
        const uint32_t b0 = row_sp / BLK_STRIDE, o0 = row_sp % BLK_STRIDE;
#pragma unroll
        for (int r = 0; r < (COVL ? KPL : 1); r++) blk_j[r] = 0;
        uint32_t done = 0;
        __syncthreads();                                                // (s_red: every lane has its row_lo)
        for (uint32_t bb = b0; done < cnt; bb += blk_round) {
            if (bb >= n_tiles2k) { if (tid == 0) info->plan_bad = 1u; return; }     // (uniform: the counts do not add up)
            uint32_t avail = blk_avail0;
            if ((uint32_t)tid < blk_round && bb != b0) {
                const uint32_t b = bb + (uint32_t)tid;
                avail = b < n_tiles2k ? bk.heads[b] : 0u;
            }
            uint32_t inc = wave_incl_scan_u32(avail);
            if (tid < 256 && lane == 63) s_red[w] = inc;
            __syncthreads();
            if (tid < 256) {
                for (int i = 0; i < w; i++) inc += s_red[i];
                s_group[tid] = done + (inc - avail);                                  // the tile's first run: its place in the row ...
                s_gcnt[tid] = (bb + (uint32_t)tid) * BLK_STRIDE + ((bb + (uint32_t)tid) == b0 ? o0 : 0u);   // ... and in the sparse arrays
            }
            __syncthreads();
            const uint32_t batch = min(cnt - done, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
#pragma unroll
            for (int r = 0; r < (COVL ? KPL : 1); r++) {
                const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
                if (e >= done && e < done + batch) {                    // last tile of the round that begins at or before e (empty tiles share their successor's place)
                    uint32_t lo = 0, hi = 256;
#pragma unroll
                    for (int st = 0; st < 8; st++) { const uint32_t mid = (lo + hi) >> 1; if (s_group[mid] <= e) lo = mid; else hi = mid; }
                    blk_j[r] = s_gcnt[lo] + (e - s_group[lo]);
                }
            }
            done += batch;
            __syncthreads();                                            // (the tables are rewritten by the next round)
        }
    }
    uint32_t m = cnt;                                    // runs of this slice
    uint32_t off = 0;                                    // runs of the row in the slices before it (= its first span slot)
    uint32_t kbase = row_lo;                             // !LOCAL: first sorted key of the slice
    if (LOCAL) {
        if (cnt > 65535u || row_lo + cnt > n_runs) {      // (16 index bits per packed key) / inconsistent counts: the host re-runs the frame
            if (tid == 0) info->plan_bad = 1u;
            return;
        }
        CRP_STAMP(0);                                                   // prologue: counts, row prefix
        if (n_slices > 1u) {
            // ---- which layers are mine: 256-bin histogram of the row's run keys, cut into n_slices ranges of ~equal counts.
            //      Every workgroup of the row computes the same cuts from the same keys. ------------------------------------
            if (tid < 256) s_group[tid] = 0;
            __syncthreads();
            for (uint32_t e = tid; e < cnt; e += TH) {
                const uint32_t bin = min(255u, ((run_lt[row_lo + e] >> 16) >> bin_shift));
                atomicAdd(&s_group[bin], 1u);
            }
            __syncthreads();
            uint32_t binc = 0;                            // inclusive prefix of the bins
            if (tid < 256) {
                binc = s_group[tid];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(binc, d, 64); if (lane >= d) binc += t; }
                if (lane == 63) s_red[w] = binc;
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t base = 0;
                for (int i = 0; i < w; i++) base += s_red[i];
                s_group[tid] = base + binc;
            }
            __syncthreads();
            if (tid == 0) {
                // slice h owns the bins [cut(h), cut(h + 1)): cut(h) = bins that lie entirely inside the first h / n_slices of the runs
                uint32_t cutv[2];
                for (int q = 0; q < 2; q++) {
                    const uint32_t hh = slice + (uint32_t)q;
                    uint32_t b = 0;
                    if (hh >= n_slices) b = 256u;
                    else if (hh > 0) {
                        const uint32_t target = (uint32_t)(((uint64_t)hh * cnt) / n_slices);
                        uint32_t lo = 0, hi = 256;
                        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_group[mid] <= target) lo = mid + 1; else hi = mid; }
                        b = lo;
                    }
                    cutv[q] = b;
                }
                s_cut[0] = cutv[0]; s_cut[1] = cutv[1];
                s_cut[2] = cutv[0] ? s_group[cutv[0] - 1] : 0u;                           // runs below my bins
                s_cut[3] = cutv[1] ? s_group[cutv[1] - 1] : 0u;                           // runs below the next slice's bins
            }
            __syncthreads();
            const uint32_t blo = s_cut[0], bhi = s_cut[1];
            off = s_cut[2]; m = s_cut[3] - s_cut[2];
            if (m > (uint32_t)CAP) {                      // does not fit this variant's LDS: the host re-runs with the large variant / global sort
                if (tid == 0) info->plan_bad = 1u;
                return;
            }
            // ---- stable compaction of my runs, in stream (tile_x-major) order, 1024 keys per round ----------------------------
            uint32_t filled = 0;
            for (uint32_t e0 = 0; e0 < cnt; e0 += TH) {
                const uint32_t e = e0 + (uint32_t)tid;
                uint32_t l16 = 0; bool keep = false;
                if (e < cnt) {
                    l16 = run_lt[row_lo + e] >> 16;
                    const uint32_t bin = min(255u, l16 >> bin_shift);
                    keep = bin >= blo && bin < bhi;
                }
                const uint64_t bal = __ballot(keep);
                if (lane == 0) s_red[w] = (uint32_t)__popcll(bal);
                __syncthreads();
                uint32_t base = filled, tot = 0;
#pragma unroll
                for (int i = 0; i < (TH / 64); i++) { const uint32_t t = s_red[i]; if (i < w) base += t; tot += t; }
                if (keep) s_ka[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (l16 << 16) | e;
                filled += tot;
                __syncthreads();
            }
        } else {
            if (cnt > (uint32_t)CAP) { if (tid == 0) info->plan_bad = 1u; return; }
            if (COVL) {
                // the run digests (coalesced), then the layers' style summaries: a gather behind the digest that lands while the sort runs
                uint32_t rl[KPL];
#pragma unroll
                for (int r = 0; r < KPL; r++) {
                    const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
                    rl[r] = e < cnt ? (blocks ? bk.run_lt_sp[blk_j[r]] : run_lt[row_lo + e]) : 0u;
                }
#pragma unroll
                for (int r = 0; r < KPL; r++) {
                    const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
                    const uint32_t layer = rl[r] >> 16;
                    lsf_reg[r] = (e < cnt && layer < n_orders) ? layer_sf[layer] : 0u;
                    if (e < cnt) s_ka[e] = (rl[r] & 0xFFFF0000u) | e;
                }
            } else {
                for (uint32_t e = tid; e < cnt; e += TH) s_ka[e] = (run_lt[row_lo + e] & 0xFFFF0000u) | e;
            }
            __syncthreads();
        }
        if (COVL && n_slices > 1u) { if (tid == 0) info->plan_bad = 1u; return; }        // (the host launches COVL for one slice per row only)
        if (tid == 0 && m) atomicMax(&info->max_slice_runs, m);
        CRP_STAMP(1);                                                   // the slice's run keys into LDS
        uint32_t* src = s_ka;
        uint32_t* dst = s_kb;
        const uint32_t R = (m + TH - 1) / TH, CW = R * 64;         // key rows per wave, keys per wave
        const int npass = n_orders > 256u ? 2 : 1;
        for (int pass = 0; pass < npass; pass++) {
            const int sh = 16 + 8 * pass;
            for (int i = tid; i < (TH / 64) * 256; i += TH) s_wh[i] = 0;
            __syncthreads();
            CRP_STAMP(5);                                               // sort: counters cleared
            uint32_t kreg[KPL], rreg[KPL];
#pragma unroll
            for (int r = 0; r < KPL; r++) {
                if ((uint32_t)r < R) {
                    const uint32_t e = w * CW + r * 64 + lane;
                    const uint32_t key = e < m ? src[e] : 0xFFFFFFFFu;             // padding: last in stream order, digit 255
                    const uint32_t dg = (key >> sh) & 0xFFu;
                    uint32_t mlo, mhi;
                    match_any<8>(dg, mlo, mhi);
                    const uint32_t below = lanes_below(mlo, mhi);
                    const uint32_t c = (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
                    if (below == 0) atomicAdd(&s_wh[w * 256 + dg], c);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    const uint32_t after = s_wh[w * 256 + dg];
                    kreg[r] = key; rreg[r] = after - c + below;
                }
            }
            __syncthreads();
            CRP_STAMP(6);                                               // sort: keys ranked inside their wave
            if (tid < 256) {                              // digit tid: start of every wave's share of it
                uint32_t tot = 0;
#pragma unroll
                for (int i = 0; i < (TH / 64); i++) tot += s_wh[i * 256 + tid];
                uint32_t inc = tot;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
                if (lane == 63) s_red[w] = inc;
                s_group[tid] = inc - tot;                 // exclusive within the wave (s_group is free until the main loop)
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t acc = s_group[tid];
                for (int i = 0; i < w; i++) acc += s_red[i];
#pragma unroll
                for (int i = 0; i < (TH / 64); i++) { const uint32_t c = s_wh[i * 256 + tid]; s_wh[i * 256 + tid] = acc; acc += c; }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < KPL; r++) {
                if ((uint32_t)r < R) {
                    const uint32_t e = w * CW + r * 64 + lane;
                    if (e < m) dst[s_wh[w * 256 + ((kreg[r] >> sh) & 0xFFu)] + rreg[r]] = kreg[r];
                }
            }
            __syncthreads();
            uint32_t* t = src; src = dst; dst = t;
        }
        lkeys = src;
        CRP_STAMP(2);                                                   // in-LDS sort by layer
    } else if (n_slices > 1u) {
        // ---- !LOCAL: the row's keys are sorted by layer already; cut [row_lo, row_lo + cnt) at layer boundaries near h / n_slices
        if (row_lo + cnt > n_runs) { if (tid == 0) info->plan_bad = 1u; return; }
        for (int q = 0; q < 2; q++) {
            const uint32_t hh = slice + (uint32_t)q;
            uint32_t pos = hh >= n_slices ? cnt : (uint32_t)(((uint64_t)hh * cnt) / n_slices);
            if (hh > 0 && hh < n_slices) {                // forward to the next position where the layer changes
                if (tid == 0) s_cut[q] = cnt;
                __syncthreads();
                for (uint32_t p0 = pos; p0 < cnt; p0 += TH) {
                    const uint32_t pp = p0 + (uint32_t)tid;
                    bool brk = false;
                    if (pp < cnt && pp > 0) brk = (uint32_t)(sorted_keys[row_lo + pp] >> 32) != (uint32_t)(sorted_keys[row_lo + pp - 1] >> 32);
                    if (pp == 0) brk = true;
                    if (brk) atomicMin(&s_cut[q], pp);
                    __syncthreads();
                    if (s_cut[q] < cnt) break;            // (uniform)
                }
                pos = s_cut[q];
                __syncthreads();
            }
            if (q == 0) off = pos; else m = pos - off;
        }
        kbase = row_lo + off;
    }
    // The row is walked in pieces of CR_PIECE = 1024 x CR_RPT runs, a lane owning CR_RPT CONSECUTIVE runs of the (layer,
    // tile_x) order: all of a lane's gathers (record, cover sums, the layer's style summary) are in flight at once, the
    // segmented scan runs serially over the lane's runs and once across lanes per piece, and the barriers order LDS only.
    // (One run per lane — 1023 runs per piece, five pieces for an average 4K row — spent 43 % of the kernel in the per-piece
    // cross-lane machinery and 27 % waiting for one round of gathers per piece: tools/cr_prof.py.)
    // Group / tile_x of every run of the piece (+ the run after it) sit in LDS for the neighbour tests; LOCAL: in the sort's
    // idle buffer.
    uint32_t* idle = lkeys == s_ka ? s_kb : s_ka;
    uint32_t* a_group = NB_IN_IDLE ? idle : s_nb;
    uint16_t* a_txb = reinterpret_cast<uint16_t*>(a_group + (CR_PIECE + 1));
    // LOCAL, one slice per row: the low halves of the row's run digests (tile column + 1, "open" flag) by run index, next to them
    uint16_t* s_txo = nullptr;
    // COVL: the sort is done — its wave counters' space takes the row's cover sums (one round of coalesced loads: the records of a
    // row are contiguous), the idle key buffer's upper half the style summaries (its lower half: the run digests, s_txo below)
    uint16_t* s_sf16 = COVL ? reinterpret_cast<uint16_t*>(idle) + CAP : nullptr;
    if (COVL && !blocks) {
        uint4 cv[COVL ? KPL : 1];
#pragma unroll
        for (int r = 0; r < (COVL ? KPL : 1); r++) {
            const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
            cv[r] = e < cnt ? *reinterpret_cast<const uint4*>(&records[row_lo + e]) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int r = 0; r < (COVL ? KPL : 1); r++) {
            const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
            if (e < cnt) { s_cov[e] = cv[r]; s_sf16[e] = (uint16_t)((lsf_reg[r] & 0x7FFFu) | ((lsf_reg[r] & LSF_VALID) ? 0x8000u : 0u)); }
        }
    }
    if (COVL && blocks) {
        // the same from the sparse arrays (blk_j): cover sums and the digests' low halves in one round of loads, then the records'
        // second halves — on their way to the dense records.  A run that crosses its chunk is made whole HERE from the following
        // chunks' edges (cover sum and segment count: the walk below never sees an open run).
        uint16_t* txo_b = reinterpret_cast<uint16_t*>(idle);
        uint4 cv[COVL ? KPL : 1], tl[COVL ? KPL : 1];                   // (both halves of the records and the digests: ONE round of loads)
        uint32_t lt[COVL ? KPL : 1];
#pragma unroll
        for (int r = 0; r < (COVL ? KPL : 1); r++) {
            const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
            cv[r] = e < cnt ? reinterpret_cast<const uint4*>(&bk.rec_sp[blk_j[r]])[0] : make_uint4(0u, 0u, 0u, 0u);
            tl[r] = e < cnt ? reinterpret_cast<const uint4*>(&bk.rec_sp[blk_j[r]])[1] : make_uint4(0u, 0u, 0u, 0u);
            lt[r] = e < cnt ? bk.run_lt_sp[blk_j[r]] : 0u;
        }
#pragma unroll
        for (int r = 0; r < (COVL ? KPL : 1); r++) {
            const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
            if (e < cnt) {
                s_cov[e] = cv[r];
                s_sf16[e] = (uint16_t)((lsf_reg[r] & 0x7FFFu) | ((lsf_reg[r] & LSF_VALID) ? 0x8000u : 0u));
                txo_b[e] = (uint16_t)(lt[r] & 0xFFFu);                  // (neither "open" nor "first of its tile": both are dealt with here)
            }
        }
        // the runs that cross their chunk, all of a lane's at once: one round of edge loads per step of the longest walk (nearly
        // always one or two), not one walk after the other
        uint32_t eb[COVL ? KPL : 1], openm = 0;
#pragma unroll
        for (int r = 0; r < (COVL ? KPL : 1); r++) {
            const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
            eb[r] = tl[r].x / edge_segs + 1u;
            if (e < cnt && (tl[r].y & RUN_OPEN)) { tl[r].y &= ~RUN_OPEN; if (eb[r] < n_blk) openm |= 1u << r; }
        }
        while (__any(openm != 0u)) {
            uint4 ec[COVL ? KPL : 1]; uint2 ex[COVL ? KPL : 1];
#pragma unroll
            for (int r = 0; r < (COVL ? KPL : 1); r++) {
                if ((openm >> r) & 1u) {
                    const uint4* ep = reinterpret_cast<const uint4*>(&blk_edge[eb[r]]);
                    ec[r] = ep[0]; ex[r] = *reinterpret_cast<const uint2*>(ep + 1);
                }
            }
#pragma unroll
            for (int r = 0; r < (COVL ? KPL : 1); r++) {
                if ((openm >> r) & 1u) {
                    const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
                    const uint4 c = s_cov[e];
                    const uint64_t clo = swar_add8((uint64_t)c.x | ((uint64_t)c.y << 32), (uint64_t)ec[r].x | ((uint64_t)ec[r].y << 32));
                    const uint64_t chi = swar_add8((uint64_t)c.z | ((uint64_t)c.w << 32), (uint64_t)ec[r].z | ((uint64_t)ec[r].w << 32));
                    s_cov[e] = make_uint4((uint32_t)clo, (uint32_t)(clo >> 32), (uint32_t)chi, (uint32_t)(chi >> 32));
                    tl[r].y += ex[r].x;
                    eb[r]++;
                    if (ex[r].y || eb[r] >= n_blk) openm &= ~(1u << r);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < (COVL ? KPL : 1); r++) {
            const uint32_t e = (uint32_t)r * TH + (uint32_t)tid;
            if (e < cnt) {
                reinterpret_cast<uint4*>(&records[row_lo + e])[1] = tl[r];
                const uint32_t tx = lt[r] & 0xFFFu;
                if ((lt[r] & RUN_LT_NEWTILE) && tx >= 1u) bk.tile_first_run[ty * tiles_w + (tx - 1u)] = row_lo + e + 1u;   // (0 = the tile has no run)
            }
        }
    }
    if (LOCAL && n_slices == 1u) {
        s_txo = reinterpret_cast<uint16_t*>(NB_IN_IDLE ? idle + (CR_PIECE + 1) + (CR_PIECE + 4) / 2 : idle);
        if (!(COVL && blocks)) for (uint32_t e = tid; e < cnt; e += TH) s_txo[e] = (uint16_t)run_lt[row_lo + e];
        __syncthreads();
    }
    static_assert(!NB_IN_IDLE || (CR_PIECE + 1) + (CR_PIECE + 4) / 2 + CAP / 2 <= CAP, "group / tile_x / digest arrays share the idle sort buffer");
    for (uint32_t c0 = 0; c0 < m; c0 += CR_PIECE) {
        CRP_STAMP(4);                                                   // (rest of the previous piece: span compaction + stores)
        CarryLoad cl[CR_RPT];
#pragma unroll
        for (int k = 0; k < CR_RPT; k++)
            cl[k] = carry_load<LOCAL, COVL>(c0, tid * CR_RPT + k, m, row_lo, kbase, n_runs, ty, lkeys, sorted_keys, records, layer_sf, n_orders, s_txo, run_lt, s_cov, s_sf16);
        if (tid == 0) {                                                 // the run after the piece: only its group and tile_x
            const CarryLoad la = carry_load<LOCAL, COVL>(c0, CR_PIECE, m, row_lo, kbase, n_runs, ty, lkeys, sorted_keys, records, layer_sf, n_orders, s_txo, run_lt, s_cov, s_sf16);
            a_group[CR_PIECE] = la.active ? la.group : 0xFFFFFFFEu;
            a_txb[CR_PIECE] = (uint16_t)(la.active ? (la.tile & 0xFFFu) : 0u);
        }
        uint32_t group[CR_RPT], txb[CR_RPT], meta[CR_RPT];              // meta: sfl (11 bits) | unch << 11 | even_odd << 12 | active << 13
        uint64_t own_lo[CR_RPT], own_hi[CR_RPT];
#pragma unroll
        for (int k = 0; k < CR_RPT; k++) {
            const CarryLoad& cu = cl[k];
            const bool active = cu.active;
            group[k] = active ? cu.group : 0xFFFFFFFEu;
            txb[k] = active ? (cu.tile & 0xFFFu) : 0u;
            own_lo[k] = 0; own_hi[k] = 0;
            uint32_t sfl = 0, unch = 0;
            bool even_odd = false;
            if (active) {
                TileRecord* r = &records[cu.jrun];
                own_lo[k] = (uint64_t)cu.oc.x | ((uint64_t)cu.oc.y << 32); own_hi[k] = (uint64_t)cu.oc.z | ((uint64_t)cu.oc.w << 32);
                uint32_t sc = cu.sc;
                if (sc & RUN_OPEN) {                     // complete a run that crosses k_runs tiles with their edges
                    sc &= ~RUN_OPEN;
                    for (uint32_t bb = cu.seg_start / edge_segs + 1; bb < n_blk; bb++) {
                        const BlkEdge e = blk_edge[bb];
                        own_lo[k] = swar_add8(own_lo[k], (uint64_t)e.cov[0] | ((uint64_t)e.cov[1] << 32));
                        own_hi[k] = swar_add8(own_hi[k], (uint64_t)e.cov[2] | ((uint64_t)e.cov[3] << 32));
                        sc += e.cnt;
                        if (e.has_boundary) break;
                    }
                    r->seg_count = sc;
                }
                if (cu.lsf & LSF_VALID) {
                    // (the style bits and the "unchanged" flag are in the record already: the run kernel put them there, so that
                    //  a tile can classify its layer list without touching the style table — here they go into the span keys)
                    sfl = cu.lsf & ~LSF_VALID;
                    even_odd = (sfl & SF_EVENODD) != 0;
                    if (unchanged && unchanged[cu.layer]) unch = 1u;        // Layer::is_unchanged(cache_id)
                } else atomicOr(&info->error, 1u);
            }
            meta[k] = sfl | (unch << 11) | ((even_odd ? 1u : 0u) << 12) | ((active ? 1u : 0u) << 13);
            a_group[tid * CR_RPT + k] = group[k]; a_txb[tid * CR_RPT + k] = (uint16_t)txb[k];
        }
        lds_barrier();
        CRP_STAMP(3);                                                   // piece: gathers landed (records, covers, style summary)
        // ---- heads, and the lane's own segmented sums: (lo, hi) = sum since the lane's last head, f = the lane holds a head --
        const uint32_t prev_group0 = tid ? a_group[tid * CR_RPT - 1] : s_cgroup;
        uint32_t headm = 0;
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < CR_RPT; k++) {
            const bool active = (meta[k] >> 13) & 1u;
            const bool head = active && group[k] != (k ? group[k - 1] : prev_group0);
            if (head) { headm |= 1u << k; lo = own_lo[k]; hi = own_hi[k]; }
            else { lo = swar_add8(lo, own_lo[k]); hi = swar_add8(hi, own_hi[k]); }
        }
        uint32_t f = headm ? 1u : 0u;
        // ---- segmented inclusive scan of the lane sums over the piece --------------------------------------------------------
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t tl = __shfl_up(lo, d, 64), th = __shfl_up(hi, d, 64);
            const uint32_t tf = __shfl_up(f, d, 64);
            if (lane >= d) { if (!f) { lo = swar_add8(lo, tl); hi = swar_add8(hi, th); } f |= tf; }
        }
        if (lane == 63) { s_wlo[w] = lo; s_whi[w] = hi; s_wflag[w] = f; }
        lds_barrier();
        if (w == 0) {                                     // wave carry-ins: segmented scan of the 16 wave totals, seeded by the
            const bool in = lane < (TH / 64);              // piece carry — done by 16 lanes, not by one lane 16 times
            uint64_t l = in ? s_wlo[lane] : 0ull, h = in ? s_whi[lane] : 0ull;
            uint32_t fl = in ? s_wflag[lane] : 0u;
            const uint64_t seed_lo = s_clo, seed_hi = s_chi;
            if (lane == 0 && !fl) { l = swar_add8(seed_lo, l); h = swar_add8(seed_hi, h); }
#pragma unroll
            for (int d = 1; d < (TH / 64); d <<= 1) {
                const uint64_t tl = __shfl_up(l, d, 64), th = __shfl_up(h, d, 64);
                const uint32_t tf = __shfl_up(fl, d, 64);
                if (lane >= d) { if (!fl) { l = swar_add8(l, tl); h = swar_add8(h, th); } fl |= tf; }
            }
            uint64_t el = __shfl_up(l, 1, 64), eh = __shfl_up(h, 1, 64);      // exclusive: what runs into wave `lane`
            if (lane == 0) { el = seed_lo; eh = seed_hi; }
            if (in) { s_wlo[lane] = el; s_whi[lane] = eh; }
        }
        lds_barrier();
        if (!f) { lo = swar_add8(lo, s_wlo[w]); hi = swar_add8(hi, s_whi[w]); }     // lo/hi = inclusive value at the lane's last run
        // what runs into the lane's first run = inclusive value at the previous lane's last run
        uint64_t pl = __shfl_up(lo, 1, 64), ph = __shfl_up(hi, 1, 64);
        lds_barrier();                                    // s_wlo reused below: every wave has read its carry-in
        if (lane == 63) { s_wlo[w] = lo; s_whi[w] = hi; }
        lds_barrier();
        if (lane == 0) {
            if (w == 0) { pl = s_clo; ph = s_chi; } else { pl = s_wlo[w - 1]; ph = s_whi[w - 1]; }
        }
        // ---- second walk over the lane's runs, seeded: carry-in of every run, its carry-out, the span behind it --------------
        uint64_t out_lo[CR_RPT], out_hi[CR_RPT];
        uint32_t span_lo[CR_RPT], span_hi[CR_RPT], spanm = 0;
#pragma unroll
        for (int k = 0; k < CR_RPT; k++) {
            const bool active = (meta[k] >> 13) & 1u;
            if ((headm >> k) & 1u) { pl = 0; ph = 0; }
            if (active) {
                TileRecord* r = &records[cl[k].jrun];
                *reinterpret_cast<uint4*>(&r->cover[0]) = make_uint4((uint32_t)pl, (uint32_t)(pl >> 32), (uint32_t)ph, (uint32_t)(ph >> 32));
            }
            pl = swar_add8(pl, own_lo[k]); ph = swar_add8(ph, own_hi[k]);            // inclusive = this run's carry-out
            out_lo[k] = pl; out_hi[k] = ph;
            span_lo[k] = txb[k]; span_hi[k] = 0;
            if (active) {
                const uint32_t e = c0 + tid * CR_RPT + k;
                const bool last = e + 1 == m;                            // (slices end at layer boundaries: nothing of this group follows)
                const uint32_t ngroup = (k + 1 < CR_RPT) ? group[(k + 1) % CR_RPT] : a_group[tid * CR_RPT + CR_RPT];
                const uint32_t ntxb = (k + 1 < CR_RPT) ? txb[(k + 1) % CR_RPT] : a_txb[tid * CR_RPT + CR_RPT];
                const bool same_next = !last && ngroup == group[k];
                uint32_t sh = same_next ? ntxb - 1u : tiles_w;          // exclusive; next tile_x = txb_next - 1
                if (sh > tiles_w) sh = tiles_w;
                span_hi[k] = sh;                                         // span_lo = tile_x + 1
                // (a clip layer is never masked: its span is what expires or overwrites an earlier clip in the tiles to the
                //  right, painter/mod.rs:302-334 — dropping it would let a later clipped layer see a stale mask)
                const bool is_clip = (meta[k] & SF_IS_CLIP) != 0u;
                const bool empty = is_clip ? cover_is_empty(pl, ph, (meta[k] >> 12) & 1u)
                                           : cover_is_empty(pl & vis_lo, ph & vis_hi, (meta[k] >> 12) & 1u);
                if (!empty && span_lo[k] < sh) spanm |= 1u << k;
                // The first painted tile of a row (column `left_start`: 0, or the crop's first column) starts with EVERY layer
                // that has a segment to its left — empty cover or not (paint_tile_row collects them in a map and hands all of it
                // to LayerWorkbench::init, painter/mod.rs:500-522; the empty ones are dropped after that tile, :335-339).  Such an
                // entry paints nothing, but it counts: the tile's layer count and what the passes decide are remembered by a
                // buffer-layer cache (CachedTile).  With a cache attached the layer's last run left of the column therefore leaves
                // a one-tile span for it even when its carry is empty; without one nothing can observe the entry and it is skipped
                // (unless the channel order tells a folded tile from a painted one: api.cpp run_paint, fold_equals_paint).
                else if (empty && span_lo[k] <= left_start && left_start < sh) { spanm |= 1u << k; span_lo[k] = left_start; span_hi[k] = left_start + 1u; }
            }
        }
        // ordered compaction of the spans
        const uint32_t nsp = (uint32_t)__popc(spanm);
        const uint32_t nsp_incl = wave_incl_scan_u32(nsp);
        if (lane == 63) s_wspan[w] = nsp_incl;
        lds_barrier();
        uint32_t sbase = s_spans + nsp_incl - nsp, stot = 0;
#pragma unroll
        for (int i = 0; i < (TH / 64); i++) { const uint32_t t = s_wspan[i]; if (i < w) sbase += t; stot += t; }
#pragma unroll
        for (int k = 0; k < CR_RPT; k++) {
            if ((spanm >> k) & 1u) {
                const uint32_t si = row_lo + off + sbase;
                sbase++;
                const bool even_odd = (meta[k] >> 12) & 1u;
                const uint32_t sfl = meta[k] & 0x7FFu, unch = (meta[k] >> 11) & 1u;
                const uint32_t c4[4] = {(uint32_t)out_lo[k], (uint32_t)(out_lo[k] >> 32), (uint32_t)out_hi[k], (uint32_t)(out_hi[k] >> 32)};
                const uint32_t full = cover_full(c4, even_odd) ? SF_FULL : 0u;
                span_key[si] = ((uint64_t)(cl[k].layer | ((sfl | full) << 21)) << 32) | ((uint64_t)unch << 31) | ((uint64_t)span_lo[k] << 16) | span_hi[k];
                span_cov[si] = make_uint4(c4[0], c4[1], c4[2], c4[3]);
            }
        }
        lds_barrier();
        if (tid == TH - 1) { s_clo = out_lo[CR_RPT - 1]; s_chi = out_hi[CR_RPT - 1]; s_cgroup = group[CR_RPT - 1]; }   // only used when the piece is full
        if (tid == 0) s_spans += stot;
        lds_barrier();
    }
    if (tid == 0) {
        row_span_lo[ty * n_slices + slice] = row_lo + off; row_span_cnt[ty * n_slices + slice] = s_spans;   // (one pair per (row, slice))
        if (s_spans) atomicAdd(&info->n_spans, s_spans);
    }
    // ---- the slice's spans once more, by tile-column group (SpanGroups in common.h): wave w owns the groups w, w + 16, ... and
    //      compacts, in list order (= ascending layer), the spans that overlap each.  The (lo, hi) pairs are first brought into
    //      LDS in one round of coalesced loads (the sort buffers are idle by now); a painter then reads ~1/10 of the keys. ------
    if (!groups.tab) return;
    __syncthreads();                                                    // s_spans is final, every span of the slice is stored
    const uint32_t S = s_spans, sb0 = row_lo + off;
    const uint32_t G = (tiles_w + SPAN_GROUP_TILES - 1u) >> SPAN_GROUP_SHIFT;
    uint2* gt = groups.tab + (size_t)(ty * n_slices + slice) * G;
    // A row list that a painter reads in one round anyway (<= 256 keys: the 8192 x 8192 triangle scene has 115 per row) gains
    // nothing from group lists and this epilogue would cost its 512 small workgroups 12 us: such a slice says "none" and its
    // row's painters scan the row list.  (The other two conditions cannot happen: tile_x has 12 bits, spans <= runs <= CAP.)
    if (S * n_slices <= groups.min_row || G > 256u || (LOCAL && S > (uint32_t)CAP)) {
        for (uint32_t g = tid; g < G; g += TH) gt[g] = make_uint2(0u, SPAN_GROUP_NONE);
        return;
    }
    // the slice's keys: low words (unch | lo | hi) and high words (layer | SF_*) into the idle sort buffers, one round of loads
    uint32_t* s_lohi = LOCAL ? s_ka : nullptr;                          // (!LOCAL has no such buffers: it reads the keys in place)
    uint32_t* s_khi = LOCAL ? s_kb : nullptr;
    if (tid < 256) s_gcnt[tid] = 0;
    if (LOCAL) {
        for (uint32_t e = tid; e < S; e += TH) { const uint64_t k = span_key[sb0 + e]; s_lohi[e] = (uint32_t)k; s_khi[e] = (uint32_t)(k >> 32); }
    }
    __syncthreads();
    auto lohi_at = [&](uint32_t e) -> uint32_t { return LOCAL ? s_lohi[e] : (uint32_t)span_key[sb0 + e]; };
    for (uint32_t e = tid; e < S; e += TH) {                    // entries per group
        const uint32_t lh = lohi_at(e) & 0x7FFFFFFFu;
        const uint32_t lo = lh >> 16, hi = lh & 0xFFFFu;
        if (hi > lo) for (uint32_t g = lo >> SPAN_GROUP_SHIFT; g <= ((hi - 1u) >> SPAN_GROUP_SHIFT) && g < G; g++) atomicAdd(&s_gcnt[g], 1u);
    }
    __syncthreads();
    if (tid < 256) {                                                    // exclusive prefix over the groups
        const uint32_t c = (uint32_t)tid < G ? s_gcnt[tid] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) s_red[w] = inc;
        s_gpre[tid] = inc - c;
    }
    __syncthreads();
    // The slice's share of the pool is static — two entries per run of the slice, at twice its first run — so nothing is
    // allocated at run time; a slice whose spans overlap more groups than that keeps only its row list.
    const uint32_t total = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const uint32_t gbase = 2u * sb0;
    const bool fits = total <= 2u * m && 2ull * sb0 + total <= groups.cap;
    for (uint32_t g = w; g < G; g += (TH / 64)) {
        const uint32_t x0 = g << SPAN_GROUP_SHIFT, x1 = x0 + SPAN_GROUP_TILES;
        const uint32_t cntg = s_gcnt[g];                                // (an upper bound once spans are culled below: the list's slots)
        uint32_t first = gbase + s_gpre[g];                             // s_gpre is exclusive within the scanning wave of g
        for (uint32_t q = 0; q < (g >> 6); q++) first += s_red[q];
        if (!fits || !cntg) { if (lane == 0) gt[g] = fits ? make_uint2(first, 0u) : make_uint2(0u, SPAN_GROUP_NONE); continue; }
        // occlusion culling (PaintParams::cull): a span that covers the WHOLE group with a full cover of an opaque solid colour,
        // blended Over, unclipped, hides every lower layer in each of the group's tiles — their painters would drop those entries
        // one by one (k_paint_wave); they do not enter the group's list in the first place.
        uint32_t gocc = 0;
        if (cull) {
            for (uint32_t e0 = 0; e0 < S; e0 += 64) {
                const uint32_t e = e0 + (uint32_t)lane;
                const uint32_t lh = e < S ? (lohi_at(e) & 0x7FFFFFFFu) : 0u;
                if ((lh >> 16) <= x0 && (lh & 0xFFFFu) >= min(x1, tiles_w) && (lh & 0xFFFFu) > (lh >> 16)) {
                    const uint32_t khi = LOCAL ? s_khi[e] : (uint32_t)(span_key[sb0 + e] >> 32);
                    if (span_is_occluder(khi)) gocc = max(gocc, (khi & LAYER_MASK) + 1u);
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) gocc = max(gocc, (uint32_t)__shfl_xor(gocc, d, 64));
        }
        uint32_t pos = first;
        for (uint32_t e0 = 0; e0 < S; e0 += 64) {                       // ordered compaction: list order = ascending layer
            const uint32_t e = e0 + (uint32_t)lane;
            const uint32_t lhu = e < S ? lohi_at(e) : 0u;
            const uint32_t lh = lhu & 0x7FFFFFFFu;
            bool hit = (lh >> 16) < x1 && (lh & 0xFFFFu) > x0 && (lh & 0xFFFFu) > (lh >> 16);          // (as counted above)
            const uint32_t khi = hit ? (LOCAL ? s_khi[e] : (uint32_t)(span_key[sb0 + e] >> 32)) : 0u;
            if (hit && (khi & LAYER_MASK) + 1u < gocc) hit = false;
            const uint64_t bal = __ballot(hit);
            if (hit)
                groups.list[pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] =
                    make_uint4(khi, REF_SPAN | ((lhu >> 31) ? REF_UNCH : 0u) | (sb0 + e), lh, 0u);
            pos += (uint32_t)__popcll(bal);
        }
        if (lane == 0) gt[g] = make_uint2(first, pos - first);
    }
}

uint32_t carry_rows_local_cap() { return CR_CAP; }
uint32_t carry_rows_small_cap() { return CR_CAP_S; }
uint32_t carry_rows_half_cap() { return CR_CAP_H; }
uint32_t carry_rows_covl_cap() { return CR_CAP_L; }

void launch_carry_rows(hipStream_t s, bool local_sort, bool small, bool half, uint32_t n_slices, uint32_t bin_shift,
                       const uint64_t* sorted_run_keys, TileRecord* records,
                       const BlkEdge* blk_edge, DevCount n_segments, DevCount n_runs, const uint32_t* layer_sf,
                       uint32_t n_orders, uint32_t tiles_w, uint32_t tiles_h, const uint32_t* row_count,
                       uint32_t* row_span_lo, uint32_t* row_span_cnt, uint64_t* span_key, uint4* span_cov,
                       const uint8_t* unchanged, FrameInfo* info, uint32_t edge_segs, uint32_t vis_last, uint32_t row0, uint32_t row1,
                       SpanGroups groups, const uint32_t* run_lt, bool cull, uint32_t left_start, const uint32_t* row_base, bool covl, BlkRuns bk) {
    row1 = row1 < tiles_h ? row1 : tiles_h;
    if (tiles_h == 0 || row0 >= row1) return;             // (only the tile rows that are painted: the others' carries are never read)
    if (n_slices < 1u) n_slices = 1u;
    if (n_slices > CR_MAX_SLICES) n_slices = CR_MAX_SLICES;
    const dim3 grid((row1 - row0) * n_slices);
    const bool covl_variant = local_sort && covl && n_slices == 1u && (!small || half);
    if (bk.rec_sp && !covl_variant) bk = BlkRuns{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 256u};   // (the caller's condition)
    if (bk.round_tiles < 1u || bk.round_tiles > 256u || (bk.round_tiles & (bk.round_tiles - 1u))) bk.round_tiles = 256u;
#define CR_LAUNCH(L, C, R, T_, ...) FORMA_LAUNCH((k_carry_rows<L, C, R, T_ __VA_OPT__(,) __VA_ARGS__>), grid, dim3(T_), 0, s, sorted_run_keys, records, blk_edge, n_segments, \
                                              n_runs, layer_sf, n_orders, tiles_w, tiles_h, row_count, row_span_lo, row_span_cnt, span_key, \
                                              span_cov, unchanged, info, edge_segs, vis_last, n_slices, bin_shift, row0, groups, run_lt, cull ? 1u : 0u, left_start, row_base, bk)
    if (!local_sort) CR_LAUNCH(false, CR_CAP, 4, CR_THREADS);
    else if (small && half && covl && n_slices == 1u) CR_LAUNCH(true, CR_CAP_H, 4, CR_THREADS_H, true);
    else if (small && half) CR_LAUNCH(true, CR_CAP_H, 4, CR_THREADS_H);
    else if (small) CR_LAUNCH(true, CR_CAP_S, 2, CR_THREADS);
    else if (covl && n_slices == 1u) CR_LAUNCH(true, CR_CAP_L, 3, CR_THREADS, true);
    else CR_LAUNCH(true, CR_CAP, 4, CR_THREADS);
#undef CR_LAUNCH
}

// ================================================================================================
// the painter: one 256-lane workgroup per 16x16 tile, lane = local_y * 16 + local_x
// ================================================================================================
#define TSEG_CAP 512      // pixel segments of a tile kept in LDS (the rest is read from HBM/L2)
#define PBATCH   64       // painted entries whose per-layer data is staged in LDS at a time

// entry flags
#define EF_HAS_SEGS  0x001u
#define EF_FULL      0x002u
#define EF_IS_CLIP   0x004u
#define EF_CLIPPED   0x008u
#define EF_SOLID     0x010u
#define EF_OVER      0x020u
#define EF_OPAQUE    0x040u
#define EF_MASK      0x080u     // still enabled after the optimizer passes
#define EF_SKIPCLIP  0x100u     // passes_shared_state.skip_clipping contains this id
#define EF_EVENODD   0x200u

__device__ __forceinline__ float avx_min(float a, float b) { return a < b ? a : b; }   // _mm256_min_ps semantics
__device__ __forceinline__ float avx_max(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ int dpp_row_shr(int v, int n) {
    switch (n) {
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
        case 4: return __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
        default: return __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    }
}

__device__ __forceinline__ float coverage_of(int A, bool even_odd) {          // painter/mod.rs:76-94
    if (!even_odd) {
        float v = fabsf((float)A * (1.0f / 512.0f));
        return avx_max(avx_min(v, 1.0f), 0.0f);
    }
    int v = (A & 1023) - 512;
    v = v < 0 ? -v : v;
    return (float)(512 - v) * (1.0f / 512.0f);
}

__device__ __forceinline__ float lum3(float r, float g, float b) { return fmaf(r, 0.3f, fmaf(g, 0.59f, b * 0.11f)); }
__device__ __forceinline__ float sat3(float r, float g, float b) {
    return avx_max(r, avx_max(g, b)) - avx_min(r, avx_min(g, b));
}
__device__ __forceinline__ void clip_color(float& r, float& g, float& b) {     // styling.rs:364-396
    float l = lum3(r, g, b);
    float n = avx_min(r, avx_min(g, b));
    float x = avx_max(r, avx_max(g, b));
    float l_1 = l - 1.0f;
    float x_l_recip = 1.0f / (x - l);          // exact reciprocal (portable f32x8::recip, simd/auto.rs:727-730)
    float l_n_recip_l = (1.0f / (l - n)) * l;
    float c[3] = {r, g, b}, o[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float hi = fmaf(x_l_recip, fmaf(l, l_1 - c[k], c[k]), l);
        float lo = fmaf(l_n_recip_l, c[k] - l, l);
        float inner = (n < 0.0f) ? lo : c[k];
        o[k] = (1.0f < x) ? hi : inner;
    }
    r = o[0]; g = o[1]; b = o[2];
}
__device__ __forceinline__ void set_lum(float& r, float& g, float& b, float l) {
    float d = l - lum3(r, g, b);
    r += d; g += d; b += d;
    clip_color(r, g, b);
}
__device__ __forceinline__ void set_sat(float sat_dst, float sr, float sg, float sb, float* o) {   // styling.rs:408-435
    float src_min = avx_min(sr, avx_min(sg, sb));
    float src_max = avx_max(sr, avx_max(sg, sb));
    float src_mid = sr + sg + sb - src_min - src_max;
    bool lt = src_min < src_max;
    float sat_mid = lt ? (fmaf(sat_dst, -src_min, sat_dst * src_mid) / (src_max - src_min)) : 0.0f;
    float sat_max = lt ? sat_dst : 0.0f;
    float in[3] = {sr, sg, sb};
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] = (in[k] == src_max) ? sat_max : ((in[k] == src_min) ? 0.0f : sat_mid);
}

// blend_function! (cpu/painter/styling.rs:342-594), per pixel
__device__ __forceinline__ void blend_rgb(uint32_t mode, float dr, float dg, float db, float sr, float sg, float sb,
                                          float* o) {
    float d[3] = {dr, dg, db}, s[3] = {sr, sg, sb};
    switch (mode) {
        case 0: o[0] = sr; o[1] = sg; o[2] = sb; break;
        case 1:
#pragma unroll
            for (int k = 0; k < 3; k++) o[k] = d[k] * s[k];
            break;
        case 2:
#pragma unroll
            for (int k = 0; k < 3; k++) o[k] = fmaf(d[k], -s[k], d[k]) + s[k];
            break;
        case 3:
#pragma unroll
            for (int k = 0; k < 3; k++)
                o[k] = (d[k] <= 0.5f) ? (d[k] * s[k] * 2.0f) : (2.0f * (d[k] + s[k] - fmaf(d[k], s[k], 0.5f)));
            break;
        case 4:
#pragma unroll
            for (int k = 0; k < 3; k++) o[k] = avx_min(d[k], s[k]);
            break;
        case 5:
#pragma unroll
            for (int k = 0; k < 3; k++) o[k] = avx_max(d[k], s[k]);
            break;
        case 6:
#pragma unroll
            for (int k = 0; k < 3; k++) o[k] = (s[k] == 1.0f) ? 1.0f : avx_min(1.0f, d[k] / (1.0f - s[k]));
            break;
        case 7:
#pragma unroll
            for (int k = 0; k < 3; k++) o[k] = (s[k] == 0.0f) ? 0.0f : (1.0f - avx_min(1.0f, (1.0f - d[k]) / s[k]));
            break;
        case 8:
#pragma unroll
            for (int k = 0; k < 3; k++)
                o[k] = (s[k] <= 0.5f) ? (d[k] * s[k] * 2.0f) : (2.0f * (d[k] + s[k] - fmaf(d[k], s[k], 0.5f)));
            break;
        case 9:
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float dd = (d[k] <= 0.25f) ? (fmaf(fmaf(16.0f, d[k], -12.0f), d[k], 4.0f) * d[k]) : sqrtf(d[k]);
                float m = fmaf(2.0f, s[k], -1.0f);
                o[k] = (s[k] <= 0.5f) ? fmaf(d[k] * (1.0f - d[k]), m, d[k]) : fmaf(dd - d[k], m, d[k]);
            }
            break;
        case 10:
#pragma unroll
            for (int k = 0; k < 3; k++) o[k] = fabsf(d[k] - s[k]);
            break;
        case 11:
#pragma unroll
            for (int k = 0; k < 3; k++) o[k] = fmaf(-2.0f * d[k], s[k], d[k]) + s[k];
            break;
        case 12: { set_sat(sat3(dr, dg, db), sr, sg, sb, o); set_lum(o[0], o[1], o[2], lum3(dr, dg, db)); break; }
        case 13: { set_sat(sat3(sr, sg, sb), dr, dg, db, o); set_lum(o[0], o[1], o[2], lum3(dr, dg, db)); break; }
        case 14: { o[0] = sr; o[1] = sg; o[2] = sb; set_lum(o[0], o[1], o[2], lum3(dr, dg, db)); break; }
        default: { o[0] = dr; o[1] = dg; o[2] = db; set_lum(o[0], o[1], o[2], lum3(sr, sg, sb)); break; }
    }
}

// ---- scalar BlendMode::blend (styling.rs:195-340): only used to fold all-solid tiles ------------------
struct Col { float r, g, b, a; };
__device__ float sc_ch(Col k, int i) { return i == 0 ? k.r : (i == 1 ? k.g : k.b); }
__device__ float sc_lum(Col k) { return fmaf(k.r, 0.3f, fmaf(k.g, 0.59f, k.b * 0.11f)); }
__device__ float sc_min(Col k) { return fminf(k.r, fminf(k.g, k.b)); }
__device__ float sc_max(Col k) { return fmaxf(k.r, fmaxf(k.g, k.b)); }
__device__ float sc_clip_color(int i, Col k) {
    float l = sc_lum(k), n = sc_min(k), x = sc_max(k);
    float v = sc_ch(k, i);
    if (n < 0.0f) { float t = (1.0f / (l - n)) * l; v = fmaf(t, v - l, l); }
    if (x > 1.0f) { float l_1 = l - 1.0f; float xr = 1.0f / (x - l); v = fmaf(xr, fmaf(l, l_1 - v, v), l); }
    return v;
}
__device__ float sc_set_lum(int i, Col k, float l) { float dd = l - sc_lum(k); k.r += dd; k.g += dd; k.b += dd; return sc_clip_color(i, k); }
__device__ Col sc_set_sat(Col k, float s) {
    float cc[3] = {k.r, k.g, k.b};
    int imin, imid, imax;
    bool a = cc[0] < cc[1], b = cc[0] < cc[2], cq = cc[1] < cc[2];
    if (a && b && cq) { imin = 0; imid = 1; imax = 2; }
    else if (a && b && !cq) { imin = 0; imid = 2; imax = 1; }
    else if (a && !b) { imin = 2; imid = 0; imax = 1; }
    else if (!a && b && cq) { imin = 1; imid = 0; imax = 2; }
    else if (!a && !cq) { imin = 2; imid = 1; imax = 0; }
    else { imin = 1; imid = 2; imax = 0; }
    if (cc[imax] > cc[imin]) { cc[imid] = fmaf(s, cc[imid], -s * cc[imin]) / (cc[imax] - cc[imin]); cc[imax] = s; }
    else { cc[imid] = 0.0f; cc[imax] = 0.0f; }
    cc[imin] = 0.0f;
    Col o = {cc[0], cc[1], cc[2], k.a};
    return o;
}
__device__ float sc_blend_fn(uint32_t mode, int c, Col dst, Col src) {
    float d = sc_ch(dst, c), s = sc_ch(src, c);
    switch (mode) {
        case 0: return s;
        case 1: return d * s;
        case 2: return d + s - (d * s);
        case 3: return d <= 0.5f ? s * (2.0f * d) : (s + (2.0f * d - 1.0f) - (s * (2.0f * d - 1.0f)));   // hard_light(src, dst)
        case 4: return fminf(d, s);
        case 5: return fmaxf(d, s);
        case 6: return d == 0.0f ? 0.0f : (s == 1.0f ? 1.0f : fminf(1.0f, d / (1.0f - s)));
        case 7: return d == 1.0f ? 1.0f : (s == 0.0f ? 0.0f : 1.0f - fminf(1.0f, (1.0f - d) / s));
        case 8: return s <= 0.5f ? d * (2.0f * s) : (d + (2.0f * s - 1.0f) - (d * (2.0f * s - 1.0f)));   // hard_light(dst, src)
        case 9: {
            float dd = d <= 0.25f ? ((16.0f * d - 12.0f) * d + 4.0f) * d : sqrtf(d);
            return s <= 0.5f ? d - (1.0f - 2.0f * s) * d * (1.0f - d) : d + (2.0f * s - 1.0f) * (dd - d);
        }
        case 10: return fabsf(d - s);
        case 11: return d + s - 2.0f * d * s;
        case 12: return sc_set_lum(c, sc_set_sat(src, sc_max(dst) - sc_min(dst)), sc_lum(dst));
        case 13: return sc_set_lum(c, sc_set_sat(dst, sc_max(src) - sc_min(src)), sc_lum(dst));
        case 14: return sc_set_lum(c, src, sc_lum(dst));
        default: return sc_set_lum(c, dst, sc_lum(src));
    }
}
__device__ Col sc_blend(uint32_t mode, Col dst, Col src) {
    float ida = 1.0f - dst.a, k1 = ida * src.a, isa = 1.0f - src.a, k2 = dst.a * src.a;
    float cr = fmaf(src.r, k1, sc_blend_fn(mode, 0, dst, src) * k2);
    float cg = fmaf(src.g, k1, sc_blend_fn(mode, 1, dst, src) * k2);
    float cb = fmaf(src.b, k1, sc_blend_fn(mode, 2, dst, src) * k2);
    Col o = {fmaf(dst.r, isa, cr), fmaf(dst.g, isa, cg), fmaf(dst.b, isa, cb), fmaf(dst.a, isa, src.a)};
    return o;
}

// ---- encode (painter/mod.rs:96-162) ---------------------------------------------------------------------
__device__ __forceinline__ float linear_to_srgb(float l) {
    float s = sqrtf(l), s3 = l * s;
    float m = l * 12.92f;
    float n = fmaf(0.20101772f, s3, fmaf(-0.51280147f, l, fmaf(1.344401f, s, -0.030656587f)));
    return (l <= 0.0031308f) ? m : n;
}
__device__ __forceinline__ uint32_t to_u8_x8(float v) {        // to_u32x8: clamp = min(max).max(min)
    float scaled = avx_max(avx_min(v * 255.0f, 255.0f), 0.0f);
    return __float_as_uint(scaled + __uint_as_float(0x4B000000u)) & 0xFFu;
}
__device__ __forceinline__ uint32_t to_u8_x4(float v) {        // to_u32x4: clamp = min(max(v, 0), 255)
    float scaled = avx_min(avx_max(v * 255.0f, 0.0f), 255.0f);
    return __float_as_uint(scaled + __uint_as_float(0x4B000000u)) & 0xFFu;
}
__device__ __forceinline__ float sel_channel(uint32_t c, float r, float g, float b, float a) {   // channel.rs:44-55
    switch (c) { case 0: return r; case 1: return g; case 2: return b; case 3: return a; case 4: return 0.0f; default: return 1.0f; }
}

// ---- fills (cpu/painter/styling.rs:58-193), per pixel -----------------------------------------------------
template <typename WP>
__device__ __forceinline__ void gradient_at(WP w, uint32_t fill, uint32_t nstops, float x,
                                            float ybase, int j, float* out) {
    float sx = __uint_as_float(w[2]), sy = __uint_as_float(w[3]), ex = __uint_as_float(w[4]), ey = __uint_as_float(w[5]);
    float dx = ex - sx, dy = ey - sy;
    float dot = dx * dx + dy * dy;
    float dot_recip = 1.0f / dot;
    float t;
    if (fill == FORMA_FILL_LINEAR) {
        float tx = (x - sx) * dx * dot_recip;
        float ty = ybase - sy;
        t = fmaf(((float)j + ty) * dy, dot_recip, tx);
    } else {
        float px = x - sx;
        float px2 = px * px;
        float py = (float)j + (ybase - sy);
        t = sqrtf(fmaf(py, py, px2) * dot_recip);
    }
    const auto st = w + 6;
    uint32_t ch[4] = {0, 0, 0, 0};
    bool acc = t <= __uint_as_float(st[4]);
    if (acc) { ch[0] |= st[0]; ch[1] |= st[1]; ch[2] |= st[2]; ch[3] |= st[3]; }
    float start_stop = 0.0f;
    uint32_t sc[4] = {st[0], st[1], st[2], st[3]};
    for (uint32_t k = 1; k < nstops; k++) {
        const auto q = st + 5 * k;
        float end_stop = __uint_as_float(q[4]);
        bool mask = acc ^ (t < end_stop);
        if (mask) {
            float d = end_stop - start_stop;
            float lt = (t - start_stop) * (1.0f / d);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float s0 = __uint_as_float(sc[c]);
                ch[c] |= __float_as_uint(fmaf(lt, __uint_as_float(q[c]), fmaf(-lt, s0, s0)));
            }
            acc = true;
        }
        start_stop = end_stop;
        sc[0] = q[0]; sc[1] = q[1]; sc[2] = q[2]; sc[3] = q[3];
    }
    if (!acc) {
        const auto q = st + 5 * (nstops - 1);
        ch[0] |= q[0]; ch[1] |= q[1]; ch[2] |= q[2]; ch[3] |= q[3];
    }
    out[0] = __uint_as_float(ch[0]); out[1] = __uint_as_float(ch[1]); out[2] = __uint_as_float(ch[2]); out[3] = __uint_as_float(ch[3]);
}

__device__ __forceinline__ float f16b_to_f32(uint32_t h) { return h != 0 ? __uint_as_float(0x38000000u + (h << 13)) : 0.0f; }

__device__ __forceinline__ void texture_at(const uint32_t* __restrict__ w, const forma_image_t* __restrict__ images,
                                           const uint16_t* __restrict__ texels, float x, float y, float* out) {
    const forma_image_t im = images[w[8]];
    float max_x = (float)im.width - 1.0f, max_y = (float)im.height - 1.0f;
    float ux = __uint_as_float(w[2]), uy = __uint_as_float(w[3]), vx = __uint_as_float(w[4]), vy = __uint_as_float(w[5]);
    float tx = __uint_as_float(w[6]), ty = __uint_as_float(w[7]);
    float fx = fmaf(x, ux, fmaf(vx, y, tx));
    float fy = fmaf(x, uy, fmaf(vy, y, ty));
    float cx = avx_max(avx_min(fx, max_x), 0.0f), cy = avx_max(avx_min(fy, max_y), 0.0f);
    uint32_t ix = (uint32_t)(int)cx, iy = (uint32_t)(int)cy;
    uint32_t off = iy * im.width + ix;
    const uint16_t* p = texels + 4 * (im.texel_offset + off);
    out[0] = f16b_to_f32(p[0]); out[1] = f16b_to_f32(p[1]); out[2] = f16b_to_f32(p[2]); out[3] = f16b_to_f32(p[3]);
}

// texture_at with the image descriptor already fetched (once per layer instead of once per pixel)
template <typename WP>
__device__ __forceinline__ void texture_at_im(WP w, const forma_image_t im, const uint16_t* __restrict__ texels, float x, float y,
                                              float* out) {
    float max_x = (float)im.width - 1.0f, max_y = (float)im.height - 1.0f;
    float ux = __uint_as_float(w[2]), uy = __uint_as_float(w[3]), vx = __uint_as_float(w[4]), vy = __uint_as_float(w[5]);
    float tx = __uint_as_float(w[6]), ty = __uint_as_float(w[7]);
    float fx = fmaf(x, ux, fmaf(vx, y, tx));
    float fy = fmaf(x, uy, fmaf(vy, y, ty));
    float cx = avx_max(avx_min(fx, max_x), 0.0f), cy = avx_max(avx_min(fy, max_y), 0.0f);
    uint32_t ix = (uint32_t)(int)cx, iy = (uint32_t)(int)cy;
    uint32_t off = iy * im.width + ix;
    const uint16_t* p = texels + 4 * (im.texel_offset + off);
    out[0] = f16b_to_f32(p[0]); out[1] = f16b_to_f32(p[1]); out[2] = f16b_to_f32(p[2]); out[3] = f16b_to_f32(p[3]);
}

// gradient_at for the wave painter: the layer's style words sit in the wave's LDS window `ws` (words 0..5 = header and
// geometry, then WSTOPS stops of five words).  Called by ALL 64 lanes (the caller selects afterwards), so a gradient with
// more stops than the window holds re-fills the window from `wg` as it goes and puts the first stops back at the end.
#define WSTYLE 64         // style words of a non-solid layer the wave painter keeps in LDS: header, geometry, 11 gradient stops
#define WSTOPS ((WSTYLE - 6) / 5)
__device__ __forceinline__ void gradient_at_lds(uint32_t* ws, const uint32_t* __restrict__ wg, uint32_t fill, uint32_t nstops, float x,
                                                float ybase, int j, int lane, float* out) {
    float sx = __uint_as_float(ws[2]), sy = __uint_as_float(ws[3]), ex = __uint_as_float(ws[4]), ey = __uint_as_float(ws[5]);
    float dx = ex - sx, dy = ey - sy;
    float dot = dx * dx + dy * dy;
    float dot_recip = 1.0f / dot;
    float t;
    if (fill == FORMA_FILL_LINEAR) {
        float tx = (x - sx) * dx * dot_recip;
        float ty = ybase - sy;
        t = fmaf(((float)j + ty) * dy, dot_recip, tx);
    } else {
        float px = x - sx;
        float px2 = px * px;
        float py = (float)j + (ybase - sy);
        t = sqrtf(fmaf(py, py, px2) * dot_recip);
    }
    const uint32_t* st = ws + 6;
    uint32_t ch[4] = {0, 0, 0, 0};
    bool acc = t <= __uint_as_float(st[4]);
    if (acc) { ch[0] |= st[0]; ch[1] |= st[1]; ch[2] |= st[2]; ch[3] |= st[3]; }
    float start_stop = 0.0f;
    uint32_t sc[4] = {st[0], st[1], st[2], st[3]};
    uint32_t base = 0;                                                  // first stop of the window
    for (uint32_t k = 1; k < nstops; k++) {
        if (k - base >= (uint32_t)WSTOPS) {                             // (uniform) next window
            base = k;
            wave_lds_fence();
            if (lane < WSTOPS * 5) ws[6 + lane] = 5u * base + (uint32_t)lane < 5u * nstops ? wg[6 + 5 * base + lane] : 0u;
            wave_lds_fence();
        }
        const uint32_t* q = st + 5 * (k - base);
        float end_stop = __uint_as_float(q[4]);
        bool mask = acc ^ (t < end_stop);
        if (mask) {
            float d = end_stop - start_stop;
            float lt = (t - start_stop) * (1.0f / d);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float s0 = __uint_as_float(sc[c]);
                ch[c] |= __float_as_uint(fmaf(lt, __uint_as_float(q[c]), fmaf(-lt, s0, s0)));
            }
            acc = true;
        }
        start_stop = end_stop;
        sc[0] = q[0]; sc[1] = q[1]; sc[2] = q[2]; sc[3] = q[3];
    }
    if (!acc) {
        const uint32_t* q = st + 5 * (nstops - 1 - base);
        ch[0] |= q[0]; ch[1] |= q[1]; ch[2] |= q[2]; ch[3] |= q[3];
    }
    if (base) {                                                         // the window goes back to the first stops
        wave_lds_fence();
        if (lane < WSTOPS * 5) ws[6 + lane] = wg[6 + lane];
        wave_lds_fence();
    }
    out[0] = __uint_as_float(ch[0]); out[1] = __uint_as_float(ch[1]); out[2] = __uint_as_float(ch[2]); out[3] = __uint_as_float(ch[3]);
}

// The span lists of a tile row: one per slice of the carry pre-pass (launch_carry_rows), ascending layers from slice to slice.
// The painters see them as ONE logical list [0, total): logical index -> position in the span arrays by <= 7 compares
// (uniform; a row with one slice pays nothing).
// NS = 1: the kernel variant for frames whose rows have ONE slice (every whole-frame render: slices are for multi-GPU bands) —
// the mapping is an add and the lists cost 2 scalar registers instead of 18.
template <int NS> struct SpanListsT { uint32_t n, total; uint32_t pre[NS], base[NS]; };
typedef SpanListsT<CR_MAX_SLICES> SpanLists;
template <int NS>
__device__ __forceinline__ SpanListsT<NS> load_span_lists_t(const uint32_t* __restrict__ row_span_lo, const uint32_t* __restrict__ row_span_cnt,
                                                            uint32_t ty, uint32_t n_slices) {
    SpanListsT<NS> L;
    L.n = n_slices; L.total = 0;
    uint32_t b[NS], c[NS];
#pragma unroll
    for (int q = 0; q < NS; q++) {                         // (all loads in flight together)
        const bool in = (uint32_t)q < n_slices;
        b[q] = in ? row_span_lo[ty * n_slices + q] : 0u; c[q] = in ? row_span_cnt[ty * n_slices + q] : 0u;
    }
#pragma unroll
    for (int q = 0; q < NS; q++) { L.pre[q] = L.total; L.base[q] = b[q]; L.total += c[q]; }
    return L;
}
__device__ __forceinline__ SpanLists load_span_lists(const uint32_t* __restrict__ row_span_lo, const uint32_t* __restrict__ row_span_cnt,
                                                     uint32_t ty, uint32_t n_slices) {
    return load_span_lists_t<CR_MAX_SLICES>(row_span_lo, row_span_cnt, ty, n_slices);
}
// the same for one tile-column group: tab points at the group's entry of slice 0, the slices are n_groups entries apart.
// total = SPAN_GROUP_NONE: some slice of the row has no group lists (pool full) — the caller scans the row lists instead.
template <int NS>
__device__ __forceinline__ SpanListsT<NS> load_group_lists(const uint2* __restrict__ tab, uint32_t n_groups, uint32_t n_slices) {
    SpanListsT<NS> L;
    L.n = n_slices; L.total = 0;
    uint2 t[NS];
#pragma unroll
    for (int q = 0; q < NS; q++) t[q] = (uint32_t)q < n_slices ? tab[(size_t)q * n_groups] : make_uint2(0u, 0u);
    bool none = false;
#pragma unroll
    for (int q = 0; q < NS; q++) { L.pre[q] = L.total; L.base[q] = t[q].x; L.total += t[q].y; none |= t[q].y == SPAN_GROUP_NONE; }
    if (none) L.total = SPAN_GROUP_NONE;
    return L;
}
template <int NS>
__device__ __forceinline__ uint32_t span_phys(const SpanListsT<NS>& L, uint32_t i) {
    uint32_t p = L.base[0] + i;
#pragma unroll
    for (int q = 1; q < NS; q++) {
        if ((uint32_t)q >= L.n) break;
        if (i >= L.pre[q]) p = L.base[q] + (i - L.pre[q]);
    }
    return p;
}

// One tile, one 256-lane workgroup.  The tile's layer list lives in three caller-provided arrays: e_key (4 x stage entries:
// the four waves stage their span hits there, then it holds the merged list, cap entries), e_tmp and e_flag (cap entries
// each).  k_paint_deep passes LDS (cap 4096); a tile that does not fit is recorded as {tile, entries} in `over2`, and
// k_paint_huge paints it with lists in global memory sized for exactly that tile — the reference has no limit on the layers
// of a tile (LayerWorkbench::populate_layers, layer_workbench/mod.rs:250-278) and neither has this path.
__device__ __forceinline__ void paint_tile(const uint32_t MAXE, const uint32_t STAGE, uint64_t* e_key, uint64_t* e_tmp, uint32_t* e_flag,
                                           const PaintParams& P, const uint32_t tile, const uint64_t* __restrict__ sorted,
                                           const TileRecord* __restrict__ records, const uint32_t n_runs_frame,
                                           const uint32_t* __restrict__ tile_first_run,
                                           const uint32_t* __restrict__ row_span_lo,
                                           const uint32_t* __restrict__ row_span_cnt,
                                           const uint64_t* __restrict__ span_key, const uint4* __restrict__ span_cov,
                                           const uint4* __restrict__ layer_col,
                                           const uint32_t* __restrict__ style_offsets,
                                           const uint32_t* __restrict__ style_words,
                                           const forma_image_t* __restrict__ images,
                                           const uint16_t* __restrict__ texels, uint8_t* __restrict__ image,
                                           TileCacheArgs cache, FrameInfo* __restrict__ info, uint32_t* __restrict__ over2_n,
                                           uint32_t* __restrict__ over2_list) {
    __shared__ int cells[2][256];             // double-buffered coverage cells: one barrier per layer with segments
    __shared__ uint64_t t_seg[TSEG_CAP];      // the tile's own pixel segments (contiguous in the sorted stream)
    __shared__ uint4 b_cov[PBATCH];           // per painted entry of the current batch: carry-in cover,
    __shared__ uint4 b_col[PBATCH];           //   style words 2..5 (solid colour / gradient geometry),
    __shared__ uint32_t b_seg0[PBATCH], b_nseg[PBATCH], b_flag[PBATCH], b_layer[PBATCH];
    __shared__ uint32_t s_solid, s_solid_bytes, s_seg0, s_seg1, s_over;
    __shared__ uint32_t s_wcnt[4], s_wblk[4];

    const uint32_t ty = tile / P.tiles_w, tx = tile - ty * P.tiles_w;
    if (tx < P.crop_x0 || tx >= P.crop_x1 || ty < P.crop_y0 || ty >= P.crop_y1) return;   // print_row :588-592, :525-529
    const int tid = threadIdx.x;
    const int lx = tid & 15, ly = tid >> 4;
    const int lane = tid & 63, wv = tid >> 6;

    // ---- this tile's layer list (LayerWorkbench::populate_layers, layer_workbench/mod.rs:250-278):
    //      its own runs (contiguous records, ascending layer) merged with the row's spans that cross it.
    //      entry = (layer << 32) | ref, ref = run index, or 0x80000000 | span index ------------------------------
    const uint32_t my_tile_key = ((ty + 1u) << 12) | (tx + 1u);
    const uint32_t n_runs = PAINT_RUN_END(P, ty, n_runs_frame);
    // first round of loads, all independent: where this tile's runs start, where this row's spans are
    const uint32_t j0 = tile_first_run[tile] - 1u;                      // 0 stored = no run -> FORMA_NONE
    const SpanLists SL = load_span_lists(row_span_lo, row_span_cnt, ty, P.n_slices);
    const uint32_t sc = SL.total;
    uint32_t na = 0;
    if (tid == 0) { s_seg0 = 0; s_seg1 = 0; s_over = 0; }
    // second round, issued together: the run records probed below and this wave's first 256 span keys
    const uint32_t sq = (sc + 3u) / 4u;
    const uint32_t c_lo = min(sc, (uint32_t)wv * sq), c_hi = min(sc, (uint32_t)(wv + 1) * sq);
    uint64_t sk[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const uint32_t i = c_lo + u * 64 + lane; sk[u] = i < c_hi ? span_key[span_phys(SL, i)] : 0ull; }
    __syncthreads();
    if (j0 != FORMA_NONE) {
        for (uint32_t c = 0;; c += 256) {                              // a tile's runs are contiguous from j0
            const uint32_t j = j0 + c + tid;
            bool mine = false; uint32_t layer = 0, st = 0, cn = 0, unch = 0;
            if (j < n_runs) { const TileRecord* r = &records[j]; mine = (r->tile & 0x7FFFFFFFu) == my_tile_key; layer = r->layer; unch = (r->tile >> 31) ? REF_UNCH : 0u; st = r->seg_start; cn = r->seg_count; }
            if (mine && c + tid < MAXE) e_tmp[c + tid] = ((uint64_t)layer << 32) | unch | j;
            const uint64_t mb = __ballot(mine);
            if (mine && c + tid == 0) s_seg0 = st;
            // the last run of the tile: mine, and the next record is not (runs of a tile are a prefix of the probes)
            const bool next_mine = (mb >> ((lane + 1) & 63)) & 1ull;
            if (mine && lane < 63 && !next_mine) s_seg1 = st + cn;
            if (mine && lane == 63) {
                const uint32_t jn = j + 1;
                if (jn >= n_runs || (records[jn].tile & 0x7FFFFFFFu) != my_tile_key) s_seg1 = st + cn;
            }
            const uint32_t got = (uint32_t)__syncthreads_count(mine ? 1 : 0);
            na += got;
            if (got < 256u) break;
        }
    }
    // ---- the tile's own segments, once (they are contiguous in the sorted stream); the loads fly during the span scan
    const uint32_t seg0 = s_seg0, seg1 = s_seg1;
    uint64_t tsv[TSEG_CAP / 256];
#pragma unroll
    for (int u = 0; u < TSEG_CAP / 256; u++) { const uint32_t i = u * 256 + tid; tsv[u] = i < seg1 - seg0 ? sorted[seg0 + i] : 0ull; }
    // ---- spans of this tile row that cross the tile.  Each wave scans its own quarter of the row's span list and
    //      stages its hits in order (no barrier per probe); the quarters are then concatenated. ---------------------
    uint32_t nb = 0;
    {
        uint32_t cw = 0;
        for (uint32_t c = c_lo; c < c_hi; c += 256) {
            if (c != c_lo) {
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t i = c + u * 64 + lane; sk[u] = i < c_hi ? span_key[span_phys(SL, i)] : 0ull; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t lo = (uint32_t)(sk[u] >> 16) & 0x7FFFu, hi = (uint32_t)sk[u] & 0xFFFFu;   // padding: lo = hi = 0
                const bool hit = tx >= lo && tx < hi;
                const uint64_t bal = __ballot(hit);
                if (hit) {
                    const uint32_t pos = cw + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (pos < (uint32_t)STAGE) e_key[(size_t)wv * STAGE + pos] = (sk[u] & 0xFFFFFFFF00000000ull) | REF_SPAN | (((uint32_t)sk[u] >> 31) ? REF_UNCH : 0u) | span_phys(SL, c + u * 64 + lane);
                }
                cw += (uint32_t)__popcll(bal);
            }
        }
        if (lane == 0) s_wcnt[wv] = cw;
#pragma unroll
        for (int u = 0; u < TSEG_CAP / 256; u++) t_seg[u * 256 + tid] = tsv[u];
        __syncthreads();
        uint32_t base = na, over = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t t = s_wcnt[i]; if (i < wv) base += t; nb += t; over |= t > (uint32_t)STAGE ? 1u : 0u; }
        if (over) s_over = 1;                                         // more crossing spans in one quarter than a wave may stage
        else for (uint32_t i = lane; i < cw; i += 64) if (base + i < MAXE) e_tmp[base + i] = e_key[(size_t)wv * STAGE + i];
        __syncthreads();
    }
    const uint32_t ne = na + nb;                                      // the true length of the list, staged or not
    uint64_t* keys = e_key;
    uint32_t* flags = e_flag;
    if (ne > MAXE || s_over) {                                        // does not fit the caller's lists
        if (tid == 0) {
            if (over2_n) { const uint32_t q = atomicAdd(over2_n, 1u); over2_list[2 * q] = tile; over2_list[2 * q + 1] = ne; atomicOr(&info->error, 8u); }
            else atomicOr(&info->error, 2u);                          // (cannot happen: k_paint_huge sizes its lists from `ne`)
        }
        return;
    }
    // merge by layer: a (tile, layer) pair is either a run or a span, so layers are unique across both lists
    for (uint32_t i = tid; i < ne; i += 256) {
        const uint64_t k = e_tmp[i];
        const uint32_t layer = (uint32_t)(k >> 32) & LAYER_MASK;
        uint32_t lo, hi;                                            // # entries of the OTHER list with a smaller layer
        if (i < na) { lo = na; hi = ne; } else { lo = 0; hi = na; }
        const uint32_t other0 = lo;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (((uint32_t)(e_tmp[mid] >> 32) & LAYER_MASK) < layer) lo = mid + 1; else hi = mid; }
        const uint32_t rank = (i < na ? i : i - na) + (lo - other0);
        keys[rank] = k;
    }
    __syncthreads();
    // ---- per-entry facts the optimizer passes need: decoded from the SF_* bits the carry pre-pass put in the keys ----
    for (uint32_t i = tid; i < ne; i += 256) {
        const uint64_t k = keys[i];
        const uint32_t sfl = (uint32_t)(k >> 53), ref = (uint32_t)k;
        uint32_t f = EF_MASK;
        if (sfl & SF_EVENODD) f |= EF_EVENODD;
        if (!(ref & 0x80000000u)) f |= EF_HAS_SEGS;                          // a run always owns segments
        else if (sfl & SF_FULL) f |= EF_FULL;
        if (sfl & SF_IS_CLIP) f |= EF_IS_CLIP;
        else {
            if (sfl & SF_CLIPPED) f |= EF_CLIPPED;
            if (((sfl >> SF_FILL_SHIFT) & 3u) == FORMA_FILL_SOLID) f |= EF_SOLID;
            if (sfl & SF_OPAQUE) f |= EF_OPAQUE;
            if (((sfl >> SF_BLEND_SHIFT) & 15u) == 0u) f |= EF_OVER;
        }
        flags[i] = f;
    }
    __syncthreads();

    const Col clear = {P.clear[0], P.clear[1], P.clear[2], P.clear[3]};
    // ---- buffer-layer cache: tile_unchanged_pass (passes/tile_unchanged.rs), the first pass ------------------------------
    bool layers_were_removed = true;                                    // PassesSharedState default
    uint32_t ct_tags = 0, ct_solid = 0;
    if (cache.tiles) {
        const uint2 ct = cache.tiles[tile];
        ct_solid = ct.y;
        int all_unch = 1;
        for (uint32_t i = tid; i < ne; i += 256) if (!((uint32_t)keys[i] & REF_UNCH)) all_unch = 0;
        all_unch = __syncthreads_and(all_unch);
        const bool had = (ct.x & 2u) != 0;
        const uint32_t prev = ct.x >> 8;
        ct_tags = (ct.x & 1u) | 2u;                                     // update_layer_count(Some(layers))
        bool tile_unchanged = false;
        if (had) { layers_were_removed = ne < prev; tile_unchanged = prev == ne && all_unch; }
        if (P.clear_unchanged && tile_unchanged) {                      // TileWriteOp::None
            __syncthreads();                                            // every thread has read ct
            if (tid == 0) cache.tiles[tile] = make_uint2(ct_tags | (ne << 8), ct_solid);
            return;
        }
    }
    // ---- optimizer passes (layer_workbench/passes/*.rs) -----------------------------------------------------------
    if (P.scene_has_clips && tid == 0) {                               // skip_trivial_clips_pass: serial (clip state machine)
        bool has = false, c_full = false, c_used = false; uint32_t c_last = 0, c_i = 0;
        for (uint32_t i = 0; i < ne; i++) {
            uint32_t f = flags[i];
            if (!(f & EF_MASK)) continue;
            uint32_t id = (uint32_t)(keys[i] >> 32) & LAYER_MASK;
            if (f & EF_IS_CLIP) {
                c_full = (f & EF_FULL) != 0;
                c_last = id + style_words[style_offsets[id] + 1]; c_i = i; c_used = false; has = true;
                if (c_full) { f &= ~EF_MASK; flags[i] = f; }
            }
            if (!(f & EF_IS_CLIP) && (f & EF_CLIPPED)) {
                if (has && id <= c_last) { if (c_full) { f |= EF_SKIPCLIP; flags[i] = f; } else c_used = true; }
                else { f &= ~EF_MASK; flags[i] = f; }
            }
            if (has && id > c_last) { has = false; if (!c_used) flags[c_i] &= ~EF_MASK; }
        }
        if (has && !c_used) flags[c_i] &= ~EF_MASK;
    }
    if (P.scene_has_clips) __syncthreads();
    // skip_fully_covered_layers_pass (passes/skip_fully_covered_layers.rs), data-parallel: walking the list top-down,
    // the reference stops at the topmost FULL, unclipped, solid, Over, opaque layer ("cover"); it may fold the tile to
    // one colour only if nothing above that layer is clipped or partial ("blocker").
    uint32_t top = 0, blk = 0;                                          // index + 1 of the topmost cover / blocker
    for (uint32_t i = tid; i < ne; i += 256) {
        const uint32_t f = flags[i];
        if (!(f & EF_MASK)) continue;
        const bool clipped = !(f & EF_IS_CLIP) && (f & EF_CLIPPED) && !(f & EF_SKIPCLIP);
        if (clipped || !(f & EF_FULL)) blk = i + 1;
        else if (!(f & EF_IS_CLIP) && (f & EF_SOLID) && (f & EF_OVER) && (f & EF_OPAQUE)) top = i + 1;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { top = max(top, (uint32_t)__shfl_xor(top, d, 64)); blk = max(blk, (uint32_t)__shfl_xor(blk, d, 64)); }
    if (lane == 0) { s_wcnt[wv] = top; s_wblk[wv] = blk; }
    __syncthreads();
    top = max(max(s_wcnt[0], s_wcnt[1]), max(s_wcnt[2], s_wcnt[3]));
    blk = max(max(s_wblk[0], s_wblk[1]), max(s_wblk[2], s_wblk[3]));
    const uint32_t skipped = top ? top - 1u : 0u;
    const int first = top ? (blk > top ? 2 : 1) : (blk ? 2 : 0);
    if (cache.tiles && first == 1 && !layers_were_removed) {            // all visible layers unchanged: nothing to draw
        int vis = 1;
        for (uint32_t i = skipped + tid; i < ne; i += 256) if ((flags[i] & EF_MASK) && !((uint32_t)keys[i] & REF_UNCH)) vis = 0;
        if (__syncthreads_and(vis)) {
            if (tid == 0) cache.tiles[tile] = make_uint2(ct_tags | (ne << 8), ct_solid);
            return;
        }
    }
    if (tid == 0) s_solid = 0;
    if (first != 2) {                                                   // fold: every layer from `skipped` up is a full cover
        Col dst = clear; bool ok = true;
        for (uint32_t k0 = skipped; k0 < ne; k0 += PBATCH) {
            const uint32_t nbt = min((uint32_t)PBATCH, ne - k0);
            __syncthreads();
            if ((uint32_t)tid < nbt) {
                b_col[tid] = layer_col[(uint32_t)(keys[k0 + tid] >> 32) & LAYER_MASK];
            }
            __syncthreads();
            if (tid == 0) {
                for (uint32_t t = 0; t < nbt && ok; t++) {
                    const uint32_t f = flags[k0 + t];
                    if (!(f & EF_MASK)) continue;
                    const uint4 c4 = b_col[t];
                    const Col src = {__uint_as_float(c4.x), __uint_as_float(c4.y), __uint_as_float(c4.z), __uint_as_float(c4.w)};
                    if (first == 1 && k0 + t == skipped) { dst = src; continue; }       // the opaque cover is the bottom colour
                    if (!(f & EF_IS_CLIP) && (f & EF_SOLID)) dst = sc_blend((uint32_t)(keys[k0 + t] >> (53 + SF_BLEND_SHIFT)) & 15u, dst, src);
                    else ok = false;
                }
            }
        }
        if (tid == 0 && ok) {                                           // to_srgb_bytes(channels.map(color.channel)) :156-162, 690
            float sel[4];
#pragma unroll
            for (int c = 0; c < 4; c++) sel[c] = sel_channel((P.channels >> (8 * c)) & 0xFFu, dst.r, dst.g, dst.b, dst.a);
            s_solid_bytes = to_u8_x4(linear_to_srgb(sel[0])) | (to_u8_x4(linear_to_srgb(sel[1])) << 8) |
                            (to_u8_x4(linear_to_srgb(sel[2])) << 16) | (to_u8_x4(sel[3]) << 24);
            s_solid = 1;
        }
    }
    __syncthreads();

    const uint32_t px = tx * 16u + (uint32_t)lx, py = ty * 16u + (uint32_t)ly;
    const bool in_image = px < P.width && py < P.height;
    uint32_t* out_px = (uint32_t*)image + (size_t)py * P.stride_px + px;
    if (s_solid) {
        const uint32_t bytes = s_solid_bytes;
        if (cache.tiles) {                                              // CachedTile::convert_optimizer_op :690-707
            const bool same = (ct_tags & 1u) && ct_solid == bytes;
            if (tid == 0) {
                cache.tiles[tile] = make_uint2(ct_tags | 1u | (ne << 8), bytes);
                if (!same) cache.written[tile] = 1;
            }
            if (same) return;
        }
        if (in_image) *out_px = bytes;
        return;
    }

    // ---- the entries that are actually painted, in layer order (reuses e_tmp) ----------------------------------
    uint32_t* p_idx = (uint32_t*)e_tmp;
    uint32_t np = 0;
    for (uint32_t c = 0; c < ne; c += 256) {
        const uint32_t i = c + tid;
        const bool keep = i >= skipped && i < ne && (flags[i] & EF_MASK);
        const uint64_t bal = __ballot(keep);
        if (lane == 0) s_wcnt[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t base = np, tot = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t t = s_wcnt[q]; if (q < wv) base += t; tot += t; }
        if (keep) p_idx[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = i;
        np += tot;
        __syncthreads();
    }
    cells[0][tid] = 0; cells[1][tid] = 0;

    // ---- paint (Painter::paint_layer, painter/mod.rs:290-347, one lane per pixel) --------------------------
    float dr = clear.r, dg = clear.g, db = clear.b, da = clear.a;          // Painter::clear :277-288
    bool clip_valid = false; uint32_t clip_last = 0; float clip_mask = 0.0f;
    const float fx = (float)px;                                             // x - 1 + tile_x * TILE_WIDTH  (:325)
    const float fybase = (float)(ty * 16u + ((uint32_t)ly & 8u));          // y * LANES + tile_y * TILE_HEIGHT (:326)
    const int jy = ly & 7;
    uint32_t cbuf = 0;
    for (uint32_t b0 = 0; b0 < np; b0 += PBATCH) {
        const uint32_t nbt = min((uint32_t)PBATCH, np - b0);
        __syncthreads();                                                    // previous batch fully consumed; t_seg/cells visible
        if ((uint32_t)tid < nbt) {                                          // everything the layer loop needs, into LDS
            const uint32_t i = p_idx[b0 + tid];
            const uint64_t k = keys[i];
            const uint32_t ref = (uint32_t)k;
            b_flag[tid] = flags[i] | ((uint32_t)(k >> 53) << 16);           // EF_* low, SF_* (blend / fill) high
            b_layer[tid] = (uint32_t)(k >> 32) & LAYER_MASK;
            b_col[tid] = layer_col[(uint32_t)(k >> 32) & LAYER_MASK];
            if (ref & 0x80000000u) {
                b_cov[tid] = span_cov[ref & REF_IDX];
                b_seg0[tid] = 0; b_nseg[tid] = 0;
            } else {
                const TileRecord* r = &records[ref & REF_IDX];
                b_cov[tid] = make_uint4(r->cover[0], r->cover[1], r->cover[2], r->cover[3]);
                b_seg0[tid] = r->seg_start; b_nseg[tid] = r->seg_count;
            }
        }
        __syncthreads();
        for (uint32_t t = 0; t < nbt; t++) {
            const uint32_t f = b_flag[t];
            const uint32_t layer = b_layer[t];
            const uint32_t sfl = f >> 16;
            const uint4 cv4 = b_cov[t];
            const uint32_t cw = (ly >> 2) == 0 ? cv4.x : ((ly >> 2) == 1 ? cv4.y : ((ly >> 2) == 2 ? cv4.z : cv4.w));
            const int carry = (int)(int8_t)(cw >> ((ly & 3) * 8));
            int A;
            const uint32_t nseg = b_nseg[t];
            if (nseg) {
                int* cb = cells[cbuf];
                const uint32_t off = b_seg0[t] - seg0;                      // position inside the tile's segment range
                for (uint32_t sidx = tid; sidx < nseg; sidx += 256) {       // acc_segment :257-271
                    const uint32_t o = off + sidx;
                    const uint64_t v = o < TSEG_CAP ? t_seg[o] : sorted[seg0 + o];
                    const int cv = seg_cover(v);
                    atomicAdd(&cb[seg_ly(v) * 16 + seg_lx(v)], (int)((uint32_t)(seg_dam(v) * cv) << 16) + cv);
                }
                __syncthreads();
                const int S = cb[tid];
                cb[tid] = 0;                                                // ready for the layer after next
                cbuf ^= 1u;
                const int c = (int)(int16_t)(S & 0xFFFF);
                const int area = (int)(int16_t)((uint32_t)(S - c) >> 16);
                int inc = c;                                                // signed cover prefix along x (one DPP row)
                inc += dpp_row_shr(inc, 1); inc += dpp_row_shr(inc, 2); inc += dpp_row_shr(inc, 4); inc += dpp_row_shr(inc, 8);
                const int acc = (int)(int8_t)(carry + (inc - c));           // i8 wrapping column accumulator :343-345
                A = 32 * acc + area;                                        // compute_doubled_areas :388-404
            } else {
                A = 32 * carry;
            }
            if (clip_valid && clip_last < layer) clip_valid = false;        // :298-302
            const float cov = coverage_of(A, (f & EF_EVENODD) != 0);
            if (f & EF_IS_CLIP) {                                           // clip_at :449-464
                if (!clip_valid) { clip_valid = true; clip_last = layer + b_col[t].x; }
                clip_mask = cov;
                continue;
            }
            const bool apply_clip = (f & EF_CLIPPED) && !(f & EF_SKIPCLIP);
            if (cov == 0.0f) continue;                                      // :317-319 (per pixel: blend with 0 is the identity)
            if (apply_clip && !clip_valid) continue;                        // :321-323
            float fill[4];
            const uint32_t ft = (sfl >> SF_FILL_SHIFT) & 3u;
            if (ft == FORMA_FILL_SOLID) {
                const uint4 col = b_col[t];
                fill[0] = __uint_as_float(col.x); fill[1] = __uint_as_float(col.y); fill[2] = __uint_as_float(col.z); fill[3] = __uint_as_float(col.w);
            } else {                                                        // gradients / textures read their full style words
                const uint32_t* w = style_words + style_offsets[layer];
                if (ft == FORMA_FILL_TEXTURE) texture_at(w, images, texels, fx, fybase + (float)jy, fill);
                else gradient_at(w, ft, FORMA_STYLE_STOPS(w[0]), fx, fybase, jy, fill);
            }
            float src_a = fill[3] * cov;                                    // blend_at :406-447
            if (apply_clip) src_a *= clip_mask;
            float bl[3];
            blend_rgb((sfl >> SF_BLEND_SHIFT) & 15u, dr, dg, db, fill[0], fill[1], fill[2], bl);
            float ida = 1.0f - da, k1 = ida * src_a, isa = 1.0f - src_a, k2 = da * src_a;
            float cr = fmaf(fill[0], k1, bl[0] * k2);
            float cg = fmaf(fill[1], k1, bl[1] * k2);
            float cb2 = fmaf(fill[2], k1, bl[2] * k2);
            dr = fmaf(dr, isa, cr); dg = fmaf(dg, isa, cg); db = fmaf(db, isa, cb2);
            da = fmaf(da, isa, src_a);
        }
    }
    // ---- compute_srgb :466-483 + channel select, straight to the row-major RGBA8 image ----------------------
    if (in_image) {
        float sr = linear_to_srgb(dr), sg = linear_to_srgb(dg), sb = linear_to_srgb(db);
        uint32_t out = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) out |= to_u8_x8(sel_channel((P.channels >> (8 * c)) & 0xFFu, sr, sg, sb, da)) << (8 * c);
        *out_px = out;
    }
    if (cache.tiles && tid == 0) {                                      // update_solid_color(None): painted, not solid
        cache.tiles[tile] = make_uint2((ct_tags & 2u) | (ne << 8), ct_solid);
        cache.written[tile] = 1;
    }
}

// ================================================================================================
// the common-case painter: ONE WAVEFRONT PER TILE.  64 lanes own the 256 pixels (lane = local_x +
// 16 * row-group, four rows per lane), so nothing in the kernel needs a workgroup barrier: the tile's
// layer list, the coverage cells and the per-layer staging live in the wave's private slice of LDS and
// LDS operations of one wave retire in order.  A CU then holds ~28 independent tiles instead of 8, which
// is what this latency-bound stage needs (a tile is ~4 dependent memory round trips and ~6 painted
// layers).  Lists deeper than WMAX go to the workgroup-per-tile variant above (k_paint_deep).
// ================================================================================================
#ifdef PAINT_PROF
// -DPAINT_PROF (tools only): shader-clock stamps at the phase boundaries of k_paint_wave, summed over all tiles
__device__ unsigned long long g_paint_prof[256][24];       // 256 copies by workgroup: same-address atomics would dominate the stamps
#define PP_STAMP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_paint_prof[blockIdx.x & 255][i], _t - pp_t); pp_t = __builtin_readcyclecounter(); } while (0)
#define PP_COUNT(i, v) do { if (lane == 0) atomicAdd(&g_paint_prof[blockIdx.x & 255][i], (unsigned long long)(v)); } while (0)
extern "C" int forma_hip_debug_paint_prof(unsigned long long* out24, int reset) {
    static unsigned long long h[256][24];
    if (reset) { for (int c = 0; c < 256; c++) for (int i = 0; i < 24; i++) h[c][i] = 0; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_paint_prof), h, sizeof h); }
    int rc = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_paint_prof), sizeof h);
    for (int i = 0; i < 24; i++) { out24[i] = 0; for (int c = 0; c < 256; c++) out24[i] += h[c][i]; }
    return rc;
}
#define PP_STAMP_IN(i) PP_STAMP(i)
#else
#define PP_STAMP_IN(i) do { } while (0)
#define PP_STAMP(i) do { } while (0)
#define PP_COUNT(i, v) do { } while (0)
#endif
#define WMAX 128          // layer-list capacity of the wave painter
#ifndef PAINT_GENERIC_OCC
#define PAINT_GENERIC_OCC 6  // ... and the general one
#endif
#ifndef PAINT_SIMPLE_OCC
#define PAINT_SIMPLE_OCC 8   // waves per SIMD the all-solid variant is compiled for
#endif
#ifndef PAINT_STRIP_OCC
#define PAINT_STRIP_OCC 8    // ... and the strip variants (one pixel per lane: a quarter of the colour registers)
#endif
#define WB   16           // painted entries staged per batch

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d, 64));
    return v;
}

// SIMPLE: every layer of the scene is a solid colour blended with BlendMode::Over and nothing is a clip or clipped (the host
// knows from the style table, forma_hip_set_styles) — the fills, the sixteen blend modes and the clip state machine are
// not even compiled in, which is worth registers (occupancy), instruction cache and the per-layer dispatch.
// ONE_SLICE: the carry pre-pass ran one workgroup per tile row (P.n_slices == 1): SpanListsT<1>.
// NPX: pixels per lane.  4 = one wavefront per tile (lane = column lx of row group rg, pixel rows 4 rg .. 4 rg + 3).
// 1 = STRIPS: four independent wavefronts per tile, each owning the 16 x 4 strip of pixel rows 4 s .. 4 s + 3 (lane = one
// pixel, the reference GPU backend's tile shape, consts.rs:45-48).  Every strip builds the tile's layer list itself — no
// workgroup barrier, no traffic between the strips, the x-prefix stays inside 16-lane DPP rows — accumulates only the
// segments of its rows and takes its four bytes of the carry cover.  The pixel work of a tile (segments, coverage, fills,
// blends, encode) is then spread over four wavefronts with a quarter of the colour registers each: a launch's floor — its
// deepest tile walked by ONE wavefront — drops accordingly, which is what small frames (1080p, a multi-GPU band) are bound
// by.  The list work is done four times, so frames that fill the chip with one wavefront per tile keep NPX = 4.
template <bool SIMPLE, bool ONE_SLICE, int NPX>
__global__ __launch_bounds__(64, NPX == 1 ? PAINT_STRIP_OCC : (SIMPLE ? PAINT_SIMPLE_OCC : PAINT_GENERIC_OCC)) void k_paint_wave(PaintParams P, const uint64_t* __restrict__ sorted,
                                                    const TileRecord* __restrict__ records, DevCount nc_runs,
                                                    const uint32_t* __restrict__ tile_first_run,
                                                    const uint32_t* __restrict__ row_span_lo,
                                                    const uint32_t* __restrict__ row_span_cnt,
                                                    const uint64_t* __restrict__ span_key,
                                                    const uint4* __restrict__ span_cov, const uint4* __restrict__ layer_col,
                                                    const uint32_t* __restrict__ style_offsets,
                                                    const uint32_t* __restrict__ style_words,
                                                    const forma_image_t* __restrict__ images,
                                                    const uint16_t* __restrict__ texels, uint8_t* __restrict__ image,
                                                    TileCacheArgs cache, FrameInfo* __restrict__ info,
                                                    uint32_t* __restrict__ overflow_n,
                                                    uint32_t* __restrict__ overflow_list, uint32_t deep_follows,
                                                    SpanGroups groups) {
    __shared__ uint64_t w_key[1][WMAX];
    __shared__ uint64_t w_tmp[1][WMAX];
    __shared__ uint16_t w_flag[1][WMAX];                                // (EF_* fit 10 bits; 5 120 B per wave = 32 waves per CU)
    __shared__ int w_cells[1][2][64 * NPX];
    __shared__ uint4 w_cov[1][WB], w_col[1][WB];
    __shared__ uint32_t w_seg0[1][WB], w_nseg[1][WB], w_bflag[1][WB], w_blayer[1][WB];
    __shared__ uint32_t w_woff[1][SIMPLE ? 1 : WB];                     // non-solid entries of the batch: where their style words are
    __shared__ uint32_t w_style[1][SIMPLE ? 1 : WSTYLE];                // the current non-solid layer's style words

    const int lane = threadIdx.x & 63, wv = 0;
    // The guard word, the run count and the tile's three table entries are independent loads: issue all of them before
    // the first is tested (as written before, `plan_bad` was a global round trip of its own in front of everything).
    const uint32_t plan_bad = info->plan_bad;                           // mis-sorted stream (async frame): the host re-runs
    const uint32_t n_runs_frame = dev_count(nc_runs);
    // one-wave workgroups (a wave's slot frees as soon as ITS tile is done).  XCD-aware mapping: workgroup b runs on
    // XCD b % 8; give each XCD a contiguous band of tiles so a tile row's records / spans stay in one L2
    // Only the crop's tile rows are launched (a multi-GPU rank paints its band only).
    const uint32_t tile0 = P.crop_y0 * P.tiles_w, T = (P.crop_y1 - P.crop_y0) * P.tiles_w;
    const uint32_t per = PAINT_ROW_XCD ? ((P.crop_y1 - P.crop_y0 + 7u) / 8u) * P.tiles_w : (T + 7u) / 8u;
    const uint32_t bid = blockIdx.x;
    // (strips: the four wavefronts of a tile are consecutive workgroups of ONE XCD — they read the same records and segments)
    const uint32_t kx = NPX == 1 ? (bid >> 5) : (bid >> 3);
    const uint32_t strip = NPX == 1 ? ((bid >> 3) & 3u) : 0u;
    // which tile of the band: the heavy section in front of the grid takes the previous frame's heavy tiles, the main section
    // every tile in index order that is not flagged (PaintParams::order_*)
    const bool ordered = NPX == 4 && P.order_cnt_out != nullptr;
    const uint32_t hcap = ordered ? P.order_hcap : 0u;
    if (kx >= per + hcap) return;
#if PAINT_ROW_XCD
    const uint32_t band_rows = (P.crop_y1 - P.crop_y0) > (bid & 7u) ? ((P.crop_y1 - P.crop_y0) - (bid & 7u) + 7u) / 8u : 0u;
    const uint32_t band0 = (bid & 7u) * per, band_n = band_rows * P.tiles_w;
#else
    const uint32_t band0 = (bid & 7u) * per, band_n = band0 < T ? min(per, T - band0) : 0u;
#endif
    uint32_t tin;
    unsigned long long ord_t0 = 0;
    if (ordered) {
        ord_t0 = __builtin_readcyclecounter();
        if (kx < hcap) {                                                // list kx % SUBS of the band, entry kx / SUBS (hcap = SUBS x list capacity)
            const uint32_t sub = kx % PAINT_ORDER_SUBS, idx = kx / PAINT_ORDER_SUBS, cap = hcap / PAINT_ORDER_SUBS;
            if (!P.order_cnt_in || idx >= min(P.order_cnt_in[(bid & 7u) * PAINT_ORDER_SUBS + sub], cap)) return;
            tin = P.order_list_in[((size_t)(bid & 7u) * PAINT_ORDER_SUBS + sub) * cap + idx];
            if (tin >= band_n) return;                                  // (never: a list holds tiles of its own band)
        } else {
            tin = kx - hcap;
            if (tin >= band_n) return;
            if (P.order_cnt_in && P.order_flag_in[band0 + tin]) return; // painted by the heavy section
        }
    } else {
        tin = kx;
        if (tin >= band_n) return;
    }
    // a wavefront that is done with its tile says whether the next frame should start it early
    auto tile_done = [&]() {
        if (ordered && (threadIdx.x & 63) == 0) {
            const unsigned long long dt = __builtin_readcyclecounter() - ord_t0;
            // (one tile in 256 tells the host what a tile costs on average: the threshold never goes below twice that)
            if (((band0 + tin) & 255u) == 0u) { atomicAdd(&info->cost_sum, (uint32_t)min(dt >> 8, 0xFFFFull)); atomicAdd(&info->cost_n, 1u); }
            uint8_t heavy = 0;
            if (dt >= (unsigned long long)P.order_thr) {
                const uint32_t slot = (bid & 7u) * PAINT_ORDER_SUBS + (tin % PAINT_ORDER_SUBS), cap = hcap / PAINT_ORDER_SUBS;
                const uint32_t pos = atomicAdd(&P.order_cnt_out[slot], 1u);
                if (pos < cap) { P.order_list_out[(size_t)slot * cap + pos] = tin; heavy = 1; }
            }
            P.order_flag_out[band0 + tin] = heavy;
        }
    };
#if PAINT_ROW_XCD
    const uint32_t tidx = ((bid & 7u) + 8u * (tin / P.tiles_w)) * P.tiles_w + tin % P.tiles_w;
#else
    const uint32_t tidx = band0 + tin;
#endif
    const uint32_t tile = tile0 + tidx;
    const uint32_t ty = tile / P.tiles_w, tx = tile - ty * P.tiles_w;
    if (tx < P.crop_x0 || tx >= P.crop_x1 || ty < P.crop_y0 || ty >= P.crop_y1) { tile_done(); return; }   // print_row :588-592, :525-529

    uint64_t* keys = w_key[wv];
    uint64_t* tmp = w_tmp[wv];
    uint16_t* flags = w_flag[wv];
    const int lx = lane & 15, rg = lane >> 4;                           // this lane's pixels: (lx, 4 * rg + q), q = 0..3; strips: (lx, 4 * strip + rg)
    const int row0 = NPX == 1 ? (int)strip * 4 + rg : rg * 4;           // first (strips: the only) pixel row of the lane
    const bool lead = NPX == 4 || strip == 0u;                          // the wavefront that speaks for the tile (overflow list, counters)
#ifdef PAINT_PROF
    unsigned long long pp_t = __builtin_readcyclecounter();
    PP_COUNT(16, 1);
#endif

    // ---- the tile's layer list: own runs (contiguous records, ascending layer) + the row's spans that cross it --------
    const uint32_t my_tile_key = ((ty + 1u) << 12) | (tx + 1u);
    const uint32_t n_runs = PAINT_RUN_END(P, ty, n_runs_frame);
    const uint32_t j0 = tile_first_run[tile] - 1u;                      // 0 stored = no run -> FORMA_NONE
    // the spans that may cross this tile: its tile-column group's lists (one per slice; SpanGroups in common.h), or — no group
    // lists this frame, or the pool was full for this row — the row's
    constexpr int NS = ONE_SLICE ? 1 : CR_MAX_SLICES;
    SpanListsT<NS> SL = load_span_lists_t<NS>(row_span_lo, row_span_cnt, ty, P.n_slices);      // (both tables requested at once)
    bool by_group = groups.tab != nullptr;
    if (by_group) {
        const SpanListsT<NS> GL = load_group_lists<NS>(groups.tab + (size_t)ty * P.n_slices * P.n_groups + (tx >> SPAN_GROUP_SHIFT), P.n_groups, P.n_slices);
        if (GL.total == SPAN_GROUP_NONE) by_group = false;              // (uniform)
        else SL = GL;
    }
    const uint32_t sc = SL.total;
    if (plan_bad) return;
    // a candidate span as three words, whichever list it comes from: the entry (high word = layer | style bits, low word =
    // reference) and lo | hi << 16
    uint32_t eh[4], el[4], lh[4];
    auto fetch = [&](uint32_t i, uint32_t& h, uint32_t& l, uint32_t& x) {
        h = 0; l = 0; x = 0;                                            // padding: lo = hi = 0, crosses nothing
        if (i < sc) {
            const uint32_t ph = span_phys(SL, i);
            if (by_group) { const uint4 g = groups.list[ph]; h = g.x; l = g.y; x = g.z; }
            else { const uint64_t k = span_key[ph]; h = (uint32_t)(k >> 32); x = (uint32_t)k; l = REF_SPAN | ((x >> 31) ? REF_UNCH : 0u) | ph; }
        }
    };
#pragma unroll
    for (int u = 0; u < 4; u++) fetch(u * 64 + lane, eh[u], el[u], lh[u]);
    // Occlusion culling (PaintParams::cull): occ = 1 + the layer of the topmost OCCLUDER among the spans that cross this tile.
    // Everything below it is what skip_fully_covered_layers_pass skips anyway; dropped here, while the list is built, the merge
    // and the passes work on a fraction of the entries and lists beyond WMAX become rare.
    uint32_t occ = 0;
    if (P.cull) {
        for (uint32_t c = 0; c < sc; c += 256) {
            if (c) {
#pragma unroll
                for (int u = 0; u < 4; u++) fetch(c + u * 64 + lane, eh[u], el[u], lh[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {                               // (the lists ascend in layer: the LAST occluder found is the topmost)
                const uint32_t lo = (lh[u] >> 16) & 0x7FFFu, hi = lh[u] & 0xFFFFu;
                const uint64_t ob = __ballot(tx >= lo && tx < hi && span_is_occluder(eh[u]));
                if (ob) occ = ((uint32_t)__builtin_amdgcn_readlane((int)eh[u], 63 - __builtin_clzll(ob)) & LAYER_MASK) + 1u;
            }
        }
        if (sc > 256u) {                                                // (rare: the row's list in several rounds — back to the first)
#pragma unroll
            for (int u = 0; u < 4; u++) fetch(u * 64 + lane, eh[u], el[u], lh[u]);
        }
    }
    uint32_t na = 0;
    if (j0 != FORMA_NONE) {
        for (uint32_t c = 0;; c += 64) {                                // a tile's runs are contiguous from j0
            const uint32_t j = j0 + c + lane;
            bool mine = false; uint32_t layer = 0, unch = 0;
            if (j < n_runs) { const TileRecord* r = &records[j]; mine = (r->tile & 0x7FFFFFFFu) == my_tile_key; layer = r->layer; unch = (r->tile >> 31) ? REF_UNCH : 0u; }
            const bool keep = mine && (layer & LAYER_MASK) + 1u >= occ; // (ascending layers: the kept runs are a suffix of the tile's)
            const uint64_t kb = __ballot(keep);
            const uint32_t pos = na + __builtin_amdgcn_mbcnt_hi((uint32_t)(kb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kb, 0u));
            if (keep && pos < WMAX) tmp[pos] = ((uint64_t)layer << 32) | unch | j;
            const uint32_t got = (uint32_t)__popcll(__ballot(mine));
            na += (uint32_t)__popcll(kb);
            if (got < 64u) break;
        }
    }
    uint32_t nb = 0;
    for (uint32_t c = 0; c < sc; c += 256) {
        if (c) {
#pragma unroll
            for (int u = 0; u < 4; u++) fetch(c + u * 64 + lane, eh[u], el[u], lh[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t lo = (lh[u] >> 16) & 0x7FFFu, hi = lh[u] & 0xFFFFu;
            const bool hit = tx >= lo && tx < hi && (eh[u] & LAYER_MASK) + 1u >= occ;
            const uint64_t bal = __ballot(hit);
            if (hit) {
                const uint32_t pos = na + nb + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (pos < WMAX) tmp[pos] = ((uint64_t)eh[u] << 32) | el[u];
            }
            nb += (uint32_t)__popcll(bal);
        }
    }
    const uint32_t ne = na + nb;
    PP_STAMP(0);                                                        // 0: tile's runs + crossing spans found
    PP_COUNT(17, ne); PP_COUNT(18, sc);
    if (ne > WMAX) {                                                    // too deep for a wave: the workgroup variant paints it
        if (lane == 0 && lead) {
            overflow_list[atomicAdd(overflow_n, 1u)] = tile;
            atomicOr(&info->error, 16u);                                // (not an error: "this frame has deep tiles", read by the host)
            if (!deep_follows) info->plan_bad = 1u;                     // the host guessed "none" and did not launch k_paint_deep: re-run
        }
        tile_done();
        return;
    }
    wave_lds_sync();
    // merge by layer (a (tile, layer) pair is either a run or a span: layers are unique across the two lists)
    for (uint32_t i = lane; i < ne; i += 64) {
        const uint64_t k = tmp[i];
        const uint32_t layer = (uint32_t)(k >> 32) & LAYER_MASK;
        uint32_t lo, hi;
        if (i < na) { lo = na; hi = ne; } else { lo = 0; hi = na; }
        const uint32_t other0 = lo;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (((uint32_t)(tmp[mid] >> 32) & LAYER_MASK) < layer) lo = mid + 1; else hi = mid; }
        const uint32_t rank = (i < na ? i : i - na) + (lo - other0);
        keys[rank] = k;
        // per-entry facts for the optimizer passes, decoded from the SF_* bits of the key
        const uint32_t sfl = (uint32_t)(k >> 53), ref = (uint32_t)k;
        uint32_t f = EF_MASK;
        if (sfl & SF_EVENODD) f |= EF_EVENODD;
        if (!(ref & 0x80000000u)) f |= EF_HAS_SEGS;
        else if (sfl & SF_FULL) f |= EF_FULL;
        if (sfl & SF_IS_CLIP) f |= EF_IS_CLIP;
        else {
            if (sfl & SF_CLIPPED) f |= EF_CLIPPED;
            if (((sfl >> SF_FILL_SHIFT) & 3u) == FORMA_FILL_SOLID) f |= EF_SOLID;
            if (sfl & SF_OPAQUE) f |= EF_OPAQUE;
            if (((sfl >> SF_BLEND_SHIFT) & 15u) == 0u) f |= EF_OVER;
        }
        flags[rank] = (uint16_t)f;
    }
    wave_lds_sync();

    PP_STAMP(1);                                                        // 1: merged by layer, flags decoded
    const Col clear = {P.clear[0], P.clear[1], P.clear[2], P.clear[3]};
    // ---- buffer-layer cache: tile_unchanged_pass (passes/tile_unchanged.rs), the first pass ------------------------------
    bool layers_were_removed = true;                                    // PassesSharedState default
    uint32_t ct_tags = 0, ct_solid = 0;
    if (cache.tiles) {
        const uint2 ct = cache.tiles[tile];
        ct_solid = ct.y;
        int all_unch = 1;
        for (uint32_t i = lane; i < ne; i += 64) if (!((uint32_t)keys[i] & REF_UNCH)) all_unch = 0;
        all_unch = __all(all_unch);
        const bool had = (ct.x & 2u) != 0;
        const uint32_t prev = ct.x >> 8;
        ct_tags = (ct.x & 1u) | 2u;                                     // update_layer_count(Some(layers))
        bool tile_unchanged = false;
        if (had) { layers_were_removed = ne < prev; tile_unchanged = prev == ne && all_unch; }
        if (P.clear_unchanged && tile_unchanged) {                      // TileWriteOp::None: the buffer keeps last frame's pixels
            if (lane == 0) cache.tiles[tile] = make_uint2(ct_tags | (ne << 8), ct_solid);
            tile_done();
            return;
        }
    }
    // ---- optimizer passes (layer_workbench/passes/*.rs) -------------------------------------------------------------
    if (!SIMPLE && P.scene_has_clips) {                                 // skip_trivial_clips_pass: serial (clip state machine)
        if (lane == 0) {
            bool has = false, c_full = false, c_used = false; uint32_t c_last = 0, c_i = 0;
            for (uint32_t i = 0; i < ne; i++) {
                uint32_t f = flags[i];
                if (!(f & EF_MASK)) continue;
                const uint32_t id = (uint32_t)(keys[i] >> 32) & LAYER_MASK;
                if (f & EF_IS_CLIP) {
                    c_full = (f & EF_FULL) != 0;
                    c_last = id + style_words[style_offsets[id] + 1]; c_i = i; c_used = false; has = true;
                    if (c_full) { f &= ~EF_MASK; flags[i] = f; }
                }
                if (!(f & EF_IS_CLIP) && (f & EF_CLIPPED)) {
                    if (has && id <= c_last) { if (c_full) { f |= EF_SKIPCLIP; flags[i] = f; } else c_used = true; }
                    else { f &= ~EF_MASK; flags[i] = f; }
                }
                if (has && id > c_last) { has = false; if (!c_used) flags[c_i] &= ~EF_MASK; }
            }
            if (has && !c_used) flags[c_i] &= ~EF_MASK;
        }
        wave_lds_sync();
    }
    // skip_fully_covered_layers_pass: topmost FULL unclipped solid Over opaque layer ("cover") vs. topmost blocker
    uint32_t top = 0, blk = 0;
    for (uint32_t i = lane; i < ne; i += 64) {
        const uint32_t f = flags[i];
        if (!(f & EF_MASK)) continue;
        const bool clipped = !SIMPLE && !(f & EF_IS_CLIP) && (f & EF_CLIPPED) && !(f & EF_SKIPCLIP);
        if (clipped || !(f & EF_FULL)) blk = i + 1;
        else if (SIMPLE ? (f & EF_OPAQUE) != 0 : (!(f & EF_IS_CLIP) && (f & EF_SOLID) && (f & EF_OVER) && (f & EF_OPAQUE))) top = i + 1;
    }
    top = wave_max_u32(top); blk = wave_max_u32(blk);
    const uint32_t skipped = top ? top - 1u : 0u;
    const int first = top ? (blk > top ? 2 : 1) : (blk ? 2 : 0);
    if (cache.tiles && first == 1 && !layers_were_removed) {            // all visible layers unchanged: nothing to draw
        int vis = 1;                                                    // (skip_fully_covered_layers.rs:41-47, 88-91)
        for (uint32_t i = skipped + lane; i < ne; i += 64) if ((flags[i] & EF_MASK) && !((uint32_t)keys[i] & REF_UNCH)) vis = 0;
        if (__all(vis)) {
            if (lane == 0) cache.tiles[tile] = make_uint2(ct_tags | (ne << 8), ct_solid);
            tile_done();
            return;
        }
    }

    uint4* b_cov = w_cov[wv]; uint4* b_col = w_col[wv];
    uint32_t* b_seg0 = w_seg0[wv]; uint32_t* b_nseg = w_nseg[wv]; uint32_t* b_flag = w_bflag[wv]; uint32_t* b_layer = w_blayer[wv];
    const uint32_t px = tx * 16u + (uint32_t)lx;
    PP_STAMP(2);                                                        // 2: optimizer passes (clips, topmost cover)
    if (first != 2) {                                                   // fold: every layer from `skipped` up is a full cover
        Col dst = clear; bool ok = true;
        for (uint32_t k0 = skipped; k0 < ne; k0 += WB) {
            const uint32_t nbt = min((uint32_t)WB, ne - k0);
            wave_lds_sync();
            if ((uint32_t)lane < nbt) {
                b_col[lane] = layer_col[(uint32_t)(keys[k0 + lane] >> 32) & LAYER_MASK];
            }
            wave_lds_sync();
            for (uint32_t t = 0; t < nbt && ok; t++) {                 // every lane folds the same (uniform) values
                const uint32_t f = flags[k0 + t];
                if (!(f & EF_MASK)) continue;
                const uint4 c4 = b_col[t];
                const Col src = {__uint_as_float(c4.x), __uint_as_float(c4.y), __uint_as_float(c4.z), __uint_as_float(c4.w)};
                if (first == 1 && k0 + t == skipped) { dst = src; continue; }
                if (SIMPLE) dst = sc_blend(0u, dst, src);
                else if (!(f & EF_IS_CLIP) && (f & EF_SOLID)) dst = sc_blend((uint32_t)(keys[k0 + t] >> (53 + SF_BLEND_SHIFT)) & 15u, dst, src);
                else ok = false;
            }
        }
        if (ok) {                                                       // TileWriteOp::Solid: to_srgb_bytes :156-162, 690
            float sel[4];
#pragma unroll
            for (int c = 0; c < 4; c++) sel[c] = sel_channel((P.channels >> (8 * c)) & 0xFFu, dst.r, dst.g, dst.b, dst.a);
            const uint32_t bytes = to_u8_x4(linear_to_srgb(sel[0])) | (to_u8_x4(linear_to_srgb(sel[1])) << 8) |
                                   (to_u8_x4(linear_to_srgb(sel[2])) << 16) | (to_u8_x4(sel[3]) << 24);
            if (cache.tiles) {                                          // CachedTile::convert_optimizer_op :690-707
                const bool same = (ct_tags & 1u) && ct_solid == bytes;
                if (lane == 0) {
                    cache.tiles[tile] = make_uint2(ct_tags | 1u | (ne << 8), bytes);
                    if (!same) cache.written[tile] = 1;
                }
                if (same) { tile_done(); return; }                      // same solid colour as last frame: TileWriteOp::None
            }
#pragma unroll
            for (int q = 0; q < NPX; q++) {
                const uint32_t py = ty * 16u + (uint32_t)(row0 + q);
                if (px < P.width && py < P.height) ((uint32_t*)image)[(size_t)py * P.stride_px + px] = bytes;
            }
            PP_STAMP(3); PP_COUNT(19, 1);                               // 3: solid fold + store (19: solid tiles)
            tile_done();
            return;
        }
        PP_STAMP(4);                                                    // 4: fold attempted, failed
    }

    // ---- the entries that are actually painted, in layer order (list reuses `tmp`) --------------------------------------
    wave_lds_sync();
    uint32_t* p_idx = (uint32_t*)tmp;
    uint32_t np = 0;
    for (uint32_t c = 0; c < ne; c += 64) {
        const uint32_t i = c + lane;
        const bool keep = i >= skipped && i < ne && (flags[i] & EF_MASK);
        const uint64_t bal = __ballot(keep);
        if (keep) p_idx[np + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = i;
        np += (uint32_t)__popcll(bal);
    }
#pragma unroll
    for (int q = 0; q < NPX; q++) { w_cells[wv][0][q * 64 + lane] = 0; w_cells[wv][1][q * 64 + lane] = 0; }

    // ---- paint (Painter::paint_layer, painter/mod.rs:290-347): four pixels per lane ------------------------------------
    PP_STAMP(5); PP_COUNT(20, np);                                      // 5: painted-entry list (20: painted entries)
    float dr[NPX], dg[NPX], db[NPX], da[NPX];
#pragma unroll
    for (int q = 0; q < NPX; q++) { dr[q] = clear.r; dg[q] = clear.g; db[q] = clear.b; da[q] = clear.a; }   // Painter::clear :277-288
    bool clip_valid = false; uint32_t clip_last = 0; float clip_mask[NPX];
#pragma unroll
    for (int q = 0; q < NPX; q++) clip_mask[q] = 0.0f;
    const float fx = (float)px;
    uint32_t cbuf = 0;
    for (uint32_t b0 = 0; b0 < np; b0 += WB) {
        const uint32_t nbt = min((uint32_t)WB, np - b0);
        wave_lds_sync();
        if ((uint32_t)lane < nbt) {                                     // everything the layer loop needs, into LDS
            const uint32_t i = p_idx[b0 + lane];
            const uint64_t k = keys[i];
            const uint32_t ref = (uint32_t)k;
            b_flag[lane] = flags[i] | ((uint32_t)(k >> 53) << 16);
            b_layer[lane] = (uint32_t)(k >> 32) & LAYER_MASK;
            b_col[lane] = layer_col[(uint32_t)(k >> 32) & LAYER_MASK];
            if (!SIMPLE && (((uint32_t)(k >> 53) >> SF_FILL_SHIFT) & 3u) != FORMA_FILL_SOLID && !(flags[i] & EF_IS_CLIP))
                w_woff[wv][lane] = style_offsets[(uint32_t)(k >> 32) & LAYER_MASK];
            if (ref & 0x80000000u) {
                b_cov[lane] = span_cov[ref & REF_IDX];
                b_seg0[lane] = 0; b_nseg[lane] = 0;
            } else {
                const TileRecord* r = &records[ref & REF_IDX];
                b_cov[lane] = make_uint4(r->cover[0], r->cover[1], r->cover[2], r->cover[3]);
                b_seg0[lane] = r->seg_start; b_nseg[lane] = r->seg_count;
            }
        }
        wave_lds_sync();
        PP_STAMP(6);                                                    // 6: batch staging (records / covers / colours gathered)
        for (uint32_t t = 0; t < nbt; t++) {
            const uint32_t f = b_flag[t];
            const uint32_t layer = b_layer[t];
            const uint32_t sfl = f >> 16;
            const uint4 cv4 = b_cov[t];
            const uint32_t cwi = NPX == 1 ? strip : (uint32_t)rg;       // the cover word of the lane's rows: bytes = rows 4 i .. 4 i + 3
            const uint32_t cww = cwi == 0 ? cv4.x : (cwi == 1 ? cv4.y : (cwi == 2 ? cv4.z : cv4.w));
            const uint32_t cw = NPX == 1 ? cww >> (8 * rg) : cww;       // (strips: the lane's row in byte 0)
            const uint32_t nseg = b_nseg[t];
            int A[NPX];
            if (nseg) {
                PP_COUNT(21, nseg);
                int* cb = w_cells[wv][cbuf];
                const uint64_t* sp = sorted + b_seg0[t];
                for (uint32_t sidx = lane; sidx < nseg; sidx += 64) {   // acc_segment :257-271
                    const uint64_t v = sp[sidx];
                    const int cv = seg_cover(v);
                    if (NPX == 4) atomicAdd(&cb[seg_ly(v) * 16 + seg_lx(v)], (int)((uint32_t)(seg_dam(v) * cv) << 16) + cv);
                    else if ((uint32_t)(seg_ly(v) >> 2) == strip)       // (a strip takes the segments of its four rows)
                        atomicAdd(&cb[(seg_ly(v) & 3) * 16 + seg_lx(v)], (int)((uint32_t)(seg_dam(v) * cv) << 16) + cv);
                }
                wave_lds_sync();
#pragma unroll
                for (int q = 0; q < NPX; q++) {
                    const int ci = NPX == 1 ? lane : (rg * 4 + q) * 16 + lx;
                    const int S = cb[ci];
                    cb[ci] = 0;                                         // ready for the layer after next
                    const int c = (int)(int16_t)(S & 0xFFFF);
                    const int area = (int)(int16_t)((uint32_t)(S - c) >> 16);
                    int inc = c;                                        // signed cover prefix along x (one DPP row)
                    inc += dpp_row_shr(inc, 1); inc += dpp_row_shr(inc, 2); inc += dpp_row_shr(inc, 4); inc += dpp_row_shr(inc, 8);
                    const int carry = (int)(int8_t)(cw >> (q * 8));
                    const int acc = (int)(int8_t)(carry + (inc - c));   // i8 wrapping column accumulator :343-345
                    A[q] = 32 * acc + area;                             // compute_doubled_areas :388-404
                }
                cbuf ^= 1u;
                PP_STAMP(7);                                            // 7: segment accumulation + cover prefix
            } else {
#pragma unroll
                for (int q = 0; q < NPX; q++) A[q] = 32 * (int)(int8_t)(cw >> (q * 8));
            }
            if (!SIMPLE && clip_valid && clip_last < layer) clip_valid = false;    // :298-302
            const bool eo = (f & EF_EVENODD) != 0;
            if (!SIMPLE && (f & EF_IS_CLIP)) {                          // clip_at :449-464
                if (!clip_valid) { clip_valid = true; clip_last = layer + b_col[t].x; }
#pragma unroll
                for (int q = 0; q < NPX; q++) clip_mask[q] = coverage_of(A[q], eo);
                continue;
            }
            const bool apply_clip = !SIMPLE && (f & EF_CLIPPED) && !(f & EF_SKIPCLIP);
            if (apply_clip && !clip_valid) continue;                    // :321-323
            const uint32_t ft = SIMPLE ? (uint32_t)FORMA_FILL_SOLID : (sfl >> SF_FILL_SHIFT) & 3u;
            const uint32_t bm = SIMPLE ? 0u : (sfl >> SF_BLEND_SHIFT) & 15u;
            const uint4 col = b_col[t];
            if (SIMPLE || (ft == FORMA_FILL_SOLID && bm == 0u && !apply_clip)) {
                // the common layer — solid colour, BlendMode::Over, not clipped — as straight-line code: the generic loop
                // below dispatches on fill type and blend mode once per PIXEL (it is unrolled over the four pixels of a
                // lane), ~4x the instructions.  Same operations in the same order, so the same bits.
                const float fr = __uint_as_float(col.x), fg = __uint_as_float(col.y), fb = __uint_as_float(col.z), fa = __uint_as_float(col.w);
#pragma unroll
                for (int q = 0; q < NPX; q++) {
                    const float cov = coverage_of(A[q], eo);
                    const float src_a = fa * cov;                       // blend_at :406-447 with blend = Over (the source colour)
                    const float ida = 1.0f - da[q], k1 = ida * src_a, isa = 1.0f - src_a, k2 = da[q] * src_a;
                    const float nr = fmaf(dr[q], isa, fmaf(fr, k1, fr * k2));
                    const float ng = fmaf(dg[q], isa, fmaf(fg, k1, fg * k2));
                    const float nb2 = fmaf(db[q], isa, fmaf(fb, k1, fb * k2));
                    const float na2 = fmaf(da[q], isa, src_a);
                    const bool skip = cov == 0.0f;                      // :317-319
                    dr[q] = skip ? dr[q] : nr; dg[q] = skip ? dg[q] : ng; db[q] = skip ? db[q] : nb2; da[q] = skip ? da[q] : na2;
                }
                PP_STAMP_IN(8); PP_COUNT(23, 1);                        // 8: coverage + blend of a solid / Over layer (23: how many)
                continue;
            }
            // A gradient or a texture: its style words (geometry, stops, image transform) come into LDS with ONE round of loads,
            // all lanes at once.  Read where they are used — inside the per-pixel branches — every word was a global round trip
            // of its own (the loads cannot be hoisted out of a divergent branch): ~20 dependent L2 hits per gradient layer,
            // 18 k clocks against 2 k for a solid one.
            const uint32_t woff = (ft == FORMA_FILL_SOLID) ? 0u : w_woff[wv][t];
            const uint32_t* w = style_words + woff;
            uint32_t* ws = w_style[wv];
            forma_image_t tex_im = {0, 0, 0};
            if (ft != FORMA_FILL_SOLID) {
                const uint32_t v = woff + (uint32_t)lane < P.n_words ? w[lane] : 0u;
                wave_lds_sync();                                        // (the previous layer's readers are done)
                ws[lane] = v;
                wave_lds_sync();
                if (ft == FORMA_FILL_TEXTURE) tex_im = images[ws[8]];
            }
            // The uncommon layer (gradient, texture, one of the fifteen other blend modes, clipped): ONE pixel row of the lane at
            // a time in a loop that is not unrolled, the lane's four pixels rotating through slot 0 — unrolled, the compiler
            // interleaves four fills and four blends.  All lanes evaluate (a pixel with coverage 0 keeps its colour, :317-319:
            // blend with 0 is the identity), so nothing below sits in a divergent branch.
#pragma unroll 1
            for (int q = 0; q < NPX; q++) {
                const float cov = coverage_of(A[0], eo);
                if (__any(cov != 0.0f)) {
                    float fill[4];
                    if (ft == FORMA_FILL_SOLID) {
                        fill[0] = __uint_as_float(col.x); fill[1] = __uint_as_float(col.y); fill[2] = __uint_as_float(col.z); fill[3] = __uint_as_float(col.w);
                    } else {
                        const int ly = row0 + q;
                        const float fybase = (float)(ty * 16u + ((uint32_t)ly & 8u));                     // :326
                        if (ft == FORMA_FILL_TEXTURE) texture_at_im(ws, tex_im, texels, fx, fybase + (float)(ly & 7), fill);
                        else gradient_at_lds(ws, w, ft, FORMA_STYLE_STOPS(ws[0]), fx, fybase, ly & 7, lane, fill);
                    }
                    float src_a = fill[3] * cov;                        // blend_at :406-447
                    if (apply_clip) src_a *= clip_mask[0];
                    float bl[3];
                    blend_rgb(bm, dr[0], dg[0], db[0], fill[0], fill[1], fill[2], bl);
                    const float ida = 1.0f - da[0], k1 = ida * src_a, isa = 1.0f - src_a, k2 = da[0] * src_a;
                    const float cr = fmaf(fill[0], k1, bl[0] * k2);
                    const float cg = fmaf(fill[1], k1, bl[1] * k2);
                    const float cb2 = fmaf(fill[2], k1, bl[2] * k2);
                    const bool skip = cov == 0.0f;
                    dr[0] = skip ? dr[0] : fmaf(dr[0], isa, cr); dg[0] = skip ? dg[0] : fmaf(dg[0], isa, cg);
                    db[0] = skip ? db[0] : fmaf(db[0], isa, cb2); da[0] = skip ? da[0] : fmaf(da[0], isa, src_a);
                }
                if (NPX == 4) {
                    { const int t0 = A[0]; A[0] = A[1 % NPX]; A[1 % NPX] = A[2 % NPX]; A[2 % NPX] = A[3 % NPX]; A[3 % NPX] = t0; }
#define ROT4(v) do { const float t0_ = v[0]; v[0] = v[1 % NPX]; v[1 % NPX] = v[2 % NPX]; v[2 % NPX] = v[3 % NPX]; v[3 % NPX] = t0_; } while (0)
                    ROT4(dr); ROT4(dg); ROT4(db); ROT4(da); ROT4(clip_mask);
#undef ROT4
                }
            }
            PP_STAMP(10); PP_COUNT(22, 1);                              // 10: coverage + fill + blend of any other layer (22: how many)
        }
    }
    // ---- compute_srgb :466-483 + channel select, straight to the row-major RGBA8 image ----------------------------------
    uint32_t chan_sel = 0;                                              // v_perm_b32 selector: channel.rs:44-55 as byte indices
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t ch = (P.channels >> (8 * c)) & 0xFFu;
        chan_sel |= (ch <= 3u ? ch : (ch == 4u ? 0x0Cu : 0x0Du)) << (8 * c);     // 0x0C -> 0x00 (Zero), 0x0D -> 0xFF (One)
    }
#pragma unroll
    for (int q = 0; q < NPX; q++) {
        const uint32_t py = ty * 16u + (uint32_t)(row0 + q);
        if (px < P.width && py < P.height) {
            const float sr = linear_to_srgb(dr[q]), sg = linear_to_srgb(dg[q]), sb2 = linear_to_srgb(db[q]);
            // the four candidates as bytes once, then the channel order with one byte permute (the selectors are uniform; a
            // float select per output channel was 16 compare-and-select chains per lane)
            const uint32_t rgba = to_u8_x8(sr) | (to_u8_x8(sg) << 8) | (to_u8_x8(sb2) << 16) | (to_u8_x8(da[q]) << 24);
            ((uint32_t*)image)[(size_t)py * P.stride_px + px] = __builtin_amdgcn_perm(0u, rgba, chan_sel);
        }
    }
    PP_STAMP(9);                                                        // 9: sRGB encode + store
    if (cache.tiles && lane == 0) {                                     // update_solid_color(None): painted, not solid
        cache.tiles[tile] = make_uint2((ct_tags & 2u) | (ne << 8), ct_solid);
        cache.written[tile] = 1;
    }
    tile_done();
}

// ================================================================================================
// The QUAD painter: ONE wavefront paints FOUR neighbouring tiles of a row, for scenes whose layers are all solid / Over /
// unclipped (SIMPLE) and whose tiles are shallow — the 8192 x 8192 triangle scene: 262 144 tiles of 2.7 entries and 40
// segments, where k_paint_wave spends 70 % of a tile's 21 k clocks on the dependent round trips and LDS phases that build a
// three-entry list with three of its 64 lanes.  Here the 16-lane group g builds tile g's list — tile table, record probe,
// span scan, merge, optimizer pass, solid fold: the same steps, four tiles wide, lists of <= 16 entries held one per lane —
// and only the pixel work (segments -> coverage -> blend -> sRGB) runs tile after tile with all 64 lanes, exactly as in
// k_paint_wave<SIMPLE>.  A tile with more than 16 entries goes to k_paint_deep like a list beyond WMAX does.
// No buffer-layer cache (the host keeps k_paint_wave for cache frames), no clips (SIMPLE).
// ================================================================================================
#define QE 16             // entries per tile = lanes per group
#ifndef QUAD_ROW_XCD
#define QUAD_ROW_XCD 1
#endif
template <bool ONE_SLICE>
__global__ __launch_bounds__(64, PAINT_SIMPLE_OCC) void k_paint_quad(PaintParams P, const uint64_t* __restrict__ sorted,
                                                                    const TileRecord* __restrict__ records, DevCount nc_runs,
                                                                    const uint32_t* __restrict__ tile_first_run,
                                                                    const uint32_t* __restrict__ row_span_lo,
                                                                    const uint32_t* __restrict__ row_span_cnt,
                                                                    const uint64_t* __restrict__ span_key,
                                                                    const uint4* __restrict__ span_cov, const uint4* __restrict__ layer_col,
                                                                    uint8_t* __restrict__ image, FrameInfo* __restrict__ info,
                                                                    uint32_t* __restrict__ overflow_n,
                                                                    uint32_t* __restrict__ overflow_list, uint32_t deep_follows,
                                                                    SpanGroups groups) {
    __shared__ int q_cells[2][256];
    __shared__ uint32_t q_hi[2][4][QE], q_lo[2][4][QE];                 // [0]: arrival order (runs, then spans); [1]: by layer
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
    const uint32_t plan_bad = info->plan_bad;
    const uint32_t n_runs_frame = dev_count(nc_runs);
    // a quad = four consecutive tiles of one row (always inside one tile-column group: 4 divides SPAN_GROUP_TILES); XCD-aware
    // like k_paint_wave: workgroup b runs on XCD b % 8, every XCD a contiguous band of quads
    const uint32_t qw = (P.tiles_w + 3u) / 4u;
    const uint32_t bid = blockIdx.x;
#if QUAD_ROW_XCD
    // (the rows dealt over the XCDs like k_paint_wave's: band x = the rows x, x + 8, ...)
    const uint32_t qrow = (bid & 7u) + 8u * ((bid >> 3) / qw), qx = (bid >> 3) % qw;
    if (qrow >= P.crop_y1 - P.crop_y0) return;
    const uint32_t ty = P.crop_y0 + qrow;
#else
    const uint32_t Q = (P.crop_y1 - P.crop_y0) * qw, per = (Q + 7u) / 8u;
    if ((bid >> 3) >= per) return;
    const uint32_t qidx = (bid & 7u) * per + (bid >> 3);
    if (qidx >= Q) return;
    const uint32_t ty = P.crop_y0 + qidx / qw, qx = qidx % qw;
#endif
    const uint32_t tx = qx * 4u + (uint32_t)g;                          // this group's tile
    const bool t_in = tx < P.tiles_w && tx >= P.crop_x0 && tx < P.crop_x1;   // (a tile outside the canvas / the crop: its group idles)
    const uint32_t tile = ty * P.tiles_w + min(tx, P.tiles_w - 1u);
    const uint32_t my_tile_key = ((ty + 1u) << 12) | (tx + 1u);
    const uint32_t n_runs = PAINT_RUN_END(P, ty, n_runs_frame);
    const uint32_t j0 = t_in ? tile_first_run[tile] - 1u : FORMA_NONE;  // 0 stored = no run -> FORMA_NONE
    constexpr int NS = ONE_SLICE ? 1 : CR_MAX_SLICES;
    SpanListsT<NS> SL = load_span_lists_t<NS>(row_span_lo, row_span_cnt, ty, P.n_slices);
    bool by_group = groups.tab != nullptr;
    if (by_group) {
        const SpanListsT<NS> GL = load_group_lists<NS>(groups.tab + (size_t)ty * P.n_slices * P.n_groups + ((qx * 4u) >> SPAN_GROUP_SHIFT), P.n_groups, P.n_slices);
        if (GL.total == SPAN_GROUP_NONE) by_group = false;
        else SL = GL;
    }
    const uint32_t sc = SL.total;
    if (plan_bad) return;
    auto fetch = [&](uint32_t i, uint32_t& h, uint32_t& l, uint32_t& x) {          // candidate span i of the row / group list
        h = 0; l = 0; x = 0;
        if (i < sc) {
            const uint32_t ph = span_phys(SL, i);
            if (by_group) { const uint4 e = groups.list[ph]; h = e.x; l = e.y; x = e.z; }
            else { const uint64_t k = span_key[ph]; h = (uint32_t)(k >> 32); x = (uint32_t)k; l = REF_SPAN | ((x >> 31) ? REF_UNCH : 0u) | ph; }
        }
    };
    auto covers = [&](uint32_t x, uint32_t t) { const uint32_t lo = (x >> 16) & 0x7FFFu, hi = x & 0xFFFFu; return t >= lo && t < hi; };
    // the first 64 candidate spans and the first 16 records of every tile: all loads in flight together
    uint32_t eh, el, ex;
    fetch((uint32_t)lane, eh, el, ex);
    uint32_t r_tile = 0, r_layer = 0;
    if (j0 != FORMA_NONE && j0 + (uint32_t)li < n_runs) { const TileRecord* r = &records[j0 + (uint32_t)li]; r_tile = r->tile; r_layer = r->layer; }
    // ---- occluders (PaintParams::cull): per tile, the topmost span that crosses it with a full opaque cover ----------------
    uint32_t occ = 0;                                                   // (the same in the 16 lanes of a group)
    if (P.cull) {
        for (uint32_t c = 0; c < sc; c += 64) {
            if (c) fetch(c + (uint32_t)lane, eh, el, ex);
            const bool oc = span_is_occluder(eh);
#pragma unroll
            for (int t = 0; t < 4; t++) {                               // (lists ascend in layer: the last one found is the topmost)
                const uint64_t ob = __ballot(oc && covers(ex, qx * 4u + (uint32_t)t));
                if (ob) { const uint32_t o = ((uint32_t)__builtin_amdgcn_readlane((int)eh, 63 - __builtin_clzll(ob)) & LAYER_MASK) + 1u; if (g == t) occ = o; }
            }
        }
        if (sc > 64u) fetch((uint32_t)lane, eh, el, ex);
    }
    // ---- the tiles' own runs: 16 records per round and group, the kept ones (ascending layer: a suffix) to q_*[0][g] -------
    uint32_t ne = 0;                                                    // entries of this group's tile so far
    bool over = false;                                                  // more than QE entries: k_paint_deep paints the tile
    for (uint32_t c = 0;; c += QE) {
        if (c) {
            r_tile = 0; r_layer = 0;
            if (j0 != FORMA_NONE && j0 + c + (uint32_t)li < n_runs) { const TileRecord* r = &records[j0 + c + (uint32_t)li]; r_tile = r->tile; r_layer = r->layer; }
        }
        const bool mine = j0 != FORMA_NONE && (r_tile & 0x7FFFFFFFu) == my_tile_key;
        const bool keep = mine && (r_layer & LAYER_MASK) + 1u >= occ;
        const uint64_t mb = __ballot(mine), kb = __ballot(keep);
        const uint32_t gk = (uint32_t)(kb >> (16 * g)) & 0xFFFFu, gm = (uint32_t)(mb >> (16 * g)) & 0xFFFFu;
        if (keep) {
            const uint32_t pos = ne + (uint32_t)__popc(gk & ((1u << li) - 1u));
            if (pos < QE) { q_hi[0][g][pos] = r_layer; q_lo[0][g][pos] = ((r_tile >> 31) ? REF_UNCH : 0u) | (j0 + c + (uint32_t)li); }
        }
        ne += (uint32_t)__popc(gk);
        // another round while some tile's 16 probes were all its own
        if (!__any(gm == 0xFFFFu)) break;
        if (gm != 0xFFFFu) { /* this group is done: its later probes are not `mine` (j0 + c runs past the tile's runs) */ }
    }
    // ---- spans that cross the tiles -----------------------------------------------------------------------------------------
    for (uint32_t c = 0; c < sc; c += 64) {
        if (c) fetch(c + (uint32_t)lane, eh, el, ex);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t occ_t = (uint32_t)__builtin_amdgcn_readlane((int)occ, 16 * t);
            const uint32_t ne_t = (uint32_t)__builtin_amdgcn_readlane((int)ne, 16 * t);
            const bool hit = covers(ex, qx * 4u + (uint32_t)t) && (eh & LAYER_MASK) + 1u >= occ_t;
            const uint64_t hb = __ballot(hit);
            if (hit) {
                const uint32_t pos = ne_t + __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0u));
                if (pos < QE) { q_hi[0][t][pos] = eh; q_lo[0][t][pos] = el; }
            }
            if (g == t) ne += (uint32_t)__popcll(hb);
        }
    }
    if (!t_in) ne = 0;
    if (ne > QE) {
        over = true;
        if (li == 0) {
            overflow_list[atomicAdd(overflow_n, 1u)] = tile;
            atomicOr(&info->error, 16u);
            if (!deep_follows) info->plan_bad = 1u;
        }
        ne = 0;
    }
    wave_lds_sync();
    // ---- merge by layer: lane li holds entry li of its tile; rank = entries of the tile with a smaller layer (layers are unique) ---
    const bool have = (uint32_t)li < ne;
    uint32_t a_hi = have ? q_hi[0][g][li] : 0u, a_lo = have ? q_lo[0][g][li] : 0u;
    {
        const uint32_t ne_max = wave_max_u32(ne);
        uint32_t rank = 0;
        const uint32_t myl = a_hi & LAYER_MASK;
        for (uint32_t j = 0; j < ne_max; j++) if (j < ne && (q_hi[0][g][j] & LAYER_MASK) < myl) rank++;
        if (have) { q_hi[1][g][rank] = a_hi; q_lo[1][g][rank] = a_lo; }
    }
    wave_lds_sync();
    a_hi = have ? q_hi[1][g][li] : 0u; a_lo = have ? q_lo[1][g][li] : 0u;   // entry li of the tile's layer list
    // ---- skip_fully_covered_layers_pass (SIMPLE: every layer is a solid colour, Over, unclipped) ------------------------------
    const uint32_t sfl = a_hi >> 21;
    const bool is_span = (a_lo & 0x80000000u) != 0u;
    const bool full = have && is_span && (sfl & SF_FULL);
    const uint64_t topb = __ballot(full && (sfl & SF_OPAQUE)), blkb = __ballot(have && !full);
    const uint32_t gt = (uint32_t)(topb >> (16 * g)) & 0xFFFFu, gb = (uint32_t)(blkb >> (16 * g)) & 0xFFFFu;
    const uint32_t top = gt ? 32u - (uint32_t)__builtin_clz(gt) : 0u, blk = gb ? 32u - (uint32_t)__builtin_clz(gb) : 0u;   // index + 1
    const uint32_t skipped = top ? top - 1u : 0u;
    const int first = top ? (blk > top ? 2 : 1) : (blk ? 2 : 0);
    // what the fold and the layer loop need of the entries from `skipped` up: requested by the entry's own lane, once
    const bool used = have && (uint32_t)li >= skipped;
    const uint32_t layer = a_hi & LAYER_MASK;
    uint4 pcol = make_uint4(0u, 0u, 0u, 0u), pcov = make_uint4(0u, 0u, 0u, 0u);
    uint32_t pseg0 = 0, pnseg = 0;
    if (used) {
        pcol = layer_col[layer];
        if (first == 2) {
            if (is_span) pcov = span_cov[a_lo & REF_IDX];
            else { const TileRecord* r = &records[a_lo & REF_IDX]; pcov = make_uint4(r->cover[0], r->cover[1], r->cover[2], r->cover[3]); pseg0 = r->seg_start; pnseg = r->seg_count; }
        }
    }
    const Col clear = {P.clear[0], P.clear[1], P.clear[2], P.clear[3]};
    // ---- fold (first != 2): every layer from `skipped` up is a full cover -> the tile is one colour.  The four tiles fold side by
    //      side: step k blends entry skipped + k of every tile (its lane's colour, fetched through the group) ------------------------
    {
        Col dst = clear;
        const uint32_t steps = wave_max_u32((first != 2 && t_in && !over) ? ne - skipped : 0u);
        for (uint32_t k = 0; k < steps; k++) {
            const int src_lane = 16 * g + (int)min(skipped + k, (uint32_t)QE - 1u);
            const Col src = {__uint_as_float((uint32_t)__shfl((int)pcol.x, src_lane, 64)), __uint_as_float((uint32_t)__shfl((int)pcol.y, src_lane, 64)),
                             __uint_as_float((uint32_t)__shfl((int)pcol.z, src_lane, 64)), __uint_as_float((uint32_t)__shfl((int)pcol.w, src_lane, 64))};
            if (first != 2 && skipped + k < ne) {
                if (first == 1 && k == 0) dst = src; else dst = sc_blend(0u, dst, src);
            }
        }
        if (first != 2 && t_in && !over) {                              // TileWriteOp::Solid: to_srgb_bytes :156-162, 690
            float sel[4];
#pragma unroll
            for (int c = 0; c < 4; c++) sel[c] = sel_channel((P.channels >> (8 * c)) & 0xFFu, dst.r, dst.g, dst.b, dst.a);
            const uint32_t bytes = to_u8_x4(linear_to_srgb(sel[0])) | (to_u8_x4(linear_to_srgb(sel[1])) << 8) |
                                   (to_u8_x4(linear_to_srgb(sel[2])) << 16) | (to_u8_x4(sel[3]) << 24);
            const uint32_t px = tx * 16u + (uint32_t)li;
#pragma unroll 4
            for (int row = 0; row < 16; row++) {                         // the group's 16 lanes: one pixel column each
                const uint32_t py = ty * 16u + (uint32_t)row;
                if (px < P.width && py < P.height) ((uint32_t*)image)[(size_t)py * P.stride_px + px] = bytes;
            }
        }
    }
    // ---- the tiles that are painted, one after the other with all 64 lanes (Painter::paint_layer, painter/mod.rs:290-347:
    //      k_paint_wave<SIMPLE>'s layer loop; an entry's data comes out of its lane by readlane) --------------------------------
    const bool paint_me = first == 2 && t_in && !over;
    if (!__any(paint_me)) return;
#pragma unroll
    for (int q = 0; q < 4; q++) { q_cells[0][q * 64 + lane] = 0; q_cells[1][q * 64 + lane] = 0; }
    uint32_t chan_sel = 0;                                              // v_perm_b32 selector: channel.rs:44-55 as byte indices
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t ch = (P.channels >> (8 * c)) & 0xFFu;
        chan_sel |= (ch <= 3u ? ch : (ch == 4u ? 0x0Cu : 0x0Du)) << (8 * c);
    }
    const int lx = lane & 15, rg = lane >> 4;                           // pixel mapping of the pixel phase: (lx, 4 * rg + q)
    uint32_t cbuf = 0;
    for (int t = 0; t < 4; t++) {
        if (!__builtin_amdgcn_readlane((int)paint_me, 16 * t)) continue;
        const uint32_t ne_t = (uint32_t)__builtin_amdgcn_readlane((int)ne, 16 * t), sk_t = (uint32_t)__builtin_amdgcn_readlane((int)skipped, 16 * t);
        const uint32_t ttx = qx * 4u + (uint32_t)t, px = ttx * 16u + (uint32_t)lx;
        float dr[4], dg[4], db[4], da[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { dr[q] = clear.r; dg[q] = clear.g; db[q] = clear.b; da[q] = clear.a; }
        for (uint32_t e = sk_t; e < ne_t; e++) {
            const int sl = 16 * t + (int)e;
            const uint32_t e_sfl = (uint32_t)__builtin_amdgcn_readlane((int)a_hi, sl) >> 21;
            const uint32_t nseg = (uint32_t)__builtin_amdgcn_readlane((int)pnseg, sl);
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)pcov.x, sl), c1 = (uint32_t)__builtin_amdgcn_readlane((int)pcov.y, sl);
            const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)pcov.z, sl), c3 = (uint32_t)__builtin_amdgcn_readlane((int)pcov.w, sl);
            const uint32_t cw = rg == 0 ? c0 : (rg == 1 ? c1 : (rg == 2 ? c2 : c3));
            int A[4];
            if (nseg) {
                int* cb = q_cells[cbuf];
                const uint64_t* sp = sorted + (uint32_t)__builtin_amdgcn_readlane((int)pseg0, sl);
                for (uint32_t sidx = lane; sidx < nseg; sidx += 64) {   // acc_segment :257-271
                    const uint64_t v = sp[sidx];
                    const int cv = seg_cover(v);
                    atomicAdd(&cb[seg_ly(v) * 16 + seg_lx(v)], (int)((uint32_t)(seg_dam(v) * cv) << 16) + cv);
                }
                wave_lds_sync();
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int ci = (rg * 4 + q) * 16 + lx;
                    const int S = cb[ci];
                    cb[ci] = 0;
                    const int c = (int)(int16_t)(S & 0xFFFF);
                    const int area = (int)(int16_t)((uint32_t)(S - c) >> 16);
                    int inc = c;
                    inc += dpp_row_shr(inc, 1); inc += dpp_row_shr(inc, 2); inc += dpp_row_shr(inc, 4); inc += dpp_row_shr(inc, 8);
                    const int carry = (int)(int8_t)(cw >> (q * 8));
                    const int acc = (int)(int8_t)(carry + (inc - c));
                    A[q] = 32 * acc + area;
                }
                cbuf ^= 1u;
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) A[q] = 32 * (int)(int8_t)(cw >> (q * 8));
            }
            const bool eo = (e_sfl & SF_EVENODD) != 0;
            const float fr = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)pcol.x, sl)), fg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)pcol.y, sl));
            const float fb = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)pcol.z, sl)), fa = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)pcol.w, sl));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float cov = coverage_of(A[q], eo);
                const float src_a = fa * cov;                           // blend_at :406-447 with blend = Over
                const float ida = 1.0f - da[q], k1 = ida * src_a, isa = 1.0f - src_a, k2 = da[q] * src_a;
                const float nr = fmaf(dr[q], isa, fmaf(fr, k1, fr * k2));
                const float ng = fmaf(dg[q], isa, fmaf(fg, k1, fg * k2));
                const float nb2 = fmaf(db[q], isa, fmaf(fb, k1, fb * k2));
                const float na2 = fmaf(da[q], isa, src_a);
                const bool skip = cov == 0.0f;                          // :317-319
                dr[q] = skip ? dr[q] : nr; dg[q] = skip ? dg[q] : ng; db[q] = skip ? db[q] : nb2; da[q] = skip ? da[q] : na2;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {                                   // compute_srgb :466-483 + channel select
            const uint32_t py = ty * 16u + (uint32_t)(rg * 4 + q);
            if (px < P.width && py < P.height) {
                const float sr = linear_to_srgb(dr[q]), sg = linear_to_srgb(dg[q]), sb2 = linear_to_srgb(db[q]);
                const uint32_t rgba = to_u8_x8(sr) | (to_u8_x8(sg) << 8) | (to_u8_x8(sb2) << 16) | (to_u8_x8(da[q]) << 24);
                ((uint32_t*)image)[(size_t)py * P.stride_px + px] = __builtin_amdgcn_perm(0u, rgba, chan_sel);
            }
        }
    }
}

#define PAINT_ARGS P, tile, sorted, records, n_runs, tile_first_run, row_span_lo, row_span_cnt, span_key, span_cov, layer_col, \
                   style_offsets, style_words, images, texels, image, cache, info
#define PAINT_PARAMS PaintParams P, const uint64_t* __restrict__ sorted, const TileRecord* __restrict__ records, DevCount nc_runs, \
                     const uint32_t* __restrict__ tile_first_run, const uint32_t* __restrict__ row_span_lo, \
                     const uint32_t* __restrict__ row_span_cnt, const uint64_t* __restrict__ span_key, \
                     const uint4* __restrict__ span_cov, const uint4* __restrict__ layer_col, \
                     const uint32_t* __restrict__ style_offsets, const uint32_t* __restrict__ style_words, \
                     const forma_image_t* __restrict__ images, const uint16_t* __restrict__ texels, uint8_t* __restrict__ image, \
                     TileCacheArgs cache, FrameInfo* __restrict__ info

#define PAINT_MAXE_DEEP 4096
#define PAINT_MAXE_MID  1024
#define PAINT_ARGS2 P, tile, sorted, records, n_runs, tile_first_run, row_span_lo, row_span_cnt, span_key, span_cov, layer_col, \
                    style_offsets, style_words, images, texels, image, cache, info
// The deep tiles the first launch could not hold, in two tiers.  MAXE = 1024 (20 KB of lists + 7 KB of paint_tile's own LDS,
// 105 VGPRs: FOUR workgroups per CU) takes the wave painters' overflow list; what does not fit it goes on to MAXE = 4096 (80 KB of
// lists: one workgroup per CU), and from there to k_paint_huge.  Until round 5 there was only the 4096-entry kernel: one
// 256-lane workgroup per CU — 1 wave per SIMD — painted every tile beyond the wave painters' 128 entries, and the reference
// demo's own `circles` mode at 20 000 discs (120 layers per tile: half its tiles are "deep") spent 102 us there.
// STRIDE: the words per entry of `list` (1: tile ids — the wave painters' list; 2: {tile, entries} pairs — the mid tier's).
template <int MAXE, int STRIDE>
__global__ __launch_bounds__(256) void k_paint_deep(PAINT_PARAMS, const uint32_t* __restrict__ list_n,
                                                    const uint32_t* __restrict__ list, uint32_t* __restrict__ over_n,
                                                    uint32_t* __restrict__ over_list) {
    __shared__ uint64_t e_key[MAXE];                                   // staging for the span scan (4 x MAXE / 4), then the merged layer list
    __shared__ uint64_t e_tmp[MAXE];                                   // [0, na) own runs, [na, ne) crossing spans; later the painted entries
    __shared__ uint32_t e_flag[MAXE];
    if (info->plan_bad) return;
    const uint32_t n = list_n[0];
    const uint32_t n_runs = dev_count(nc_runs);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t tile = list[(size_t)STRIDE * i];
        paint_tile(MAXE, MAXE / 4, e_key, e_tmp, e_flag, PAINT_ARGS2, over_n, over_list);
        __syncthreads();
    }
}
// tiles deeper than the LDS lists: the same code with lists in global memory, sized per tile by the host from the entry
// counts k_paint_deep recorded (offs[i] = first entry slot of tile i; the staging array is four times as long)
__global__ __launch_bounds__(256) void k_paint_huge(PAINT_PARAMS, const uint32_t* __restrict__ over2_list, uint32_t n_tiles,
                                                    const uint64_t* __restrict__ offs, uint64_t* __restrict__ g_key,
                                                    uint64_t* __restrict__ g_tmp, uint32_t* __restrict__ g_flag) {
    if (info->plan_bad) return;
    const uint32_t n_runs = dev_count(nc_runs);
    for (uint32_t i = blockIdx.x; i < n_tiles; i += gridDim.x) {
        const uint32_t tile = over2_list[2 * i], cap = over2_list[2 * i + 1];
        const uint64_t o = offs[i];
        paint_tile(cap, cap, g_key + 4 * o, g_tmp + o, g_flag + o, PAINT_ARGS2, nullptr, nullptr);
        __syncthreads();
    }
}

void launch_paint(hipStream_t s, const PaintParams& p, const uint64_t* sorted, const TileRecord* records, DevCount n_runs,
                  const uint32_t* tile_first_run, const uint32_t* row_span_lo, const uint32_t* row_span_cnt,
                  const uint64_t* span_key, const uint4* span_cov, const uint4* layer_col,
                  const uint32_t* style_offsets, const uint32_t* style_words, const forma_image_t* images,
                  const uint16_t* texels, uint8_t* image, TileCacheArgs cache, FrameInfo* info, uint32_t* overflow_n,
                  uint32_t* overflow_list, uint32_t* over2_n, uint32_t* over2_list, bool launch_deep, SpanGroups groups, bool strips, bool quads,
                  uint32_t* mid_n, uint32_t* mid_list, uint32_t n_cus) {
    const uint32_t T = p.tiles_w * p.tiles_h;
    if (T == 0 || p.crop_y1 <= p.crop_y0) return;
    const uint32_t per = paint_band_tiles(p.crop_y1 - p.crop_y0, p.tiles_w);
    static const ForMaDebug dbg = forma_debug_parse();               // FORMA_HIP_DEBUG (debug.h), process-wide for these two
    static const bool no_simple = dbg.no_simple_paint;               // (A/B switches for tools/)
    static const bool force_simple = dbg.force_simple_paint;         // (timing experiments only: wrong pixels on other scenes)
    const bool simple = (p.scene_simple && !no_simple) || force_simple, one = p.n_slices == 1u;
    // strips (four wavefronts per tile, NPX = 1): never with a buffer-layer cache — a tile's cache entry is read by every strip
    // and rewritten by the first one that finishes
    if (cache.tiles) strips = false;
    // quads (k_paint_quad: four tiles per wavefront): all-solid scenes with shallow tiles, no cache
    if (quads && simple && !cache.tiles) {
        const uint32_t qper = QUAD_ROW_XCD ? ((p.crop_y1 - p.crop_y0 + 7u) / 8u) * ((p.tiles_w + 3u) / 4u)
                                            : ((p.crop_y1 - p.crop_y0) * ((p.tiles_w + 3u) / 4u) + 7u) / 8u;
        if (one) FORMA_LAUNCH(k_paint_quad<true>, dim3(qper * 8), dim3(64), 0, s, p, sorted, records, n_runs, tile_first_run, row_span_lo, row_span_cnt,
                              span_key, span_cov, layer_col, image, info, overflow_n, overflow_list, launch_deep ? 1u : 0u, groups);
        else FORMA_LAUNCH(k_paint_quad<false>, dim3(qper * 8), dim3(64), 0, s, p, sorted, records, n_runs, tile_first_run, row_span_lo, row_span_cnt,
                          span_key, span_cov, layer_col, image, info, overflow_n, overflow_list, launch_deep ? 1u : 0u, groups);
    } else {
#define PW_LAUNCH(S_, O_, N_) FORMA_LAUNCH((k_paint_wave<S_, O_, N_>), dim3(N_ == 1 ? per * 32 : (per + (p.order_cnt_out ? p.order_hcap : 0u)) * 8), dim3(64), 0, s, p, sorted, records, n_runs, \
                                             tile_first_run, row_span_lo, row_span_cnt, span_key, span_cov, layer_col, style_offsets, style_words, \
                                             images, texels, image, cache, info, overflow_n, overflow_list, launch_deep ? 1u : 0u, groups)
#define PW_LAUNCH_N(S_, O_) do { if (strips) PW_LAUNCH(S_, O_, 1); else PW_LAUNCH(S_, O_, 4); } while (0)
    if (simple) { if (one) PW_LAUNCH_N(true, true); else PW_LAUNCH_N(true, false); }
    else { if (one) PW_LAUNCH_N(false, true); else PW_LAUNCH_N(false, false); }
#undef PW_LAUNCH_N
#undef PW_LAUNCH
    }
    if (!launch_deep) return;                             // (read-back-free frame of a scene whose last frame had no deep tile)
    // mid tier: four workgroups per CU; its own overflow ({tile, entries} pairs in mid_list) goes to the 4096-entry tier, whose
    // launch is empty (~4 us) in every frame without a tile beyond 1024 entries — paid only by frames that have deep tiles at all
    const uint32_t g_mid = std::min<uint32_t>(T, 4u * n_cus), g_deep = std::min<uint32_t>(T, n_cus);
    FORMA_LAUNCH((k_paint_deep<PAINT_MAXE_MID, 1>), dim3(g_mid), dim3(256), 0, s, p, sorted, records, n_runs, tile_first_run,
                       row_span_lo, row_span_cnt, span_key, span_cov, layer_col, style_offsets, style_words, images,
                       texels, image, cache, info, (const uint32_t*)overflow_n, (const uint32_t*)overflow_list, mid_n, mid_list);
    FORMA_LAUNCH((k_paint_deep<PAINT_MAXE_DEEP, 2>), dim3(g_deep), dim3(256), 0, s, p, sorted, records, n_runs, tile_first_run,
                       row_span_lo, row_span_cnt, span_key, span_cov, layer_col, style_offsets, style_words, images,
                       texels, image, cache, info, (const uint32_t*)mid_n, (const uint32_t*)mid_list, over2_n, over2_list);
}

void launch_paint_huge(hipStream_t s, const PaintParams& p, const uint64_t* sorted, const TileRecord* records, DevCount n_runs,
                       const uint32_t* tile_first_run, const uint32_t* row_span_lo, const uint32_t* row_span_cnt,
                       const uint64_t* span_key, const uint4* span_cov, const uint4* layer_col,
                       const uint32_t* style_offsets, const uint32_t* style_words, const forma_image_t* images,
                       const uint16_t* texels, uint8_t* image, TileCacheArgs cache, FrameInfo* info, const uint32_t* over2_list,
                       uint32_t n_tiles, const uint64_t* offs, uint64_t* g_key, uint64_t* g_tmp, uint32_t* g_flag) {
    if (n_tiles == 0) return;
    FORMA_LAUNCH(k_paint_huge, dim3(n_tiles < 256 ? n_tiles : 256), dim3(256), 0, s, p, sorted, records, n_runs, tile_first_run,
                       row_span_lo, row_span_cnt, span_key, span_cov, layer_col, style_offsets, style_words, images, texels, image,
                       cache, info, over2_list, n_tiles, offs, g_key, g_tmp, g_flag);
}

// ================================================================================================
// copy-out of a cache frame: only the tiles the painters wrote (TileWriteOp != None) leave the device.
// k_written_list compacts the written tiles of the crop, row-major, into a list (one workgroup: the canvas has
// at most a few hundred thousand tiles); k_pack_written gathers their pixels into 1 KB slots, list order.
// The host walks the same flags in the same order and drops slot k into the k-th written tile of the caller's
// buffer (reference cpu/buffer/layout/mod.rs:264-295 writes tile by tile).
// ================================================================================================
__global__ __launch_bounds__(1024) void k_written_list(const uint8_t* __restrict__ written, uint32_t tiles_w, uint32_t tx0, uint32_t tx1,
                                                       uint32_t ty0, uint32_t ty1, uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
    __shared__ uint32_t s_w[16], s_base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t cw = tx1 - tx0, n = cw * (ty1 - ty0);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
        const uint32_t i = i0 + (uint32_t)tid;
        uint32_t tile = 0; bool hit = false;
        if (i < n) { const uint32_t ry = i / cw, rx = i - ry * cw; tile = (ty0 + ry) * tiles_w + tx0 + rx; hit = written[tile] != 0; }
        const uint64_t bal = __ballot(hit);
        if (lane == 0) s_w[w] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t base = s_base, tot = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) { const uint32_t t = s_w[q]; if (q < w) base += t; tot += t; }
        if (hit) list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = tile;
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) *count = s_base;
}
__global__ __launch_bounds__(256) void k_pack_written(const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, uint32_t max_pack,
                                                      uint32_t tiles_w, const uint32_t* __restrict__ image, uint32_t width, uint32_t height,
                                                      uint32_t* __restrict__ packed) {
    const uint32_t n = *count;
    if (n > max_pack) return;                              // (too many: the host copies the whole crop instead)
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const uint32_t tile = list[k], ty = tile / tiles_w, tx = tile - ty * tiles_w;
        const uint32_t px = tx * 16u + (threadIdx.x & 15u), py = ty * 16u + (threadIdx.x >> 4);
        packed[(size_t)k * 256 + threadIdx.x] = (px < width && py < height) ? image[(size_t)py * width + px] : 0u;
    }
}
void launch_pack_written(hipStream_t s, const uint8_t* written, uint32_t tiles_w, uint32_t tx0, uint32_t tx1, uint32_t ty0, uint32_t ty1,
                         uint32_t* list, uint32_t* count, uint32_t max_pack, const uint8_t* image, uint32_t width, uint32_t height,
                         uint32_t* packed) {
    if (tx0 >= tx1 || ty0 >= ty1) { (void)hipMemsetAsync(count, 0, 4, s); return; }
    FORMA_LAUNCH(k_written_list, dim3(1), dim3(1024), 0, s, written, tiles_w, tx0, tx1, ty0, ty1, list, count);
    const uint32_t n = (tx1 - tx0) * (ty1 - ty0);
    FORMA_LAUNCH(k_pack_written, dim3(std::max(1u, std::min<uint32_t>(std::min(n, max_pack), 4096u))), dim3(256), 0, s, (const uint32_t*)list,
                       (const uint32_t*)count, max_pack, tiles_w, (const uint32_t*)image, width, height, packed);
}

// ================================================================================================
// the end of a read-back-free frame (launch_frame_tail, common.h)
// ================================================================================================
__global__ __launch_bounds__(64) void k_frame_tail(FrameInfo* __restrict__ info, FrameInfo* __restrict__ host_info,
                                                   uint32_t* __restrict__ host_count, const uint32_t* __restrict__ order_cnt,
                                                   uint32_t* __restrict__ order_keep, const uint32_t* __restrict__ chain_rows,
                                                   uint32_t n_chain_rows, uint32_t* __restrict__ host_seq, uint32_t seq) {
    constexpr int W = (int)(sizeof(FrameInfo) / 4);
    static_assert(W <= 64, "FrameInfo fits one wave");
    const int t = threadIdx.x;
    uint32_t* src = reinterpret_cast<uint32_t*>(info);
    // the painters' list counts live in the frame's tile tables, which the next frame clears: kept where it will look for them
    uint32_t heavy = 0;
    if (order_cnt) {
        for (int i = t; i < (int)PAINT_ORDER_WORDS; i += 64) { const uint32_t c = order_cnt[i]; order_keep[i] = c; heavy += c; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) heavy += __shfl_xor(heavy, d, 64);
    }
    // launch_runs' chain numbering: nobody counted the frame's runs — they are the sum of the row counts
    uint32_t runs = 0;
    if (chain_rows) {
        for (uint32_t i = (uint32_t)t; i < n_chain_rows; i += 64) runs += chain_rows[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) runs += __shfl_xor(runs, d, 64);
    }
    if (t < W) {
        uint32_t v = t == (int)(offsetof(FrameInfo, n_heavy) / 4) ? heavy : src[t];
        if (chain_rows && t == (int)(offsetof(FrameInfo, n_runs) / 4)) v = runs;
        // pinned host memory: system-scope stores, visible to the host once the stream has drained
        if (host_info) __hip_atomic_store(reinterpret_cast<uint32_t*>(host_info) + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (host_count && t == (int)(offsetof(FrameInfo, n_segments) / 4)) __hip_atomic_store(host_count, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // the pristine FrameInfo: zeros, and all-ones where the keys' AND accumulates (forma_hip_create's template)
        const bool ones = t == (int)(offsetof(FrameInfo, key_and) / 4) || t == (int)(offsetof(FrameInfo, key_and_hi) / 4);
        src[t] = ones ? 0xFFFFFFFFu : 0u;
    }
    // the frame's number, LAST: the release waits for every store of this wave (vmcnt(0) is the wave's), so a host that sees the
    // number sees the whole FrameInfo — it polls this word instead of waiting for the stream's completion signal
    if (host_seq && t == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_frame_tail(hipStream_t s, FrameInfo* info, FrameInfo* host_info, uint32_t* host_count, const uint32_t* order_cnt,
                       uint32_t* order_keep, const uint32_t* chain_rows, uint32_t n_chain_rows, uint32_t* host_seq, uint32_t seq) {
    FORMA_LAUNCH(k_frame_tail, dim3(1), dim3(64), 0, s, info, host_info, host_count, order_cnt, order_keep, chain_rows, n_chain_rows, host_seq, seq);
}

