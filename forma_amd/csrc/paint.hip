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

// ================================================================================================
// bounds of the paintable part of the sorted stream (painter/mod.rs:731-734: tile_y < 0 dropped;
// rows >= tiles_h are never visited)
// ================================================================================================
__global__ void k_find_bounds(const uint64_t* __restrict__ sorted, uint32_t n, uint32_t tiles_h, FrameInfo* info) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if ((sorted[m] >> 53) >= 1u) hi = m; else lo = m + 1; }
    info->seg_begin = lo;
    hi = n;
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if ((sorted[m] >> 53) >= (uint64_t)tiles_h + 1u) hi = m; else lo = m + 1; }
    info->seg_end = lo;
}
void launch_find_bounds(hipStream_t s, const uint64_t* sorted, uint32_t n, uint32_t tiles_h, FrameInfo* info) {
    hipLaunchKernelGGL(k_find_bounds, dim3(1), dim3(64), 0, s, sorted, n, tiles_h, info);
}

// ================================================================================================
// runs: maximal runs of equal 44-bit key (tile_y, tile_x, layer) in the sorted stream
// ================================================================================================
#define RUN_THREADS 256
#define RUN_ITEMS   8
#define RUN_TILE    (RUN_THREADS * RUN_ITEMS)

__device__ __forceinline__ uint32_t wave_inc_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(v, d, 64); if (lane >= d) v += t; }
    return v;
}

template <bool WRITE>
__global__ __launch_bounds__(RUN_THREADS) void k_run_heads(const uint64_t* __restrict__ sorted,
                                                           const FrameInfo* __restrict__ info,
                                                           uint32_t* __restrict__ head_counts,
                                                           uint32_t* __restrict__ run_start) {
    __shared__ uint32_t lds[RUN_THREADS / 64];
    const uint32_t sb = info->seg_begin, se = info->seg_end;
    const uint32_t base = blockIdx.x * RUN_TILE + threadIdx.x * RUN_ITEMS;
    uint32_t flags = 0, cnt = 0;
    if (base < se && base + RUN_ITEMS > sb) {
        uint64_t prev = (base > sb && base > 0) ? seg_key(sorted[base - 1]) : ~0ull;
#pragma unroll
        for (int i = 0; i < RUN_ITEMS; i++) {
            uint32_t idx = base + i;
            if (idx >= sb && idx < se) {
                uint64_t k = seg_key(sorted[idx]);
                if (idx == sb || k != prev) { flags |= 1u << i; cnt++; }
                prev = k;
            }
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_inc_scan(cnt);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < RUN_THREADS / 64; i++) { uint32_t t = lds[i]; if (i < w) wbase += t; tot += t; }
    if (!WRITE) {
        if (threadIdx.x == 0) head_counts[blockIdx.x] = tot;
    } else {
        uint32_t r = head_counts[blockIdx.x] + wbase + inc - cnt;      // head_counts now holds exclusive block offsets
#pragma unroll
        for (int i = 0; i < RUN_ITEMS; i++) {
            uint32_t idx = base + i;
            if (flags & (1u << i)) run_start[r++] = idx;
            if (idx + 1 == se && idx >= sb) run_start[r] = se;          // sentinel run_start[J]
        }
    }
}

// single-block exclusive scan of per-block head counts, total -> info->n_runs
__global__ __launch_bounds__(1024) void k_scan_heads(uint32_t* __restrict__ v, uint32_t nb, FrameInfo* info) {
    __shared__ uint32_t lds[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nb; base += 1024) {
        uint32_t idx = base + threadIdx.x;
        uint32_t x = idx < nb ? v[idx] : 0;
        uint32_t inc = wave_inc_scan(x);
        if (lane == 63) lds[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { uint32_t t = lds[i]; if (i < w) wbase += t; tot += t; }
        if (idx < nb) v[idx] = s_carry + wbase + inc - x;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) info->n_runs = s_carry;
}

void launch_runs(hipStream_t s, const uint64_t* sorted, const FrameInfo* info_in, uint32_t n, uint32_t* head_counts,
                 uint32_t* scan_tmp, uint32_t* run_start, FrameInfo* info) {
    (void)scan_tmp;
    if (n == 0) { hipMemsetAsync(&info->n_runs, 0, 4, s); return; }
    uint32_t nb = (n + RUN_TILE - 1) / RUN_TILE;
    hipLaunchKernelGGL(k_run_heads<false>, dim3(nb), dim3(RUN_THREADS), 0, s, sorted, info_in, head_counts, run_start);
    hipLaunchKernelGGL(k_scan_heads, dim3(1), dim3(1024), 0, s, head_counts, nb, info);
    hipLaunchKernelGGL(k_run_heads<true>, dim3(nb), dim3(RUN_THREADS), 0, s, sorted, info_in, head_counts, run_start);
}

// ================================================================================================
// per-run cover sums (16 x i8, wrapping) — the quantity LayerWorkbench::cover_carry accumulates
// (layer_workbench/mod.rs:213-234) — plus the (tile_y, layer, run) keys for the carry scan order
// ================================================================================================
__device__ __forceinline__ uint64_t swar_add8(uint64_t a, uint64_t b) {      // 8 wrapping i8 adds
    return ((a & 0x7F7F7F7F7F7F7F7Full) + (b & 0x7F7F7F7F7F7F7F7Full)) ^ ((a ^ b) & 0x8080808080808080ull);
}

__global__ __launch_bounds__(256) void k_run_covers(const uint64_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ run_start, uint32_t n_runs,
                                                    TileRecord* __restrict__ records, uint4* __restrict__ run_cov,
                                                    uint64_t* __restrict__ run_keys, uint32_t tiles_w) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_runs; j += gridDim.x * blockDim.x) {
        uint32_t s0 = run_start[j], s1 = run_start[j + 1];
        uint64_t lo = 0, hi = 0;
        uint64_t first = sorted[s0];
        for (uint32_t s = s0; s < s1; s++) {
            uint64_t v = s == s0 ? first : sorted[s];
            int ly = seg_ly(v);
            uint64_t add = (uint64_t)((uint32_t)seg_cover(v) & 0xFFu) << ((ly & 7) * 8);
            if (ly < 8) lo = swar_add8(lo, add); else hi = swar_add8(hi, add);
        }
        TileRecord r;
        run_cov[j] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
        r.cover[0] = r.cover[1] = r.cover[2] = r.cover[3] = 0;           // carry-in, filled by k_carry
        r.seg_start = s0; r.seg_count = s1 - s0; r.layer = seg_layer(first);
        uint32_t ty = (uint32_t)(first >> 53) - 1u, txb = (uint32_t)(first >> 41) & 0xFFFu;   // txb = tile_x + 1
        r.tile = ty * tiles_w + (txb - 1u);                                                   // meaningless for txb == 0
        records[j] = r;
        run_keys[j] = ((first >> 53) << 53) | ((uint64_t)seg_layer(first) << 32) | j;
    }
}
void launch_run_covers(hipStream_t s, const uint64_t* sorted, const uint32_t* run_start, uint32_t n_runs,
                       TileRecord* records, uint4* run_cov, uint64_t* run_keys, uint32_t tiles_w) {
    if (n_runs == 0) return;
    uint32_t blocks = (n_runs + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_run_covers, dim3(blocks), dim3(256), 0, s, sorted, run_start, n_runs, records, run_cov, run_keys, tiles_w);
}

// ================================================================================================
// carry pre-pass.  Runs arrive ordered by (tile_y, layer, tile_x) — `sorted_keys` is run_keys after a
// stable radix sort on bits 32..63.  The head of each (tile_y, layer) group walks the group:
//   carry-in(run)   = wrapping sum of the covers of all runs of the group with smaller tile_x
//                     (painter/mod.rs:500-522 for the left-of-canvas bucket; layer_workbench :325-333)
//   a carry survives a tile boundary only if !Cover::is_empty(fill_rule) (painter/mod.rs:187-198)
//   tiles strictly between two runs of the group are "carry-only" tiles (span records).
// Pass 0 (fill = 0) counts the (tile, layer) pairs per tile; pass 1 writes the entries.
// ================================================================================================
__device__ __forceinline__ bool cover_is_empty(uint64_t lo, uint64_t hi, bool even_odd) {
    if (!even_odd) return (lo | hi) == 0;
    // all (|c| & 31) == 0  <=>  c mod 32 == 0 for every byte
    return ((lo | hi) & 0x1F1F1F1F1F1F1F1Full) == 0;
}

__global__ __launch_bounds__(256) void k_carry(const uint64_t* __restrict__ sorted_keys, uint32_t n_runs,
                                               TileRecord* __restrict__ records, const uint4* __restrict__ run_cov,
                                               const uint32_t* __restrict__ style_offsets,
                                               const uint32_t* __restrict__ style_words, uint32_t n_orders,
                                               uint32_t tiles_w, uint32_t tiles_h, uint32_t* __restrict__ tile_count,
                                               FrameInfo* __restrict__ info, int fill,
                                               const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ tile_fill,
                                               uint64_t* __restrict__ entries, uint32_t record_cap) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_runs; k += gridDim.x * blockDim.x) {
        uint64_t key = sorted_keys[k];
        uint32_t group = (uint32_t)(key >> 32);
        if (k > 0 && (uint32_t)(sorted_keys[k - 1] >> 32) == group) continue;     // not a group head
        uint32_t layer = group & 0x1FFFFFu;
        uint32_t ty = (group >> 21) - 1u;                                          // rows >= 0 only (bounds)
        bool even_odd = false;
        if (layer < n_orders && style_offsets[layer] != FORMA_NONE)
            even_odd = FORMA_STYLE_EVENODD(style_words[style_offsets[layer]]);
        else if (fill == 0) atomicOr(&info->error, 1u);
        uint64_t acc_lo = 0, acc_hi = 0;
        uint32_t kk = k;
        uint32_t j = (uint32_t)key;
        while (true) {
            TileRecord* r = &records[j];
            int tx = (int)(r->tile - ty * tiles_w);                               // signed tile_x (-1 = left of canvas)
            const uint4 own = run_cov[j];
            uint64_t own_lo = (uint64_t)own.x | ((uint64_t)own.y << 32), own_hi = (uint64_t)own.z | ((uint64_t)own.w << 32);
            if (fill == 0) {                                                      // carry-in of this run's tile
                r->cover[0] = (uint32_t)acc_lo; r->cover[1] = (uint32_t)(acc_lo >> 32);
                r->cover[2] = (uint32_t)acc_hi; r->cover[3] = (uint32_t)(acc_hi >> 32);
            }
            acc_lo = swar_add8(acc_lo, own_lo); acc_hi = swar_add8(acc_hi, own_hi);
            bool empty = cover_is_empty(acc_lo, acc_hi, even_odd);
            if (empty) { acc_lo = 0; acc_hi = 0; }                                 // dropped carry (mod.rs:335-339)
            // own (tile, layer) entry
            if (tx >= 0 && tx < (int)tiles_w) {
                uint32_t tile = ty * tiles_w + (uint32_t)tx;
                if (fill == 0) atomicAdd(&tile_count[tile], 1u);
                else entries[tile_off[tile] + atomicAdd(&tile_fill[tile], 1u)] = ((uint64_t)layer << 32) | j;
            }
            // next run of the group
            uint32_t nj = 0; int ntx = (int)tiles_w; bool more = false;
            if (kk + 1 < n_runs) {
                uint64_t nk = sorted_keys[kk + 1];
                if ((uint32_t)(nk >> 32) == group) {
                    more = true; nj = (uint32_t)nk;
                    const TileRecord* nr = &records[nj];
                    ntx = (int)(nr->tile - ty * tiles_w);
                }
            }
            int span_lo = tx + 1 > 0 ? tx + 1 : 0;
            int span_hi = ntx < (int)tiles_w ? ntx : (int)tiles_w;                 // exclusive
            if (!empty && span_lo < span_hi) {
                uint32_t rec = 0;
                if (fill) {
                    rec = atomicAdd(&info->n_spans, 1u);
                    TileRecord sr;
                    sr.cover[0] = (uint32_t)acc_lo; sr.cover[1] = (uint32_t)(acc_lo >> 32);
                    sr.cover[2] = (uint32_t)acc_hi; sr.cover[3] = (uint32_t)(acc_hi >> 32);
                    sr.seg_start = 0; sr.seg_count = 0; sr.layer = layer; sr.tile = ty * tiles_w + (uint32_t)span_lo;
                    rec += record_cap;                                             // span records live after the run records
                    records[rec] = sr;
                }
                for (int t = span_lo; t < span_hi; t++) {
                    uint32_t tile = ty * tiles_w + (uint32_t)t;
                    if (fill == 0) atomicAdd(&tile_count[tile], 1u);
                    else entries[tile_off[tile] + atomicAdd(&tile_fill[tile], 1u)] = ((uint64_t)layer << 32) | rec;
                }
            }
            if (!more) break;
            kk++; j = nj;
        }
    }
}

void launch_carry(hipStream_t s, const uint64_t* sorted_run_keys, uint32_t n_runs, TileRecord* records,
                  const uint4* run_cov, const uint32_t* style_offsets, const uint32_t* style_words, uint32_t n_orders,
                  uint32_t tiles_w, uint32_t tiles_h, uint32_t* tile_count, FrameInfo* info, int fill,
                  const uint32_t* tile_off, uint32_t* tile_fill, uint64_t* entries, uint32_t record_cap) {
    if (n_runs == 0) return;
    uint32_t blocks = (n_runs + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_carry, dim3(blocks), dim3(256), 0, s, sorted_run_keys, n_runs, records, run_cov, style_offsets,
                       style_words, n_orders, tiles_w, tiles_h, tile_count, info, fill, tile_off, tile_fill, entries,
                       record_cap);
}

// ================================================================================================
// the painter: one 256-lane workgroup per 16x16 tile, lane = local_y * 16 + local_x
// ================================================================================================
#define MAXE_LDS 1024

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
#define EF_BAD       0x400u

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
__device__ __forceinline__ void gradient_at(const uint32_t* __restrict__ w, uint32_t fill, uint32_t nstops, float x,
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
    const uint32_t* st = w + 6;
    uint32_t ch[4] = {0, 0, 0, 0};
    bool acc = t <= __uint_as_float(st[4]);
    if (acc) { ch[0] |= st[0]; ch[1] |= st[1]; ch[2] |= st[2]; ch[3] |= st[3]; }
    float start_stop = 0.0f;
    uint32_t sc[4] = {st[0], st[1], st[2], st[3]};
    for (uint32_t k = 1; k < nstops; k++) {
        const uint32_t* q = st + 5 * k;
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
        const uint32_t* q = st + 5 * (nstops - 1);
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

__global__ __launch_bounds__(256) void k_paint(PaintParams P, const uint64_t* __restrict__ sorted,
                                               const uint32_t* __restrict__ tile_off, uint64_t* __restrict__ entries,
                                               const TileRecord* __restrict__ records,
                                               const uint32_t* __restrict__ style_offsets,
                                               const uint32_t* __restrict__ style_words,
                                               const forma_image_t* __restrict__ images,
                                               const uint16_t* __restrict__ texels, uint8_t* __restrict__ image,
                                               FrameInfo* __restrict__ info) {
    __shared__ uint64_t e_key[MAXE_LDS];
    __shared__ uint64_t e_tmp[MAXE_LDS];
    __shared__ uint32_t e_flag[MAXE_LDS];
    __shared__ int cells[256];
    __shared__ uint32_t s_skipped, s_solid, s_solid_bytes;

    // XCD-aware tile mapping: consecutive workgroups land on different XCDs (block b -> XCD b % 8); give
    // each XCD a contiguous band of tiles so a tile row's records/styles stay in one L2.
    const uint32_t T = P.tiles_w * P.tiles_h;
    uint32_t bid = blockIdx.x;
    uint32_t per = (T + 7) / 8;
    uint32_t tile = (bid & 7u) * per + (bid >> 3);
    if (tile >= T || (bid >> 3) >= per) return;
    const uint32_t ty = tile / P.tiles_w, tx = tile - ty * P.tiles_w;
    if (tx < P.crop_x0 || tx >= P.crop_x1 || ty < P.crop_y0 || ty >= P.crop_y1) return;   // print_row :588-592, :525-529

    const int tid = threadIdx.x;
    const int lx = tid & 15, ly = tid >> 4;
    const uint32_t e0 = tile_off[tile], e1 = tile_off[tile + 1];
    const uint32_t ne = e1 - e0;

    // ---- sort this tile's (layer, record) entries by layer: rank sort (keys are unique per tile) --------
    uint64_t* keys = e_key;
    uint32_t* flags = e_flag;
    if (ne > MAXE_LDS) {
        // pathological tile (> MAXE_LDS layers): not supported by the LDS path
        if (tid == 0) atomicOr(&info->error, 2u);
        return;
    }
    for (uint32_t i = tid; i < ne; i += 256) e_tmp[i] = entries[e0 + i];
    __syncthreads();
    for (uint32_t i = tid; i < ne; i += 256) {
        uint64_t k = e_tmp[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < ne; j++) rank += e_tmp[j] < k ? 1u : 0u;      // broadcast LDS reads
        keys[rank] = k;
    }
    __syncthreads();
    // ---- per-entry facts the optimizer passes need ---------------------------------------------------------
    for (uint32_t i = tid; i < ne; i += 256) {
        uint64_t k = keys[i];
        uint32_t layer = (uint32_t)(k >> 32);
        const TileRecord* r = &records[(uint32_t)k];
        uint32_t f = EF_MASK;
        if (layer >= P.n_orders || style_offsets[layer] == FORMA_NONE) f |= EF_BAD;
        else {
            const uint32_t* w = style_words + style_offsets[layer];
            uint32_t h = w[0];
            bool eo = FORMA_STYLE_EVENODD(h);
            if (eo) f |= EF_EVENODD;
            if (r->seg_count) f |= EF_HAS_SEGS;
            else if (cover_full(r->cover, eo)) f |= EF_FULL;
            if (FORMA_STYLE_IS_CLIP(h)) f |= EF_IS_CLIP;
            else {
                if (FORMA_STYLE_CLIPPED(h)) f |= EF_CLIPPED;
                if (FORMA_STYLE_FILL(h) == FORMA_FILL_SOLID) { f |= EF_SOLID; if (__uint_as_float(w[5]) == 1.0f) f |= EF_OPAQUE; }
                if (FORMA_STYLE_BLEND(h) == 0) f |= EF_OVER;
            }
        }
        flags[i] = f;
    }
    __syncthreads();

    const Col clear = {P.clear[0], P.clear[1], P.clear[2], P.clear[3]};
    // ---- optimizer passes (layer_workbench/passes/*.rs), serial over the tile's short layer list ------------
    if (tid == 0) {
        uint32_t skipped = 0, solid = 0;
        Col solid_col = clear;
        if (P.scene_has_clips) {                                   // skip_trivial_clips_pass
            bool has = false, c_full = false, c_used = false; uint32_t c_last = 0, c_i = 0;
            for (uint32_t i = 0; i < ne; i++) {
                uint32_t f = flags[i];
                if (!(f & EF_MASK) || (f & EF_BAD)) continue;
                uint32_t id = (uint32_t)(keys[i] >> 32);
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
        {                                                          // skip_fully_covered_layers_pass
            int first = 0; Col opaque = clear; uint32_t op_i = 0;
            for (uint32_t k = ne; k-- > 0;) {
                uint32_t f = flags[k];
                if (!(f & EF_MASK) || (f & EF_BAD)) continue;
                bool clipped = !(f & EF_IS_CLIP) && (f & EF_CLIPPED) && !(f & EF_SKIPCLIP);
                if (clipped || !(f & EF_FULL)) { if (first == 0) first = 2; }
                else if (!(f & EF_IS_CLIP) && (f & EF_SOLID) && (f & EF_OVER)) {
                    if (f & EF_OPAQUE) {
                        if (first == 0) {
                            first = 1; op_i = k;
                            const uint32_t* w = style_words + style_offsets[(uint32_t)(keys[k] >> 32)];
                            opaque.r = __uint_as_float(w[2]); opaque.g = __uint_as_float(w[3]);
                            opaque.b = __uint_as_float(w[4]); opaque.a = __uint_as_float(w[5]);
                        }
                        skipped = k;
                        break;
                    }
                }
            }
            if (first != 2) {
                Col dst = first == 1 ? opaque : clear;
                bool ok = true;
                for (uint32_t k = skipped; k < ne && ok; k++) {
                    uint32_t f = flags[k];
                    if (!(f & EF_MASK) || (f & EF_BAD)) continue;
                    if (first == 1 && k == op_i) continue;         // the opaque layer itself is the bottom colour
                    if (!(f & EF_IS_CLIP) && (f & EF_SOLID)) {
                        const uint32_t* w = style_words + style_offsets[(uint32_t)(keys[k] >> 32)];
                        Col src = {__uint_as_float(w[2]), __uint_as_float(w[3]), __uint_as_float(w[4]), __uint_as_float(w[5])};
                        dst = sc_blend(FORMA_STYLE_BLEND(w[0]), dst, src);
                    } else ok = false;
                }
                if (ok) { solid = 1; solid_col = dst; }
            }
        }
        s_skipped = skipped; s_solid = solid;
        if (solid) {                                                // to_srgb_bytes(channels.map(color.channel)) :156-162, 690
            float sel[4];
#pragma unroll
            for (int c = 0; c < 4; c++) sel[c] = sel_channel((P.channels >> (8 * c)) & 0xFFu, solid_col.r, solid_col.g, solid_col.b, solid_col.a);
            uint32_t bytes = to_u8_x4(linear_to_srgb(sel[0])) | (to_u8_x4(linear_to_srgb(sel[1])) << 8) |
                             (to_u8_x4(linear_to_srgb(sel[2])) << 16) | (to_u8_x4(sel[3]) << 24);
            s_solid_bytes = bytes;
        }
    }
    __syncthreads();

    const uint32_t px = tx * 16u + (uint32_t)lx, py = ty * 16u + (uint32_t)ly;
    const bool in_image = px < P.width && py < P.height;
    uint32_t* out_px = (uint32_t*)image + (size_t)py * P.stride_px + px;
    if (s_solid) {
        if (in_image) *out_px = s_solid_bytes;
        return;
    }

    // ---- paint (Painter::paint_layer, painter/mod.rs:290-347, one lane per pixel) --------------------------
    float dr = clear.r, dg = clear.g, db = clear.b, da = clear.a;          // Painter::clear :277-288
    bool clip_valid = false; uint32_t clip_last = 0; float clip_mask = 0.0f;
    const float fx = (float)px;                                             // x - 1 + tile_x * TILE_WIDTH  (:325)
    const float fybase = (float)(ty * 16u + ((uint32_t)ly & 8u));          // y * LANES + tile_y * TILE_HEIGHT (:326)
    const int jy = ly & 7;
    const uint32_t skipped = s_skipped;
    for (uint32_t i = skipped; i < ne; i++) {
        const uint32_t f = flags[i];
        if (!(f & EF_MASK) || (f & EF_BAD)) continue;
        const uint64_t k = keys[i];
        const uint32_t layer = (uint32_t)(k >> 32);
        const TileRecord* r = &records[(uint32_t)k];
        const uint32_t* w = style_words + style_offsets[layer];
        const uint32_t h = w[0];
        int carry = (int)(int8_t)(r->cover[ly >> 2] >> ((ly & 3) * 8));
        int A;
        const uint32_t nseg = r->seg_count;
        if (nseg) {
            cells[tid] = 0;
            __syncthreads();
            const uint64_t* sp = sorted + r->seg_start;
            for (uint32_t s = tid; s < nseg; s += 256) {                    // acc_segment :257-271
                uint64_t v = sp[s];
                int cv = seg_cover(v);
                atomicAdd(&cells[seg_ly(v) * 16 + seg_lx(v)], (int)((uint32_t)(seg_dam(v) * cv) << 16) + cv);
            }
            __syncthreads();
            int S = cells[tid];
            int c = (int)(int16_t)(S & 0xFFFF);
            int area = (int)(int16_t)((uint32_t)(S - c) >> 16);
            int inc = c;                                                    // signed cover prefix along x (one DPP row)
            inc += dpp_row_shr(inc, 1); inc += dpp_row_shr(inc, 2); inc += dpp_row_shr(inc, 4); inc += dpp_row_shr(inc, 8);
            int acc = (int)(int8_t)(carry + (inc - c));                     // i8 wrapping column accumulator :343-345
            A = 32 * acc + area;                                            // compute_doubled_areas :388-404
        } else {
            A = 32 * carry;
        }
        if (clip_valid && clip_last < layer) clip_valid = false;            // :298-302
        const float cov = coverage_of(A, (f & EF_EVENODD) != 0);
        if (f & EF_IS_CLIP) {                                               // clip_at :449-464
            if (!clip_valid) { clip_valid = true; clip_last = layer + w[1]; }
            clip_mask = cov;
            continue;
        }
        const bool apply_clip = (f & EF_CLIPPED) && !(f & EF_SKIPCLIP);
        if (cov == 0.0f) continue;                                          // :317-319 (per pixel: blend with 0 is the identity)
        if (apply_clip && !clip_valid) continue;                            // :321-323
        float fill[4];
        const uint32_t ft = FORMA_STYLE_FILL(h);
        if (ft == FORMA_FILL_SOLID) {
            fill[0] = __uint_as_float(w[2]); fill[1] = __uint_as_float(w[3]); fill[2] = __uint_as_float(w[4]); fill[3] = __uint_as_float(w[5]);
        } else if (ft == FORMA_FILL_TEXTURE) {
            texture_at(w, images, texels, fx, fybase + (float)jy, fill);
        } else {
            gradient_at(w, ft, FORMA_STYLE_STOPS(h), fx, fybase, jy, fill);
        }
        float src_a = fill[3] * cov;                                        // blend_at :406-447
        if (apply_clip) src_a *= clip_mask;
        float bl[3];
        blend_rgb(FORMA_STYLE_BLEND(h), dr, dg, db, fill[0], fill[1], fill[2], bl);
        float ida = 1.0f - da, k1 = ida * src_a, isa = 1.0f - src_a, k2 = da * src_a;
        float cr = fmaf(fill[0], k1, bl[0] * k2);
        float cg = fmaf(fill[1], k1, bl[1] * k2);
        float cb = fmaf(fill[2], k1, bl[2] * k2);
        dr = fmaf(dr, isa, cr); dg = fmaf(dg, isa, cg); db = fmaf(db, isa, cb);
        da = fmaf(da, isa, src_a);
    }
    // ---- compute_srgb :466-483 + channel select, straight to the row-major RGBA8 image ----------------------
    if (in_image) {
        float sr = linear_to_srgb(dr), sg = linear_to_srgb(dg), sb = linear_to_srgb(db);
        uint32_t out = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) out |= to_u8_x8(sel_channel((P.channels >> (8 * c)) & 0xFFu, sr, sg, sb, da)) << (8 * c);
        *out_px = out;
    }
}

void launch_paint(hipStream_t s, const PaintParams& p, const uint64_t* sorted, const uint32_t* tile_off,
                  uint64_t* entries, const TileRecord* records, const uint32_t* style_offsets,
                  const uint32_t* style_words, const forma_image_t* images, const uint16_t* texels, uint8_t* image,
                  FrameInfo* info) {
    uint32_t T = p.tiles_w * p.tiles_h;
    if (T == 0) return;
    uint32_t per = (T + 7) / 8;
    hipLaunchKernelGGL(k_paint, dim3(per * 8), dim3(256), 0, s, p, sorted, tile_off, entries, records, style_offsets,
                       style_words, images, texels, image, info);
}
