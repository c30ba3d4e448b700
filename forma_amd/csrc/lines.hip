// lines.hip — stage 1 (flatten map), line preparation, prefix sums and stage 2 (pixel-grid
// intersector) for gfx950.  Built with -ffp-contract=off: the reference (Rust) never contracts
// a*b+c, every fused operation below is an explicit fmaf()/fma() exactly where the reference
// writes `mul_add`.  All kernels are HBM/L2-bound integer + f32/f64 scalar work: no MFMA.
#include "common.h"
#include "lookback.h"
#include <string.h>
#ifdef RAS_PROF
// -DRAS_PROF (tools only): shader-clock stamps at the phase boundaries of k_rasterize (thread 0 of every workgroup)
__device__ unsigned long long g_ras_prof[64][8];
#define RP_STAMP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_ras_prof[blockIdx.x & 63][i], _t - rp_t); rp_t = __builtin_readcyclecounter(); } while (0)
extern "C" int forma_hip_debug_ras_prof(unsigned long long* out8, int reset) {
    static unsigned long long h[64][8];
    if (reset) { for (int c = 0; c < 64; c++) for (int i = 0; i < 8; i++) h[c][i] = 0; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ras_prof), h, sizeof h); }
    int rc = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ras_prof), sizeof h);
    for (int i = 0; i < 8; i++) { out8[i] = 0; for (int c = 0; c < 64; c++) out8[i] += h[c][i]; }
    return rc;
}
#else
#define RP_STAMP(i) do { } while (0)
#endif

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// block-wide exclusive scan helper (256 threads), returns exclusive prefix; *total = block sum.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds /* THREADS/64 + 1 */, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; i++) {
        uint32_t t = lds[i];
        if (i < w) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------------
// u32 prefix sums: reduce -> scan of block sums (single block) -> scan with offsets (in place).
// ------------------------------------------------------------------------------------------------
#define SCAN_THREADS 256
#define SCAN_ITEMS   8
#define SCAN_TILE    (SCAN_THREADS * SCAN_ITEMS)

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t* __restrict__ data, uint32_t n,
                                                              uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds[SCAN_THREADS / 64 + 1];
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint32_t idx = base + i * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += data[idx];
    }
    uint32_t tot;
    block_exclusive_scan<SCAN_THREADS>(s, lds, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block_sums[nb] in place; total -> *d_total
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint32_t* __restrict__ block_sums, DevCount nc, uint32_t per,
                                                          uint32_t* __restrict__ d_total) {
    const uint32_t nb = (dev_count(nc) + per - 1) / per;
    __shared__ uint32_t lds[1024 / 64 + 1];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += 1024) {
        uint32_t idx = base + threadIdx.x;
        uint32_t v = idx < nb ? block_sums[idx] : 0;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<1024>(v, lds, &tot);
        if (idx < nb) block_sums[idx] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && d_total) *d_total = carry;
}

template <bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(uint32_t* __restrict__ data, uint32_t n,
                                                             const uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds[SCAN_THREADS / 64 + 1];
    // blocked arrangement through registers: thread t owns items [t*ITEMS, t*ITEMS+ITEMS)
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = (base + i < n) ? data[base + i] : 0;
        s += v[i];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan<SCAN_THREADS>(s, lds, &tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint32_t out = INCLUSIVE ? ex + v[i] : ex;
        ex += v[i];
        if (base + i < n) data[base + i] = out;
    }
}

// exclusive scan in place of a small array (one workgroup); total -> *d_total
void launch_scan_small_u32(hipStream_t s, uint32_t* data, DevCount n, uint32_t per, uint32_t* d_total) {
    FORMA_LAUNCH(k_scan_block_sums, dim3(1), dim3(1024), 0, s, data, n, per, d_total);
}

size_t scan_tmp_words(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 8; }

static void scan_u32(hipStream_t s, uint32_t* data, size_t n, uint32_t* tmp, uint32_t* d_total, bool inclusive) {
    if (n == 0) {
        if (d_total) hipMemsetAsync(d_total, 0, 4, s);
        return;
    }
    uint32_t nb = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
    FORMA_LAUNCH(k_scan_reduce, dim3(nb), dim3(SCAN_THREADS), 0, s, data, (uint32_t)n, tmp);
    FORMA_LAUNCH(k_scan_block_sums, dim3(1), dim3(1024), 0, s, tmp, DevCount{nullptr, nb}, 1u, d_total);
    if (inclusive)
        FORMA_LAUNCH(k_scan_apply<true>, dim3(nb), dim3(SCAN_THREADS), 0, s, data, (uint32_t)n, tmp);
    else
        FORMA_LAUNCH(k_scan_apply<false>, dim3(nb), dim3(SCAN_THREADS), 0, s, data, (uint32_t)n, tmp);
}
void launch_inclusive_scan_u32(hipStream_t s, uint32_t* data, size_t n, uint32_t* tmp, uint32_t* d_total) {
    scan_u32(s, data, n, tmp, d_total, true);
}
void launch_exclusive_scan_u32(hipStream_t s, uint32_t* data, size_t n, uint32_t* tmp, uint32_t* d_total) {
    scan_u32(s, data, n, tmp, d_total, false);
}

// ------------------------------------------------------------------------------------------------
// line parameters — the map of SegmentBuffer::fill_cpu_view (reference forma/src/segment.rs:298-383).
// Layer lookup is a dense table gather instead of two hash lookups.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2u_sat(float v) {   // Rust `as u32`: saturating, NaN -> 0
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
__device__ __forceinline__ uint32_t integers_between(float a, float b) {   // segment.rs:54-59
    float mn = fminf(a, b), mx = fmaxf(a, b);
    return f2u_sat(ceilf(mx) - floorf(mn) - 1.0f);
}

struct LineP { uint32_t order, len; float x0, y0, dx, dy, a, b, c, d; };

// the arithmetic half of line_params: everything after the loads (so that a caller can have the loads of several lines in
// flight before it looks at any of them)
__device__ __forceinline__ LineP line_params_loaded(const forma_geom_t& g, float p0x, float p0y, float p1x, float p1y, float width,
                                                    float height, float band_lo, float band_hi) {
    LineP L;
    L.order = 0; L.len = 0; L.x0 = L.y0 = L.dx = L.dy = L.a = L.b = L.c = L.d = 0.0f;
    if (g.order == FORMA_NONE) return L;
    if (g.flags & FORMA_GEOM_HAS_XF) {                      // transform_point segment.rs:30-39
        float ax = fmaf(g.xf[0], p0x, fmaf(g.xf[2], p0y, g.xf[4]));
        float ay = fmaf(g.xf[1], p0x, fmaf(g.xf[3], p0y, g.xf[5]));
        float bx = fmaf(g.xf[0], p1x, fmaf(g.xf[2], p1y, g.xf[4]));
        float by = fmaf(g.xf[1], p1x, fmaf(g.xf[3], p1y, g.xf[5]));
        p0x = ax; p0y = ay; p1x = bx; p1y = by;
    }
    // skip_line segment.rs:41-52 (left is NOT culled) + the multi-GPU tile-row band
    const bool skip = (p0y == p1y) || (p0y >= height && p1y >= height) || (p0x >= width && p1x >= width) ||
                      (p0y <= 0.0f && p1y <= 0.0f) || (p0y >= band_hi && p1y >= band_hi) ||
                      (p0y <= band_lo && p1y <= band_lo);
    if (skip) return L;
    const float dx = p1x - p0x, dy = p1y - p0y;
    const float dxr = 1.0f / dx, dyr = 1.0f / dy;
    L.c = dx != 0.0f ? fmaxf((ceilf(p0x) - p0x) * dxr, (floorf(p0x) - p0x) * dxr) : 0.0f;
    L.d = dy != 0.0f ? fmaxf((ceilf(p0y) - p0y) * dyr, (floorf(p0y) - p0y) * dyr) : 0.0f;
    L.order = g.order;
    L.x0 = p0x * 16.0f; L.y0 = p0y * 16.0f; L.dx = dx * 16.0f; L.dy = dy * 16.0f;
    L.a = fabsf(dxr); L.b = fabsf(dyr);
    L.len = integers_between(p0x, p1x) + integers_between(p0y, p1y) + 1u;   // :86-88
    return L;
}
__device__ __forceinline__ LineP line_params(const float* __restrict__ x, const float* __restrict__ y,
                                             const uint32_t* __restrict__ line_slot, uint32_t i,
                                             const forma_geom_t* __restrict__ geoms, uint32_t n_geoms, float width,
                                             float height, float band_lo, float band_hi) {
    const uint32_t slot = line_slot[i];
    if (slot == FORMA_NONE || slot >= n_geoms) {
        LineP L;
        L.order = 0; L.len = 0; L.x0 = L.y0 = L.dx = L.dy = L.a = L.b = L.c = L.d = 0.0f;
        return L;
    }
    const forma_geom_t g = geoms[slot];
    if (g.order == FORMA_NONE) {
        LineP L;
        L.order = 0; L.len = 0; L.x0 = L.y0 = L.dx = L.dy = L.a = L.b = L.c = L.d = 0.0f;
        return L;
    }
    return line_params_loaded(g, x[i], y[i], x[i + 1], y[i + 1], width, height, band_lo, band_hi);
}

// SoA writer: the parity entry point forma_hip_prepare_lines (one thread per line)
__global__ __launch_bounds__(256) void k_prepare_lines(
    const float* __restrict__ x, const float* __restrict__ y, const uint32_t* __restrict__ line_slot, uint32_t n_lines,
    const forma_geom_t* __restrict__ geoms, uint32_t n_geoms, float width, float height, float band_lo, float band_hi,
    uint32_t* __restrict__ orders, float* __restrict__ ox0, float* __restrict__ oy0, float* __restrict__ odx,
    float* __restrict__ ody, float* __restrict__ oa, float* __restrict__ ob, float* __restrict__ oc,
    float* __restrict__ od, uint32_t* __restrict__ lengths) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += gridDim.x * blockDim.x) {
        const LineP L = line_params(x, y, line_slot, i, geoms, n_geoms, width, height, band_lo, band_hi);
        orders[i] = L.order; ox0[i] = L.x0; oy0[i] = L.y0; odx[i] = L.dx; ody[i] = L.dy;
        oa[i] = L.a; ob[i] = L.b; oc[i] = L.c; od[i] = L.d; lengths[i] = L.len;
    }
}

void launch_prepare_lines(hipStream_t s, const float* x, const float* y, const uint32_t* line_slot, uint32_t n_lines,
                          const forma_geom_t* geoms, uint32_t n_geoms, float width, float height, float band_lo,
                          float band_hi, uint32_t* orders, float* x0, float* y0, float* dx, float* dy, float* a,
                          float* b, float* c, float* d, uint32_t* lengths) {
    if (n_lines == 0) return;
    uint32_t blocks = (n_lines + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    FORMA_LAUNCH(k_prepare_lines, dim3(blocks), dim3(256), 0, s, x, y, line_slot, n_lines, geoms, n_geoms, width,
                       height, band_lo, band_hi, orders, x0, y0, dx, dy, a, b, c, d, lengths);
}

// ------------------------------------------------------------------------------------------------
// prepare + scan + compact in ONE launch (frame path).  Replaces the serial `prefix_sum` of the
// reference (segment.rs:90-98) and feeds the PrefixScanIter work split (utils/prefix_scan.rs:30-63):
// a chained scan with decoupled look-back over tiles of PC_TILE lines; status word =
// [flag 2 | line count 30 | segment sum 32] published with one relaxed agent-scope 8-byte store.
// ------------------------------------------------------------------------------------------------
#define PC_THREADS 256
#define PC_IPT     8
#define PC_TILE    (PC_THREADS * PC_IPT)


__device__ __forceinline__ void mark_block_first(uint32_t* __restrict__ block_first, uint32_t bf_cap, uint32_t start,
                                                 uint32_t len, uint32_t c) {
    // every boundary b * RAS_TILE in [start, start + len) is owned by compacted line c
    for (uint32_t b = (start + RAS_TILE - 1) / RAS_TILE; (uint64_t)b * RAS_TILE < (uint64_t)start + len && b < bf_cap; b++)
        block_first[b] = c;
}

// pass A: segment count of every line + per-tile (sum, non-empty count)
// A line costs a chain of dependent loads (slot -> geom entry; the points) behind data-dependent early exits, so the lines of
// one thread are served one after the other: with 8 lines per thread (256-lane workgroups) the kernel took 8 such chains,
// 19 us for 1.6 M lines on a chip that was three-quarters idle.  1024 lanes x 2 lines per tile of the same 2048 lines.
#define LINE_LEN_SAT 0x40000000u      // a line's segment count as the frame path sums it (>= this: the frame cannot be rendered)
#define SEG_SUM_SAT  0x7FFFFFFFu      // where 32-bit stores of segment sums saturate
#define PL_THREADS 1024
#define PL_IPT     (PC_TILE / PL_THREADS)
__global__ __launch_bounds__(PL_THREADS) void k_line_len(LineSource S, uint32_t n_lines, uint32_t* __restrict__ lens,
                                                         uint32_t* __restrict__ tile_sum, uint32_t* __restrict__ tile_cnt, ZeroJobs Z) {
    // the frame's first kernel also clears what later stages expect to be zero (ZeroJobs, common.h): every workgroup a slice
    // of every job, 16 bytes per lane (the buffers are hipMalloc'd and the counts padded by their owners; the tail goes by word)
    for (uint32_t q = 0; q < Z.n; q++) {
        const uint32_t words = Z.words[q], quads = words >> 2;
        const uint32_t per = (quads + gridDim.x - 1) / gridDim.x;
        const uint32_t q0 = blockIdx.x * per, q1 = min(quads, q0 + per);
        uint4* z4 = reinterpret_cast<uint4*>(Z.p[q]);
        for (uint32_t i = q0 + threadIdx.x; i < q1; i += PL_THREADS) z4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (blockIdx.x == 0 && threadIdx.x < (words & 3u)) Z.p[q][(quads << 2) + threadIdx.x] = 0u;
    }
    // Segment counts are summed in 64 bits and SATURATE at SEG_SUM_SAT where they are stored as 32-bit words: geometry that
    // asks for more pixel segments than a device holds (a line from x = -1e20, a transform gone wild) must end as
    // FORMA_E_CAPACITY on the host, not as prefix sums that wrapped around and tables that point anywhere (the reference wraps
    // in release builds and panics in debug ones, segment.rs:86-98).  A single line is cut off at LINE_LEN_SAT: longer ones
    // fail the frame anyway.
    __shared__ uint64_t s_wsum[PL_THREADS / 64];
    __shared__ uint32_t s_wcnt[PL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t base = blockIdx.x * PC_TILE;
    uint64_t sum = 0;
    uint32_t cnt = 0;
    if (S.sums) {
#pragma unroll
        for (int r = 0; r < PL_IPT; r++) {
            const uint32_t i = base + r * PL_THREADS + tid;
            uint32_t len = 0;
            if (i < n_lines) { len = min(S.sums[i] - (i ? S.sums[i - 1] : 0u), LINE_LEN_SAT); lens[i] = len; }
            sum += len; cnt += len ? 1u : 0u;
        }
    } else {
        // two rounds of loads for ALL of the thread's lines (slot + points, then the geom entries), then the arithmetic
        uint32_t slot[PL_IPT];
        float p0x[PL_IPT], p0y[PL_IPT], p1x[PL_IPT], p1y[PL_IPT];
        forma_geom_t g[PL_IPT];
#pragma unroll
        for (int r = 0; r < PL_IPT; r++) {
            const uint32_t i = base + r * PL_THREADS + tid;
            const bool in = i < n_lines;
            slot[r] = in ? S.line_slot[i] : FORMA_NONE;
            p0x[r] = in ? S.x[i] : 0.0f; p0y[r] = in ? S.y[i] : 0.0f; p1x[r] = in ? S.x[i + 1] : 0.0f; p1y[r] = in ? S.y[i + 1] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < PL_IPT; r++) {
            if (slot[r] != FORMA_NONE && slot[r] < S.n_geoms) g[r] = S.geoms[slot[r]];
            else { g[r].order = FORMA_NONE; g[r].flags = 0; }
        }
#pragma unroll
        for (int r = 0; r < PL_IPT; r++) {
            const uint32_t i = base + r * PL_THREADS + tid;
            const uint32_t len = min(line_params_loaded(g[r], p0x[r], p0y[r], p1x[r], p1y[r], S.width, S.height, S.band_lo, S.band_hi).len, LINE_LEN_SAT);
            if (i < n_lines) lens[i] = len;
            sum += len; cnt += len ? 1u : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sum += __shfl_xor(sum, d, 64); cnt += __shfl_xor(cnt, d, 64); }
    if (lane == 0) { s_wsum[w] = sum; s_wcnt[w] = cnt; }
    __syncthreads();
    if (tid == 0) {
        uint64_t ts = 0; uint32_t tc = 0;
        for (int i = 0; i < PL_THREADS / 64; i++) { ts += s_wsum[i]; tc += s_wcnt[i]; }
        tile_sum[blockIdx.x] = (uint32_t)min(ts, (uint64_t)SEG_SUM_SAT); tile_cnt[blockIdx.x] = tc;
    }
}

// pass B: one workgroup, exclusive scan of both per-tile arrays in place; totals -> info
__global__ __launch_bounds__(1024) void k_scan_line_tiles(uint32_t* __restrict__ tile_sum, uint32_t* __restrict__ tile_cnt,
                                                          uint32_t nb, FrameInfo* __restrict__ info) {
    __shared__ uint64_t lds_a[17];
    __shared__ uint32_t lds_b[17];
    uint64_t carry_a = 0;                               // (64-bit sums, saturating where stored: see k_line_len)
    uint32_t carry_b = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t idx = base + threadIdx.x;
        const uint64_t a = idx < nb ? tile_sum[idx] : 0u;
        const uint32_t b = idx < nb ? tile_cnt[idx] : 0u;
        uint64_t ia = a; uint32_t ib = b;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t ta = __shfl_up(ia, d, 64); const uint32_t tb = __shfl_up(ib, d, 64);
            if (lane >= d) { ia += ta; ib += tb; }
        }
        if (lane == 63) { lds_a[w] = ia; lds_b[w] = ib; }
        __syncthreads();
        uint64_t wa = 0, ta = 0; uint32_t wb = 0, tb = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { if (i < w) { wa += lds_a[i]; wb += lds_b[i]; } ta += lds_a[i]; tb += lds_b[i]; }
        if (idx < nb) { tile_sum[idx] = (uint32_t)min(carry_a + wa + ia - a, (uint64_t)SEG_SUM_SAT); tile_cnt[idx] = carry_b + wb + ib - b; }
        carry_a += ta; carry_b += tb;
        __syncthreads();
    }
    if (threadIdx.x == 0) { info->n_segments = (uint32_t)min(carry_a, (uint64_t)SEG_SUM_SAT); info->n_compact = carry_b; }
}

// pass C: compacted line table (cl_idx, cl_start) + block_first; no inter-workgroup dependency
#define PC_SELF_SCAN_TILES 2048u
__global__ __launch_bounds__(PC_THREADS) void k_line_compact(const uint32_t* __restrict__ lens, uint32_t n_lines,
                                                             const uint32_t* __restrict__ tile_sum,
                                                             const uint32_t* __restrict__ tile_cnt,
                                                             uint32_t* __restrict__ cl_idx, uint32_t* __restrict__ cl_start,
                                                             uint32_t* __restrict__ block_first, uint32_t bf_cap,
                                                             FrameInfo* __restrict__ info /* non-null: tile_sum / tile_cnt hold the
                                                             tiles' OWN totals (no k_scan_line_tiles ran) */) {
    __shared__ uint64_t s_wsum[PC_THREADS / 64];
    __shared__ uint32_t s_wcnt[PC_THREADS / 64];
    __shared__ uint64_t s_psum[PC_THREADS / 64];
    __shared__ uint32_t s_pcnt[PC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // With a few hundred tiles every workgroup adds up the totals of the tiles in front of it by itself (a handful of cached
    // loads per lane) — the one-workgroup scan kernel between k_line_len and this one was a launch and 5 us of an idle chip.
    uint64_t ps = 0; uint32_t pc = 0;
    if (info) {
        for (uint32_t i = tid; i < blockIdx.x; i += PC_THREADS) { ps += tile_sum[i]; pc += tile_cnt[i]; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { ps += __shfl_xor(ps, d, 64); pc += __shfl_xor(pc, d, 64); }
        if (lane == 0) { s_psum[w] = ps; s_pcnt[w] = pc; }
    }
    const uint32_t base = blockIdx.x * PC_TILE + tid * PC_IPT;         // 8 consecutive lines per thread
    uint32_t l[PC_IPT], cnt = 0;
    uint64_t sum = 0;                                                   // (64-bit sums, saturating where stored: see k_line_len)
#pragma unroll
    for (int q = 0; q < PC_IPT; q++) { l[q] = base + q < n_lines ? lens[base + q] : 0u; sum += l[q]; cnt += l[q] ? 1u : 0u; }
    uint64_t isum = sum; uint32_t icnt = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t ts = __shfl_up(isum, d, 64); const uint32_t tc = __shfl_up(icnt, d, 64);
        if (lane >= d) { isum += ts; icnt += tc; }
    }
    if (lane == 63) { s_wsum[w] = isum; s_wcnt[w] = icnt; }
    uint64_t bsum = 0; uint32_t bcnt = 0;
    if (!info) { bsum = tile_sum[blockIdx.x]; bcnt = tile_cnt[blockIdx.x]; }      // (in flight across the barrier)
    __syncthreads();
    if (info) {
#pragma unroll
        for (int i = 0; i < PC_THREADS / 64; i++) { bsum += s_psum[i]; bcnt += s_pcnt[i]; }
        bsum = min(bsum, (uint64_t)SEG_SUM_SAT);                        // (what the scan kernel stores)
    }
    const uint32_t tile_c0 = bcnt;                                      // first compacted index of the tile
    if (info && blockIdx.x == gridDim.x - 1 && tid == 0) {              // the frame's totals
        uint64_t ts = bsum; uint32_t tc = bcnt;
        for (int i = 0; i < PC_THREADS / 64; i++) { ts += s_wsum[i]; tc += s_wcnt[i]; }
        info->n_segments = (uint32_t)min(ts, (uint64_t)SEG_SUM_SAT); info->n_compact = tc;
    }
#pragma unroll
    for (int i = 0; i < PC_THREADS / 64; i++) if (i < w) { bsum += s_wsum[i]; bcnt += s_wcnt[i]; }
    // the tile's entries go through LDS so that they leave as full, coalesced rows (a lane's entries are consecutive but the
    // lanes' pieces are ~8 words apart: written directly, every store instruction touched 16 partial cache lines)
    __shared__ uint32_t s_idx[PC_TILE], s_start[PC_TILE];
    uint32_t tile_n = 0;
#pragma unroll
    for (int i = 0; i < PC_THREADS / 64; i++) tile_n += s_wcnt[i];
    uint64_t start = bsum + isum - sum; uint32_t c = bcnt + icnt - cnt;
#pragma unroll
    for (int q = 0; q < PC_IPT; q++) {
        if (l[q]) {
            const uint32_t st = (uint32_t)min(start, (uint64_t)SEG_SUM_SAT);
            s_idx[c - tile_c0] = base + q;
            s_start[c - tile_c0] = st;
            mark_block_first(block_first, bf_cap, st, l[q], c);
            start += l[q]; c++;
        }
    }
    __syncthreads();
    for (uint32_t e = tid; e < tile_n; e += PC_THREADS) { cl_idx[tile_c0 + e] = s_idx[e]; cl_start[tile_c0 + e] = s_start[e]; }
}

size_t prepare_scratch_words(size_t n_lines) { return n_lines + 2 * ((n_lines + PC_TILE - 1) / PC_TILE + 1) + 16; }

void launch_prepare_compact(hipStream_t s, const LineSource& src, uint32_t n_lines, uint32_t* cl_idx, uint32_t* cl_start,
                            uint32_t* block_first, uint32_t bf_cap, uint32_t* scratch, FrameInfo* info, const ZeroJobs* zero) {
    if (n_lines == 0) return;                             // (no kernel, nothing cleared: the caller keeps its memsets)
    ZeroJobs Z;
    memset(&Z, 0, sizeof Z);
    if (zero) Z = *zero;
    // chain-free: count -> scan (one workgroup) -> compact.  (A single-pass chained scan was 2x slower here: with only
    // ~800 tiles the whole grid is resident at once, so every tile walks back through aggregates to tile 0.)
    const uint32_t ntiles = (n_lines + PC_TILE - 1) / PC_TILE;
    uint32_t* lens = scratch;
    uint32_t* tile_sum = scratch + n_lines;
    uint32_t* tile_cnt = tile_sum + ntiles + 1;
    FORMA_LAUNCH(k_line_len, dim3(ntiles), dim3(PL_THREADS), 0, s, src, n_lines, lens, tile_sum, tile_cnt, Z);
    const bool self_scan = ntiles <= PC_SELF_SCAN_TILES;               // (beyond: tiles^2 / 2 loads stop being "a handful")
    if (!self_scan) FORMA_LAUNCH(k_scan_line_tiles, dim3(1), dim3(1024), 0, s, tile_sum, tile_cnt, ntiles, info);
    FORMA_LAUNCH(k_line_compact, dim3(ntiles), dim3(PC_THREADS), 0, s, (const uint32_t*)lens, n_lines,
                       (const uint32_t*)tile_sum, (const uint32_t*)tile_cnt, cl_idx, cl_start, block_first, bf_cap,
                       self_scan ? info : (FrameInfo*)nullptr);
}

// only the per-line pixel-segment counts (the planner of a multi-device context cuts its line shares from their prefix sums)
void launch_line_lengths(hipStream_t s, const LineSource& src, uint32_t n_lines, uint32_t* lens, uint32_t* scratch) {
    if (n_lines == 0) return;
    const uint32_t ntiles = (n_lines + PC_TILE - 1) / PC_TILE;
    uint32_t* tile_sum = scratch + n_lines;
    uint32_t* tile_cnt = tile_sum + ntiles + 1;
    ZeroJobs Z;
    memset(&Z, 0, sizeof Z);
    FORMA_LAUNCH(k_line_len, dim3(ntiles), dim3(PL_THREADS), 0, s, src, n_lines, lens, tile_sum, tile_cnt, Z);
}

__global__ __launch_bounds__(256) void k_block_first(const uint32_t* __restrict__ cl_start, uint32_t n_compact,
                                                     uint32_t n_segments, uint32_t* __restrict__ block_first) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_compact; c += gridDim.x * blockDim.x) {
        const uint32_t start = cl_start[c];
        const uint32_t end = c + 1 < n_compact ? cl_start[c + 1] : n_segments;
        mark_block_first(block_first, 0xFFFFFFFFu, start, end - start, c);
    }
}
void launch_block_first(hipStream_t s, const uint32_t* cl_start, uint32_t n_compact, uint32_t n_segments,
                        uint32_t* block_first) {
    if (n_compact == 0) return;
    uint32_t blocks = (n_compact + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    FORMA_LAUNCH(k_block_first, dim3(blocks), dim3(256), 0, s, cl_start, n_compact, n_segments, block_first);
}

// ------------------------------------------------------------------------------------------------
// rasterize — reference forma/src/cpu/rasterizer.rs:32-159, one thread per pixel segment.
// A workgroup owns RAS_TILE consecutive pixel segments.  The compacted lines that own them are staged
// in LDS once per workgroup together with everything that is constant along a line: the f32
// parameters of segment.rs:341-383 (recomputed from the points: 12 B/line of HBM traffic instead of
// 40) and the three f64 constants of `find` (rasterizer.rs:104-110, one f64 division per LINE instead
// of one per pixel segment).  The flat index -> (line, i) map of PrefixScanIter
// (utils/prefix_scan.rs:30-63) is a binary search over the staged window.
// (Round 6 built the transposed form — a wave walks its 512 segments in eight steps of 64 CONSECUTIVE ones, a segment's line is a
// population count over a 2 048-bit map of line starts plus v_mbcnt, no search and no line walk, what is constant along a line comes
// out of an 80-byte LDS record per step, the sort's digits are counted along the wave — and measured it: 84.4 -> 97.8 us on the 4K
// scene, 50.0 -> 55.2 on the 8K one, 18.9 -> 23.0 at 1080p: five 16-byte LDS reads per segment cost more than the search and the
// walk they replace, which a lane pays once per eight segments.  profiles/r06_experiments.txt, r6x.)
// ------------------------------------------------------------------------------------------------
#ifndef RAS_THREADS
#define RAS_THREADS 256
#endif
#define RAS_PER_THREAD (RAS_TILE / RAS_THREADS)
#ifndef RAS_WIN
#define RAS_WIN 256
#endif

__device__ __forceinline__ float find_term(int i, double a_ab, double b_ab, double cd_ab, float a, float b, float c,
                                           float d) {               // rasterizer.rs:32-61
    float fi = (float)i;
    float ja = isfinite(b) ? (float)ceil(fma(b_ab, (double)fi, -cd_ab)) : fi;
    float jb = isfinite(a) ? (float)ceil(fma(a_ab, (double)fi, cd_ab)) : fi;
    return fminf(fmaf(a, ja, c), fmaf(b, jb, d));
}

__device__ __forceinline__ uint64_t rasterize_one(uint32_t order, float lx0, float ly0, float ldx, float ldy, float a,
                                                  float b, float c, float d, double a_ab, double b_ab, double cd_ab,
                                                  uint32_t seg_i) {
    int i = (int)seg_i - (c != 0.0f ? 1 : 0) - (d != 0.0f ? 1 : 0);                 // rasterizer.rs:63-76
    float t0 = fmaxf(find_term(i, a_ab, b_ab, cd_ab, a, b, c, d), 0.0f);
    float t1 = fminf(find_term(i + 1, a_ab, b_ab, cd_ab, a, b, c, d), 1.0f);
    float x0f = fmaf(t0, ldx, lx0), y0f = fmaf(t0, ldy, ly0);                        // :112-127
    float x1f = fmaf(t1, ldx, lx0), y1f = fmaf(t1, ldy, ly0);
    int x0s = (int)floorf(x0f + 0.5f), x1s = (int)floorf(x1f + 0.5f);                // round :78-80
    int y0s = (int)floorf(y0f + 0.5f), y1s = (int)floorf(y1f + 0.5f);
    int border_x = min(x0s, x1s) >> 4, border_y = min(y0s, y1s) >> 4;
    int tile_x = (int)(int16_t)(border_x >> 4), tile_y = (int)(int16_t)(border_y >> 4);
    uint32_t lx = (uint32_t)border_x & 15u, ly = (uint32_t)border_y & 15u;
    int border = (int)((uint32_t)border_x << 4) + 16;
    uint32_t dam = (uint32_t)(abs(x1s - x0s) + 2 * (border - max(x0s, x1s))) & 0xFFu;   // `as u8`
    int cover = (int)(int8_t)(y1s - y0s);                                                // `as i8`
    // PixelSegment::new, pixel_segment.rs:36-71 (tile + bias wraps as i16, then max(0))
    int ty1 = (int)(int16_t)(tile_y + 1), tx1 = (int)(int16_t)(tile_x + 1);
    uint64_t v = (uint64_t)(ty1 > 0 ? ty1 : 0) & 0x7FFull;
    v = (v << 12) | ((uint64_t)(tx1 > 0 ? tx1 : 0) & 0xFFFull);
    v = (v << 21) | ((uint64_t)order & 0x1FFFFFull);
    v = (v << 4) | lx;
    v = (v << 4) | ly;
    v = (v << 6) | (dam & 0x3Fu);
    v = (v << 6) | ((uint32_t)cover & 0x3Fu);
    return v;
}

// `(x + 0.5).floor() as i32` (rasterizer.rs:78-80).  NOT gfx950's one-instruction v_cvt_rpi_i32_f32: it rounds the exact sum, the
// reference the f32 sum — tools/ubench_rpi.hip: they differ on 0.49999997, on the odd integers of magnitude 2^23..2^24 and on NaN.
__device__ __forceinline__ int round_half_up(float x) { return (int)floorf(x + 0.5f); }
// The same two functions with what is constant along a LINE folded into its staged constants (round 6):
//   * `isfinite(b) ? ceil(fma(B, i, -CD)) : i` — a non-finite b (dy so small that 1 / dy overflows) gets B' = 1, CD' = 0 instead of
//     the select: ceil(fma(1.0, (double)fi, -0.0)) = fi exactly (fi is an integer-valued float; +0 + -0 = +0), likewise for a
//     (vertical lines: 1 / dx = inf — common) with +0.0: two v_cmp_class and two v_cndmask per `find` less;
//   * `i - (c != 0) - (d != 0)` is one staged integer per line.
__device__ __forceinline__ float find_term_k(int i, double a_k, double b_k, double cda_k, double cdb_k, float a, float b, float c, float d) {
    const float fi = (float)i;
    const float ja = (float)ceil(fma(b_k, (double)fi, cdb_k));          // cdb_k = -CD (or -0.0)
    const float jb = (float)ceil(fma(a_k, (double)fi, cda_k));          // cda_k = +CD (or +0.0)
    return fminf(fmaf(a, ja, c), fmaf(b, jb, d));
}
__device__ __forceinline__ uint64_t rasterize_one_k(uint32_t order, float lx0, float ly0, float ldx, float ldy, float a, float b, float c,
                                                    float d, double a_k, double b_k, double cda_k, double cdb_k, int i) {
    float t0 = fmaxf(find_term_k(i, a_k, b_k, cda_k, cdb_k, a, b, c, d), 0.0f);
    float t1 = fminf(find_term_k(i + 1, a_k, b_k, cda_k, cdb_k, a, b, c, d), 1.0f);
    float x0f = fmaf(t0, ldx, lx0), y0f = fmaf(t0, ldy, ly0);                        // :112-127
    float x1f = fmaf(t1, ldx, lx0), y1f = fmaf(t1, ldy, ly0);
    int x0s = round_half_up(x0f), x1s = round_half_up(x1f);                          // round :78-80
    int y0s = round_half_up(y0f), y1s = round_half_up(y1f);
    int border_x = min(x0s, x1s) >> 4, border_y = min(y0s, y1s) >> 4;
    int tile_x = (int)(int16_t)(border_x >> 4), tile_y = (int)(int16_t)(border_y >> 4);
    uint32_t lx = (uint32_t)border_x & 15u, ly = (uint32_t)border_y & 15u;
    int border = (int)((uint32_t)border_x << 4) + 16;
    uint32_t dam = (uint32_t)(abs(x1s - x0s) + 2 * (border - max(x0s, x1s))) & 0xFFu;   // `as u8`
    int cover = (int)(int8_t)(y1s - y0s);                                                // `as i8`
    int ty1 = (int)(int16_t)(tile_y + 1), tx1 = (int)(int16_t)(tile_x + 1);
    uint64_t v = (uint64_t)(ty1 > 0 ? ty1 : 0) & 0x7FFull;
    v = (v << 12) | ((uint64_t)(tx1 > 0 ? tx1 : 0) & 0xFFFull);
    v = (v << 21) | ((uint64_t)order & 0x1FFFFFull);
    v = (v << 4) | lx;
    v = (v << 4) | ly;
    v = (v << 6) | (dam & 0x3Fu);
    v = (v << 6) | ((uint32_t)cover & 0x3Fu);
    return v;
}

__device__ __forceinline__ LineP load_line(const LineSource& S, uint32_t li) {
    if (S.sums) {
        LineP L;
        L.order = S.orders[li]; L.x0 = S.x0[li]; L.y0 = S.y0[li]; L.dx = S.dx[li]; L.dy = S.dy[li];
        L.a = S.a[li]; L.b = S.b[li]; L.c = S.c[li]; L.d = S.d[li]; L.len = 1;
        return L;
    }
    return line_params(S.x, S.y, S.line_slot, li, S.geoms, S.n_geoms, S.width, S.height, S.band_lo, S.band_hi);
}

#ifndef RAS_OCC
#define RAS_OCC 6             // waves per SIMD the kernel is compiled for (<= 80 VGPRs: the three-round staging alone took 82 = five)
#endif
template <bool HIST>         // HIST = false is the plain kernel, instruction for instruction
__global__ __launch_bounds__(RAS_THREADS, RAS_OCC) void k_rasterize(LineSource S, DevCount nc_compact, DevCount nc_segments,
                                                           const uint32_t* __restrict__ cl_idx,
                                                           const uint32_t* __restrict__ cl_start,
                                                           const uint32_t* __restrict__ block_first,
                                                           uint64_t* __restrict__ out, FrameInfo* __restrict__ info,
                                                           int band_row0, int band_row1, uint32_t* __restrict__ wg_masks,
                                                           RasHist RH) {
    __shared__ uint32_t lh[HIST ? RH_MAX_PASSES * SORT_BINS : 1];       // the sort's digit histograms of this workgroup's keys
    __shared__ uint32_t w_start[RAS_WIN + 1];
    __shared__ uint32_t w_order[RAS_WIN];
    __shared__ float w_x0[RAS_WIN], w_y0[RAS_WIN], w_dx[RAS_WIN], w_dy[RAS_WIN];
    __shared__ float w_a[RAS_WIN], w_b[RAS_WIN], w_c[RAS_WIN], w_d[RAS_WIN];
    __shared__ double w_aab[RAS_WIN], w_bab[RAS_WIN], w_cdab[RAS_WIN], w_cdb[RAS_WIN];   // find's constants per line (find_term_k)
    __shared__ int w_ioff[RAS_WIN];                                      // start - ioff: segment k of the line has i = k - w_ibase
    __shared__ uint32_t red[HIST ? 7 : 5][RAS_THREADS / 64];
    const int tid = threadIdx.x;
    // the two counts and the workgroup's two table entries are independent loads: all four in flight before the first test
    // (the table is provisioned for the grid, so the entries exist even for a workgroup past the end)
    const uint32_t n_segments = dev_count(nc_segments), n_compact = dev_count(nc_compact);
    const uint32_t bf_lo = block_first[blockIdx.x], bf_hi = block_first[blockIdx.x + 1];
    const uint32_t nblocks = (n_segments + RAS_TILE - 1) / RAS_TILE;
    const uint32_t k0 = blockIdx.x * RAS_TILE;
    if (k0 >= n_segments) return;                                       // the grid was sized for the bound
    const uint32_t k1 = min(k0 + RAS_TILE, n_segments);                   // exclusive
#ifdef RAS_PROF
    unsigned long long rp_t = __builtin_readcyclecounter();
    if (tid == 0) atomicAdd(&g_ras_prof[blockIdx.x & 63][7], 1ull);
#endif
    const uint32_t lo = bf_lo;
    const uint32_t hi = blockIdx.x + 1 < nblocks ? bf_hi : n_compact - 1;   // inclusive
    uint32_t k_or = 0, k_or_hi = 0, k_and = 0xFFFFFFFFu, k_and_hi = 0xFFFFFFFFu, unsorted = 0;
    const uint32_t kt = k0 + tid * RAS_PER_THREAD;                      // this thread's first segment
    uint64_t vout[RAS_PER_THREAD];
#pragma unroll
    for (int q = 0; q < RAS_PER_THREAD; q++) vout[q] = 0;
    if (HIST) for (uint32_t i = tid; i < RH.n_passes * SORT_BINS; i += RAS_THREADS) lh[i] = 0;   // (barriers: the staging loop's)

    for (uint32_t c0 = lo; c0 <= hi; c0 += RAS_WIN) {
        const uint32_t cnt = min((uint32_t)RAS_WIN, hi - c0 + 1);
        if (c0 != lo) __syncthreads();
        for (uint32_t j = tid; j < cnt; j += RAS_THREADS) {
            const uint32_t cidx = c0 + j;
            // The line's loads in THREE rounds, each issued whole before anything waits (round 6): its compacted entry and its
            // predecessor's; both slots and the four coordinates; the geom entry and the predecessor's order.  As `load_line` wrote
            // it — slot, then the order test, then the points, then (after the arithmetic) the predecessor's index -> slot ->
            // order — a workgroup's staging was SEVEN dependent round trips, 17 k of its 30 k clocks.  A compacted line owns
            // pixel segments, so its slot and its geom entry are valid: nothing here needs an early exit.
            LineP L;
            uint32_t prev_order = 0;
            if (S.sums) {
                L = load_line(S, cl_idx[cidx]);
                if (cidx > 0) prev_order = S.orders[cl_idx[cidx - 1]];
            } else {
                const uint32_t li = cl_idx[cidx], lp = cl_idx[cidx > 0 ? cidx - 1 : cidx];
                const uint32_t slot = S.line_slot[li], slotp = S.line_slot[lp];
                const float p0x = S.x[li], p1x = S.x[li + 1], p0y = S.y[li], p1y = S.y[li + 1];
                forma_geom_t g;
                g.order = FORMA_NONE; g.flags = 0;
                if (slot < S.n_geoms) g = S.geoms[slot];
                if (slotp < S.n_geoms) prev_order = S.geoms[slotp].order;
                L = line_params_loaded(g, p0x, p0y, p1x, p1y, S.width, S.height, S.band_lo, S.band_hi);
            }
            w_start[j] = cl_start[cidx];
            w_order[j] = L.order; w_x0[j] = L.x0; w_y0[j] = L.y0; w_dx[j] = L.dx; w_dy[j] = L.dy;
            w_a[j] = L.a; w_b[j] = L.b; w_c[j] = L.c; w_d[j] = L.d;
            const double sum_recip = 1.0 / ((double)L.a + (double)L.b);          // rasterizer.rs:104-110
            const double cd = ((double)L.c - (double)L.d) * sum_recip;
            const bool fa = isfinite(L.a), fb = isfinite(L.b);                   // (find: `isfinite(a) ? ceil(..) : i`, folded: find_term_k)
            w_aab[j] = fa ? (double)L.a * sum_recip : 1.0; w_bab[j] = fb ? (double)L.b * sum_recip : 1.0;
            w_cdab[j] = fa ? cd : 0.0; w_cdb[j] = fb ? -cd : -0.0;
            w_ioff[j] = (int)cl_start[cidx] + (L.c != 0.0f ? 1 : 0) + (L.d != 0.0f ? 1 : 0);   // i = k - this (rasterizer.rs:63-76)
            // is the stream non-decreasing in layer?  (lets the sort skip the layer digits)
            if (cidx > 0 && prev_order > L.order) unsorted = 1;
        }
        if (tid == 0) w_start[cnt] = (c0 + cnt < n_compact) ? cl_start[c0 + cnt] : n_segments;
        __syncthreads();
        RP_STAMP(0);                                                    // lines staged (load chain + f64 constants + barrier)
        const uint32_t ka = max(k0, w_start[0]), kb = min(k1, w_start[cnt]);   // this chunk's share of the tile
        // this thread's 8 consecutive segments that fall into the chunk: ONE binary search, then walk the lines
        const uint32_t t_lo = max(kt, ka), t_hi = min(kt + RAS_PER_THREAD, kb);
        if (t_lo < t_hi) {
            uint32_t a = 0, b = cnt;                       // last j in [0, cnt) with w_start[j] <= t_lo
            while (b - a > 1) {
                const uint32_t mid = (a + b) >> 1;
                if (w_start[mid] <= t_lo) a = mid; else b = mid;
            }
            RP_STAMP(1);                                                // binary search
            uint32_t l_next = w_start[a + 1], l_order = w_order[a];
            float l_x0 = w_x0[a], l_y0 = w_y0[a], l_dx = w_dx[a], l_dy = w_dy[a], l_a = w_a[a], l_b = w_b[a], l_c = w_c[a], l_d = w_d[a];
            double l_aab = w_aab[a], l_bab = w_bab[a], l_cdab = w_cdab[a], l_cdb = w_cdb[a];
            int l_ioff = w_ioff[a];
#pragma unroll
            for (int q = 0; q < RAS_PER_THREAD; q++) {
                const uint32_t k = kt + q;
                if (k < t_lo || k >= t_hi) continue;
                if (k >= l_next) {                         // next line (a compacted line owns >= 1 segment: never more than one step)
                    a++;
                    l_next = w_start[a + 1]; l_order = w_order[a];
                    l_x0 = w_x0[a]; l_y0 = w_y0[a]; l_dx = w_dx[a]; l_dy = w_dy[a];
                    l_a = w_a[a]; l_b = w_b[a]; l_c = w_c[a]; l_d = w_d[a];
                    l_aab = w_aab[a]; l_bab = w_bab[a]; l_cdab = w_cdab[a]; l_cdb = w_cdb[a]; l_ioff = w_ioff[a];
                }
                uint64_t v = rasterize_one_k(l_order, l_x0, l_y0, l_dx, l_dy, l_a, l_b, l_c, l_d, l_aab, l_bab, l_cdab, l_cdb, (int)k - l_ioff);
                if (band_row1 > 0) {
                    int ty = seg_tile_y(v);
                    if (ty < band_row0 || ty >= band_row1) v &= 0x001FFFFFFFFFFFFFull;   // -> tile row -1: never painted
                }
                vout[q] = v;
                uint32_t klo = (uint32_t)(v >> 20), khi = (uint32_t)(v >> 52);
                k_or |= klo; k_or_hi |= khi; k_and &= klo; k_and_hi &= khi;
            }
        }
    }
    RP_STAMP(2);                                                        // the 8 pixel segments of the lane
    uint32_t min_x = 0xFFFFu, max_x = 0, min_y = 0xFFFFu, max_y = 0;   // tile fields of this thread's keys (HIST, RH.track_range)
    if (HIST) {
        // The digits of the sort's passes, counted where the keys are made.  A thread's 8 keys are consecutive segments of a
        // line (or of neighbouring lines): mostly one tile, so a run-length pass over the registers leaves one or two LDS
        // atomics per digit and thread.  (Electing one lane per digit of the wavefront first — the 64 lanes mostly add to the
        // same word — was built and cost 44 us instead of 13: scalar loops of readlane / ballot, against LDS atomics that
        // were never the limit.  Round 6: ONE walk that cuts the keys into runs of equal `v >> 20` and counts every run in every
        // pass — fewer VALU on paper, a loop over the passes behind each of eight divergent branches in the machine: 86.3 -> 89.5
        // us on the 4K scene, 47.5 -> 53.2 on the 8K one.)
        const uint32_t nv = kt < k1 ? min((uint32_t)RAS_PER_THREAD, k1 - kt) : 0u;
        __syncthreads();                                                // (lh cleared, also for a workgroup whose loop ran dry)
        if (nv) {
            if (RH.track_range) {                                       // (only where a tile field relative to its minimum can save a pass)
#pragma unroll
                for (int q = 0; q < RAS_PER_THREAD; q++) {
                    if ((uint32_t)q < nv) {
                        const uint32_t hw = (uint32_t)(vout[q] >> 32), tx = (hw >> 9) & 0xFFFu, ty = hw >> 21;
                        min_x = min(min_x, tx); max_x = max(max_x, tx); min_y = min(min_y, ty); max_y = max(max_y, ty);
                    }
                }
            }
            // (a digit in the key's high word — every pass of a layer-sorted frame — is one 32-bit shift, chosen by a UNIFORM branch
            //  instead of a 64-bit shift, a 32-bit one and a select per key; the runs are cut by predicated adds, no branch per key:
            //  ~6 VALU + 4 SALU per key and pass where the nested form took ~6 + 10 — 1 us per frame, `r9d`)
            for (uint32_t p = 0; p < RH.n_passes; p++) {
                const uint32_t sh = RH.shift[p], mk = RH.mask[p], bs = RH.bias[p];
                uint32_t* h = lh + p * SORT_BINS;
                uint32_t d[RAS_PER_THREAD];
                if (sh >= 32) {
#pragma unroll
                    for (int q = 0; q < RAS_PER_THREAD; q++) d[q] = (((uint32_t)(vout[q] >> 32) >> (sh - 32)) - bs) & mk;
                } else {
#pragma unroll
                    for (int q = 0; q < RAS_PER_THREAD; q++) d[q] = ((uint32_t)(vout[q] >> sh) - bs) & mk;
                }
                uint32_t c = 1;
#pragma unroll
                for (int q = 0; q + 1 < RAS_PER_THREAD; q++) {
                    const bool last = (uint32_t)q + 1 >= nv;                       // (never for a lane with all its keys)
                    const bool brk = last || d[q] != d[q + 1];
                    if (brk && (uint32_t)q < nv) atomicAdd(&h[d[q]], c);
                    c = brk ? 1u : c + 1u;
                }
                if (nv == RAS_PER_THREAD) atomicAdd(&h[d[RAS_PER_THREAD - 1]], c);
            }
        }
    }
    // 64 contiguous bytes per thread: 16-byte stores (the tile base is a multiple of 2048 segments)
#pragma unroll
    for (int q = 0; q < RAS_PER_THREAD; q += 2) {
        const uint32_t k = kt + q;
        if (k + 1 < k1) *reinterpret_cast<ulonglong2*>(out + k) = make_ulonglong2(vout[q], vout[q + 1]);
        else if (k < k1) out[k] = vout[q];
    }
    // block reduction of the varying-bit masks -> a few atomics per block
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        k_or |= __shfl_xor(k_or, d, 64); k_or_hi |= __shfl_xor(k_or_hi, d, 64);
        k_and &= __shfl_xor(k_and, d, 64); k_and_hi &= __shfl_xor(k_and_hi, d, 64);
        unsorted |= __shfl_xor(unsorted, d, 64);
    }
    if (HIST && RH.track_range) {                                       // (uniform)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            min_x = min(min_x, (uint32_t)__shfl_xor(min_x, d, 64)); max_x = max(max_x, (uint32_t)__shfl_xor(max_x, d, 64));
            min_y = min(min_y, (uint32_t)__shfl_xor(min_y, d, 64)); max_y = max(max_y, (uint32_t)__shfl_xor(max_y, d, 64));
        }
    }
    const int w = tid >> 6;
    __syncthreads();
    if ((tid & 63) == 0) {
        red[0][w] = k_or; red[1][w] = k_or_hi; red[2][w] = k_and; red[3][w] = k_and_hi; red[4][w] = unsorted;
        if (HIST) { red[5][w] = min_x | (max_x << 16); red[6][w] = min_y | (max_y << 16); }
    }
    __syncthreads();
    if (HIST) {
        // flush: the non-empty bins into this workgroup's copy of the sort's histograms (a workgroup's keys cover a handful of tiles)
        uint32_t* mine = RH.hist + (size_t)(blockIdx.x % HS_COPIES) * (SORT_MAX_PASSES * SORT_BINS);
        for (uint32_t i = tid; i < RH.n_passes * SORT_BINS; i += RAS_THREADS) {
            const uint32_t v = lh[i];
            if (v) atomicAdd(&mine[i], v);
        }
    }
    if (tid == 0) {
        uint32_t o = 0, oh = 0, a = 0xFFFFFFFFu, ah = 0xFFFFFFFFu, u = 0;
        for (int i = 0; i < RAS_THREADS / 64; i++) { o |= red[0][i]; oh |= red[1][i]; a &= red[2][i]; ah &= red[3][i]; u |= red[4][i]; }
        if (HIST) {
            uint32_t lo_x = 0xFFFFu, hi_x = 0, lo_y = 0xFFFFu, hi_y = 0;
            for (int i = 0; i < RAS_THREADS / 64; i++) {
                lo_x = min(lo_x, red[5][i] & 0xFFFFu); hi_x = max(hi_x, red[5][i] >> 16);
                lo_y = min(lo_y, red[6][i] & 0xFFFFu); hi_y = max(hi_y, red[6][i] >> 16);
            }
            uint32_t* mr = wg_masks + (size_t)blockIdx.x * 8;
            mr[5] = lo_x | (hi_x << 16); mr[6] = lo_y | (hi_y << 16);
            // a digit taken relative to a field's minimum was planned from the PREVIOUS frame's span: a key outside it voids
            // the frame (the host runs it again with plain digits) — what k_sort_hist checks when it takes the histograms
            for (uint32_t p = 0; p < RH.n_passes; p++) {
                if (!RH.fmask[p]) continue;
                const bool is_x = RH.shift[p] == 41;
                const uint32_t lo_v = is_x ? lo_x : lo_y, hi_v = is_x ? hi_x : hi_y;
                if (lo_v < RH.bias[p] || hi_v - RH.bias[p] > RH.mask[p]) info->plan_bad = 1u;
            }
        }
        // one record per workgroup, combined by k_reduce_masks: five plain stores.  (Until round 2 every workgroup read the
        // frame's masks through the caches and issued atomics when it added information: thread 0 of each of the 6 700
        // workgroups then finished thousands of clocks after its workgroup's stores, holding the workgroup's LDS and
        // registers; reading the words earlier, or reducing before the stores, turned it into a same-address storm.)
        uint32_t* m = wg_masks + (size_t)blockIdx.x * 8;
        m[0] = o; m[1] = oh; m[2] = a; m[3] = ah; m[4] = u;
    }
    RP_STAMP(3);                                                        // stores + mask reduction
}

// the per-workgroup key masks of k_rasterize -> info->{key_or, key_or_hi, key_and, key_and_hi, layer_unsorted}
__global__ __launch_bounds__(1024) void k_reduce_masks(const uint32_t* __restrict__ wg_masks, DevCount nc_segments,
                                                       FrameInfo* __restrict__ info) {
    __shared__ uint32_t red[5][16];
    const uint32_t nb = (dev_count(nc_segments) + RAS_TILE - 1) / RAS_TILE;
    uint32_t o = 0, oh = 0, a = 0xFFFFFFFFu, ah = 0xFFFFFFFFu, u = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 8 * 1024) {                    // 8 records per thread in flight: one round trip for 8192 blocks
        uint4 m[8]; uint32_t mu[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t b = b0 + k * 1024 + threadIdx.x;
            m[k] = b < nb ? *reinterpret_cast<const uint4*>(wg_masks + (size_t)b * 8) : make_uint4(0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu);
            mu[k] = b < nb ? wg_masks[(size_t)b * 8 + 4] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { o |= m[k].x; oh |= m[k].y; a &= m[k].z; ah &= m[k].w; u |= mu[k]; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        o |= __shfl_xor(o, d, 64); oh |= __shfl_xor(oh, d, 64); a &= __shfl_xor(a, d, 64); ah &= __shfl_xor(ah, d, 64);
        u |= __shfl_xor(u, d, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = o; red[1][w] = oh; red[2][w] = a; red[3][w] = ah; red[4][w] = u; }
    __syncthreads();
    if (threadIdx.x == 0 && nb) {
        for (int i = 1; i < 16; i++) { o |= red[0][i]; oh |= red[1][i]; a &= red[2][i]; ah &= red[3][i]; u |= red[4][i]; }
        info->key_or = o; info->key_or_hi = oh; info->key_and = a; info->key_and_hi = ah; info->layer_unsorted = u;
    }
}

void launch_rasterize(hipStream_t s, const LineSource& src, DevCount n_compact, DevCount n_segments,
                      const uint32_t* cl_idx, const uint32_t* cl_start, const uint32_t* block_first, uint64_t* out,
                      FrameInfo* info, int band_row0, int band_row1, uint32_t* wg_masks, bool reduce_now, const RasHist* hist) {
    if (n_segments.bound == 0 || n_compact.bound == 0) return;
    uint32_t blocks = (n_segments.bound + RAS_TILE - 1) / RAS_TILE;
    RasHist RH;
    memset(&RH, 0, sizeof RH);
    if (hist && hist->hist)
        FORMA_LAUNCH(k_rasterize<true>, dim3(blocks), dim3(RAS_THREADS), 0, s, src, n_compact, n_segments, cl_idx, cl_start,
                           block_first, out, info, band_row0, band_row1, wg_masks, *hist);
    else
        FORMA_LAUNCH(k_rasterize<false>, dim3(blocks), dim3(RAS_THREADS), 0, s, src, n_compact, n_segments, cl_idx, cl_start,
                           block_first, out, info, band_row0, band_row1, wg_masks, RH);
    if (reduce_now) FORMA_LAUNCH(k_reduce_masks, dim3(1), dim3(1024), 0, s, (const uint32_t*)wg_masks, n_segments, info);
}

// ------------------------------------------------------------------------------------------------
// flatten — the parallel map of Primitives::into_segments (reference forma/src/path.rs:487-534),
// one thread per output point.  The sequential curvature bookkeeping (path.rs:252-445) is done by
// the host and arrives as work items.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lerpf(float t, float a, float b) { return fmaf(t, b, fmaf(-t, a, a)); }   // path.rs:44-46
__device__ __forceinline__ float inv_curvature(float k) {                                                   // path.rs:53-56
    const float C = 0.39f;
    return k * (1.0f - C + sqrtf(fmaf(k * k, 0.25f, C * C)));
}

__global__ __launch_bounds__(256) void k_flatten(forma_flatten_tables_t t, float* __restrict__ out_x,
                                                 float* __restrict__ out_y) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < t.n_points; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t cmd = t.point_commands[i];
        float px, py;
        if ((cmd & 0x7F800000u) == 0x7F800000u) {                       // PointCommand NaN-box, path.rs:137-168
            uint32_t si = cmd & 0x3FFFFFu;
            if ((cmd & 0x80000000u) == 0) { px = t.sp0x[si]; py = t.sp0y[si]; }      // Start
            else { px = t.sp2x[si]; py = t.sp2y[si]; }                                // End
        } else {
            float incr = __uint_as_float(cmd);
            uint32_t qi = t.quad_indices[i], pi = t.point_indices[i];
            uint32_t spline_i = t.partial_spline[qi];
            float prev = 0.0f;
            if (qi >= 1 && t.partial_spline[qi - 1] == spline_i) prev = t.partial_curv[qi - 1];
            float ratio = fmaf(incr, (float)pi, -prev) * t.curvatures_recip[qi];
            float xx = inv_curvature(fmaf(ratio, t.dk[qi], t.k0[qi]));
            float tt = (xx - t.x0[qi]) * t.dx_recip[qi];
            if (tt < 0.0f) tt = 0.0f;
            if (tt > 1.0f) tt = 1.0f;
            size_t i0 = 3 * (size_t)qi, i1 = i0 + 1, i2 = i0 + 2;       // eval_quad path.rs:447-471
            float w = lerpf(tt, lerpf(tt, t.qw[i0], t.qw[i1]), lerpf(tt, t.qw[i1], t.qw[i2]));
            float wr = 1.0f / w;
            px = lerpf(tt, lerpf(tt, t.qx[i0], t.qx[i1]), lerpf(tt, t.qx[i1], t.qx[i2])) * wr;
            py = lerpf(tt, lerpf(tt, t.qy[i0], t.qy[i1]), lerpf(tt, t.qy[i1], t.qy[i2])) * wr;
        }
        out_x[i] = px; out_y[i] = py;
    }
}

void launch_flatten(hipStream_t s, const forma_flatten_tables_t* dev_tables, float* out_x, float* out_y) {
    if (dev_tables->n_points == 0) return;
    size_t blocks = (dev_tables->n_points + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    FORMA_LAUNCH(k_flatten, dim3((uint32_t)blocks), dim3(256), 0, s, *dev_tables, out_x, out_y);
}
