// sort.hip — stage 3: device-side stable LSB radix sort of the packed u64 pixel segments
// (replaces `segments.par_crumsort()`, reference forma/src/cpu/rasterizer.rs:161-164, ordering =
// `PixelSegment::cmp` on bits 20..63, cpu/pixel_segment.rs:161-171).
//
// Design (gfx950, HBM-bound byte shuffling, no MFMA):
//   * digit plan: the caller passes the mask of key bits that vary at all (computed for free by the
//     rasterizer); digits are packed greedily over LIVE bits only, so constant fields (the unused
//     high bits of tile_y / tile_x / layer) cost nothing.  If the rasterizer stream is already
//     non-decreasing in layer (the common case: geometry inserted in paint order) the layer bits are
//     dropped from the plan too — a stable sort by (tile_y, tile_x) alone then yields exactly the
//     stream a stable sort by the full 44-bit key would.
//   * one up-front histogram kernel for ALL passes (1 read of the keys), then ONE kernel per digit
//     pass: chained scan with decoupled look-back (tile status words = {flag, count} in one u32, the
//     "R2 granule" form of the MI355X guide: the data is the flag, relaxed agent-scope accesses), so
//     a pass moves 8 B in + 8 B out per key and nothing else.
//   * ranking inside a tile: wavefront match-any from __ballot()s of the digit bits, v_mbcnt for the
//     lane prefix, per-wave LDS digit counters; keys are staged in LDS so that every digit run leaves
//     the CU as one contiguous, coalesced store.
//   * persistent workgroups pull tiles with an atomic ticket: a tile only ever waits for tiles with
//     a smaller ticket, which are already running — forward progress does not depend on dispatch
//     order (guide: "HIP promises nothing about dispatch order").
#include "common.h"
#include "lookback.h"
#include "radix_rank.h"
#include <string.h>
#include <hip/hip_ext.h>

#ifndef OS_THREADS
#define OS_THREADS 1024
#endif
#define OS_WAVES   (OS_THREADS / 64)
// keys per lane: 16 (16384 keys per tile -> 128 KiB of LDS staging, 1 workgroup = 16 waves per CU); the 512-bin pass takes 14 —
// its counters and 9-bit ranks cost registers, at 16 keys it spilled (C4: 59.2 -> 53.6 us per pass; 256 bins: 62.1 -> 63.4)
#ifndef OS_KPT9
#define OS_KPT9 14
#endif
__host__ __device__ constexpr int os_kpt(int bits) { return bits == 9 ? OS_KPT9 : 16; }
// ... and fewer where n keys are not 256 such tiles: a pass of 2.7 M keys (1080p) was 164 tiles on 256 CUs, each the full
// latency of a 16 384-key tile — load, rank, stage, look-back, scatter, about half of it proportional to the keys per lane.
// With 8 / 12 keys per lane the same keys are ONE round of shorter tiles on (nearly) every CU.  0: the default of the digit width.
#ifndef OS_SMALL_TILES
#define OS_SMALL_TILES 1
#endif
static inline int os_kpt_small(size_t n) {
#ifdef OS_FORCE_KPT
    return OS_FORCE_KPT;                                  // (tools: that many keys per lane — even — whatever the stream's size)
#endif
    if (!OS_SMALL_TILES) return 0;
    if (n <= (size_t)256 * OS_THREADS * 8) return 8;
    if (n <= (size_t)256 * OS_THREADS * 12) return 12;
    return 0;
}
// provisioning (row stride of the status words): the most tiles any pass of n keys can have
static inline size_t os_tile_min(size_t n) { const int k = os_kpt_small(n); return (size_t)OS_THREADS * (size_t)(k ? k : (OS_KPT9 < 16 ? OS_KPT9 : 16)); }
#define ST_VALMASK 0x3FFFFFFFu
#ifndef OS_LBW
#define OS_LBW     8
#endif
//                                                  // predecessor tiles examined per look-back probe

// ------------------------------------------------------------------------------------------------
// up-front histograms of every planned digit: hist[p * 256 + d].  Per-lane run-length compression
// (consecutive keys of a lane mostly share their tile digits) keeps the LDS atomics rare.
// ------------------------------------------------------------------------------------------------
#define HS_THREADS 256
#define HS_KPT     16
#define HS_TILE    (HS_THREADS * HS_KPT)

// logical index of the rank-major concatenation of the received buckets -> position in the receive buffer (ChunkedSrc)
struct ChunkMap { uint32_t n; uint32_t pre[FORMA_MAX_RANKS]; uint32_t gap[FORMA_MAX_RANKS]; uint32_t total, over; };
__device__ __forceinline__ ChunkMap load_chunk_map(const ChunkedSrc& C, uint32_t bound) {
    ChunkMap M;
    M.n = C.n_chunks; M.total = 0; M.over = 0;
    uint32_t prev = 0;
#pragma unroll
    for (int q = 0; q < FORMA_MAX_RANKS; q++) {                          // (uniform: <= 8 scalar loads)
        uint32_t c = 0;
        if (q < (int)C.n_chunks) {
            const uint64_t h = C.buckets[(size_t)q * ((size_t)C.capacity + 1) + C.capacity];   // bucket header {count | overflow << 32}
            const uint32_t c0 = (uint32_t)h;
            M.over |= (uint32_t)(h >> 32) | (c0 > C.capacity ? 1u : 0u);
            c = c0 < C.capacity ? c0 : C.capacity;
        }
        M.pre[q] = M.total;                                             // first logical index of bucket q
        M.gap[q] = q ? C.capacity + 1u - prev : 0u;                     // unused slots (and the header) between bucket q - 1's data and bucket q
        M.total += c; prev = c;
    }
    if (M.total > bound) M.total = bound;
    return M;
}
__device__ __forceinline__ uint32_t chunk_phys(const ChunkMap& M, uint32_t idx) {
    uint32_t p = idx;
#pragma unroll
    for (int q = 1; q < FORMA_MAX_RANKS; q++) {
        if (q >= (int)M.n) break;                                       // uniform: costs nothing for the buckets that do not exist
        p += idx >= M.pre[q] ? M.gap[q] : 0u;
    }
    return p;
}

template <bool CHUNKED>      // CHUNKED = false is the plain kernel, instruction for instruction
__global__ __launch_bounds__(HS_THREADS) void k_sort_hist(const uint64_t* __restrict__ in, DevCount nc, SortPlan plan,
                                                          uint32_t* __restrict__ hist, ChunkedSrc C, FrameInfo* __restrict__ info) {
    constexpr bool chunked = CHUNKED;
    ChunkMap M;
    if (chunked) {
        M = load_chunk_map(C, nc.bound);
        if (blockIdx.x == 0 && threadIdx.x == 0) { info->n_segments = M.total; if (M.over) { info->exchange_overflow = 1u; info->plan_bad = 1u; } }
    }
    const uint32_t n = chunked ? M.total : dev_count(nc);
    uint32_t k_or = 0, k_or_hi = 0, k_and = 0xFFFFFFFFu, k_and_hi = 0xFFFFFFFFu, unsorted = 0;
    uint32_t nmin_x = 0, max_x = 0, nmin_y = 0, max_y = 0;             // tile fields seen by this thread: ~min, max
    __shared__ uint32_t lh[SORT_MAX_PASSES * SORT_BINS];
    const int P = plan.n_passes;
    for (int i = threadIdx.x; i < P * SORT_BINS; i += HS_THREADS) lh[i] = 0;
    __syncthreads();
    const uint32_t ntiles = (n + HS_TILE - 1) / HS_TILE;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t base = t * HS_TILE + threadIdx.x;
        uint64_t k[HS_KPT];
#pragma unroll
        for (int j = 0; j < HS_KPT; j++) {
            uint32_t idx = base + j * HS_THREADS;
            k[j] = idx < n ? in[chunked ? chunk_phys(M, idx) : idx] : 0ull;
        }
        // the span of the tile fields (KeyRange: next frame's plan; this frame's plan is checked against it below) — the kernel
        // waits for HBM, the ALUs are idle
#pragma unroll
        for (int j = 0; j < HS_KPT; j++) {
            if (base + j * HS_THREADS < n) {
                const uint32_t hi = (uint32_t)(k[j] >> 32), tx = (hi >> 9) & 0xFFFu, ty = hi >> 21;
                nmin_x = max(nmin_x, ~tx); max_x = max(max_x, tx); nmin_y = max(nmin_y, ~ty); max_y = max(max_y, ty);
            }
        }
        if (chunked) {
            // the facts the sort plan is verified with (what k_gather_chunks computed when the stream was materialised):
            // OR / AND of the key bits, and whether the logical stream is non-decreasing in layer — also across buckets
            uint32_t front[HS_KPT];                                    // lane 0 of a wave: layer of the key in front of each of its rows
#pragma unroll
            for (int j = 0; j < HS_KPT; j++) {
                const uint32_t idx = base + j * HS_THREADS;
                front[j] = ((threadIdx.x & 63) == 0 && idx > 0 && idx < n) ? seg_layer(in[chunk_phys(M, idx - 1)]) : 0u;
            }
#pragma unroll
            for (int j = 0; j < HS_KPT; j++) {
                const uint32_t idx = base + j * HS_THREADS;
                const uint32_t klo = (uint32_t)(k[j] >> 20), khi = (uint32_t)(k[j] >> 52);
                const uint32_t layer = klo & 0x1FFFFFu;
                uint32_t pl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)layer, 0x138, 0xF, 0xF, true);   // wave_shr:1
                if ((threadIdx.x & 63) == 0) pl = front[j];
                if (idx < n) {
                    k_or |= klo; k_or_hi |= khi; k_and &= klo; k_and_hi &= khi;
                    unsorted |= pl > layer ? 1u : 0u;
                }
            }
        }
        // Lane-adjacent run-length compression, then one LDS atomic per run: consecutive segments mostly share
        // their tile digits, and 64 lanes adding to one LDS word would serialise.
        const int lane = threadIdx.x & 63;
        for (int p = 0; p < P; p++) {
            const int sh = plan.shift[p];
            const uint32_t mk = plan.mask[p], bs = plan.bias[p];
#pragma unroll
            for (int j = 0; j < HS_KPT; j++) {
                const bool valid = base + j * HS_THREADS < n;
                const uint32_t d = valid ? (((uint32_t)(k[j] >> sh) - bs) & mk) : 0xFFFFFFFFu;
                const uint32_t dprev = __shfl_up(d, 1, 64);
                const bool head = lane == 0 || d != dprev;
                const uint64_t heads = __ballot(head);
                if (head && valid) {
                    const uint64_t above = lane == 63 ? 0ull : (heads >> (lane + 1));
                    const uint32_t len = above ? (uint32_t)__builtin_ctzll(above) + 1u : (uint32_t)(64 - lane);
                    atomicAdd(&lh[p * SORT_BINS + d], len);
                }
            }
        }
    }
    __syncthreads();
    // the flush is up to 256 x P global atomics per workgroup on the same few hundred words: HS_COPIES private copies (one
    // per workgroup residue class, i.e. per XCD under round-robin dispatch) cut the same-address queue by that factor;
    // k_onesweep adds the copies up when it scans the histogram
    uint32_t* mine = hist + (size_t)(blockIdx.x % HS_COPIES) * (SORT_MAX_PASSES * SORT_BINS);
    for (int i = threadIdx.x; i < P * SORT_BINS; i += HS_THREADS) {
        uint32_t v = lh[i];
        if (v) atomicAdd(&mine[i], v);
    }
    {
        // tile-field spans: ONE 16-byte record per workgroup behind the histograms, plain stores (neutral if it saw no key);
        // k_runs_count folds the records into FrameInfo::tile_range.  (Four atomicMax per wave into 16 copies looked cheaper
        // and cost 130 us: the copies shared two cache lines, and atomics on one line serialise.)
        __shared__ uint32_t rr[4][HS_THREADS / 64];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            nmin_x = max(nmin_x, (uint32_t)__shfl_xor(nmin_x, d, 64)); max_x = max(max_x, (uint32_t)__shfl_xor(max_x, d, 64));
            nmin_y = max(nmin_y, (uint32_t)__shfl_xor(nmin_y, d, 64)); max_y = max(max_y, (uint32_t)__shfl_xor(max_y, d, 64));
        }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { rr[0][w] = nmin_x; rr[1][w] = max_x; rr[2][w] = nmin_y; rr[3][w] = max_y; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int q = 1; q < HS_THREADS / 64; q++) { nmin_x = max(nmin_x, rr[0][q]); max_x = max(max_x, rr[1][q]); nmin_y = max(nmin_y, rr[2][q]); max_y = max(max_y, rr[3][q]); }
            uint32_t* r = hist + (size_t)HS_COPIES * SORT_MAX_PASSES * SORT_BINS + 16 + (size_t)blockIdx.x * 4;
            *reinterpret_cast<uint4*>(r) = make_uint4(nmin_x, max_x, nmin_y, max_y);
            // a digit taken relative to a field's minimum (SortPlan::bias) was planned from the PREVIOUS frame's span: a key
            // outside it voids the frame (the host runs it again with plain digits)
            if (max_x | max_y | nmin_x | nmin_y) {
                for (int p = 0; p < P; p++) {
                    if (!plan.fmask[p]) continue;
                    const bool is_x = plan.shift[p] == 41;
                    const uint32_t lo = is_x ? ~nmin_x : ~nmin_y, hi_v = is_x ? max_x : max_y;
                    if (lo < plan.bias[p] || hi_v - plan.bias[p] > plan.mask[p]) info->plan_bad = 1u;
                }
            }
        }
    }
    if (chunked && C.mask_records) {                                    // one record per workgroup (neutral if it saw no key)
        __shared__ uint32_t red[5][HS_THREADS / 64];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            k_or |= __shfl_xor(k_or, d, 64); k_or_hi |= __shfl_xor(k_or_hi, d, 64);
            k_and &= __shfl_xor(k_and, d, 64); k_and_hi &= __shfl_xor(k_and_hi, d, 64);
            unsorted |= __shfl_xor(unsorted, d, 64);
        }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][w] = k_or; red[1][w] = k_or_hi; red[2][w] = k_and; red[3][w] = k_and_hi; red[4][w] = unsorted; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t o = 0, oh = 0, a = 0xFFFFFFFFu, ah = 0xFFFFFFFFu, u = 0;
            for (int q = 0; q < HS_THREADS / 64; q++) { o |= red[0][q]; oh |= red[1][q]; a &= red[2][q]; ah &= red[3][q]; u |= red[4][q]; }
            uint32_t* m = C.mask_records + (size_t)blockIdx.x * 8;
            m[0] = o; m[1] = oh; m[2] = a; m[3] = ah; m[4] = u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// one digit pass
// ------------------------------------------------------------------------------------------------
// The barriers of k_onesweep's tile loop.  An LDS-only barrier (s_waitcnt lgkmcnt(0) + s_barrier, -DOS_LDS_BARRIER) that lets the
// scatter stores drain under the next tile was measured: no difference (64.6 vs 65.0 us per pass), so the plain one stays.
__device__ __forceinline__ void lds_barrier() {
#ifdef OS_LDS_BARRIER
    __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// inclusive scan along the wave by DPP row shifts and broadcasts: six VALU instructions, no LDS crossbar (the __shfl_up form
// kept six lane addresses alive across the tile loop — spilled at the register cap)
__device__ __forceinline__ uint32_t os_wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}
// exclusive scan of `a` over the first RADIX threads of the block; every thread must call
template <int RADIX>
__device__ __forceinline__ void scan_excl(uint32_t& a, uint32_t* lds /* 8 */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t ia = os_wave_incl_scan(a);
    if (RADIX > 64) {
        if (lane == 63 && w < RADIX / 64) lds[w] = ia;
        lds_barrier();
        uint32_t ba = 0;
#pragma unroll
        for (int i = 0; i < RADIX / 64; i++) if (i < w) ba += lds[i];
        lds_barrier();
        a = ba + ia - a;
    } else {
        a = ia - a;
    }
}

#ifdef SORT_PROF
// -DSORT_PROF (tools only): shader-clock stamps at the phase boundaries of k_onesweep (thread 0), summed over all tiles
__device__ unsigned long long g_sort_prof[16];
#define SP_STAMP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g_sort_prof[i], _t - sp_t); sp_t = _t; } while (0)
extern "C" int forma_hip_debug_sort_prof(unsigned long long* out16, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sort_prof), z, sizeof z); }
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sort_prof), 16 * 8);
}
#else
#define SP_STAMP(i) do { } while (0)
#endif

// The per-wave digit counters of a tile.  256 bins (and 16): one 32-bit word per (wave, digit).  512 bins: 16-bit halves, two
// digits per word — a wave holds 1024 keys and a tile 16384, so neither a count nor a tile-local start overflows a half, and
// the 16 KB footprint (next to the 128 KB of staged keys) is the same.  The ranking adds with a 32-bit LDS atomic on the word
// that holds the digit's half; the totals / bases phase reads and writes the halves as 16-bit LDS accesses.
template <int BITS> struct WaveCounters;
template <int BITS> struct WaveCounters {
    static constexpr int RADIX = 1 << BITS;
    uint32_t c[OS_WAVES][RADIX];
    __device__ __forceinline__ uint32_t add_and_read(int w, uint32_t dg, bool leader, uint32_t cnt) {
        if (leader) atomicAdd(&c[w][dg], cnt);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // compiler barrier: keep the read after the add
        return c[w][dg];
    }
    __device__ __forceinline__ uint32_t get(int w, uint32_t dg) const { return c[w][dg]; }
    __device__ __forceinline__ void set(int w, uint32_t dg, uint32_t v) { c[w][dg] = v; }
    __device__ __forceinline__ void clear_wave(int w, int lane) {
#pragma unroll
        for (int i = lane; i < RADIX; i += 64) c[w][i] = 0;
    }
};
template <> struct WaveCounters<9> {
    static constexpr int RADIX = 512;
    uint32_t c[OS_WAVES][RADIX / 2];
    __device__ __forceinline__ uint32_t add_and_read(int w, uint32_t dg, bool leader, uint32_t cnt) {
        const uint32_t sh = (dg & 1u) * 16u;
        if (leader) atomicAdd(&c[w][dg >> 1], cnt << sh);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        return (c[w][dg >> 1] >> sh) & 0xFFFFu;
    }
    __device__ __forceinline__ uint32_t get(int w, uint32_t dg) const { return reinterpret_cast<const uint16_t*>(&c[w][0])[dg]; }
    __device__ __forceinline__ void set(int w, uint32_t dg, uint32_t v) { reinterpret_cast<uint16_t*>(&c[w][0])[dg] = (uint16_t)v; }
    __device__ __forceinline__ void clear_wave(int w, int lane) {
#pragma unroll
        for (int i = lane; i < RADIX / 2; i += 64) c[w][i] = 0;
    }
};

// HI: the digit lies in the key's high word (shift >= 32: the tile_y / tile_x fields, i.e. every pass of a layer-sorted
// frame) — one 32-bit bit-field extract instead of a 64-bit shift.
template <bool HI>
__device__ __forceinline__ uint32_t key_digit(uint64_t key, int shift, uint32_t dmask, uint32_t bias) {
    if (HI) return (((uint32_t)(key >> 32) >> (shift - 32)) - bias) & dmask;
    return ((uint32_t)(key >> shift) - bias) & dmask;
}

template <int BITS, bool CHUNKED, bool HI, int KPT_ = 0>
__global__ __launch_bounds__(OS_THREADS, 4) void k_onesweep(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                         DevCount nc, int shift, uint32_t dmask, uint32_t bias,
                                                         const uint32_t* __restrict__ ghist /* this pass, SORT_BINS per copy */,
                                                         uint32_t* __restrict__ status /* [ntiles][RADIX] */,
                                                         uint32_t* __restrict__ ticket, uint32_t* __restrict__ err,
                                                         ChunkedSrc C /* first pass of a chunked stream, else n_chunks = 0 */) {
    constexpr int RADIX = 1 << BITS;
    constexpr int KPT = KPT_ ? KPT_ : os_kpt(BITS), TILE = OS_THREADS * KPT;  // keys per lane, keys per tile
    static_assert(KPT % 2 == 0, "the 16-bit ranks of a lane are packed two per register");
    const uint32_t n = dev_count(nc);                       // (chunked: k_sort_hist has published the total)
    constexpr bool chunked = CHUNKED;                       // (one bucket at offset 0 is launched as a plain stream)
    ChunkMap M;
    if (chunked) M = load_chunk_map(C, nc.bound);
    __shared__ uint64_t staged[TILE];
    __shared__ WaveCounters<BITS> whist;
    __shared__ uint32_t s_gdelta[RADIX];
    __shared__ uint32_t s_scan[16];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t ntiles = (n + TILE - 1) / TILE;

    // a wave clears its own counters: here once, then right after it has staged a tile's keys (nobody else reads a wave's
    // row between the barrier in front of the staging and the one behind the look-back) — the tile loop has no clearing
    // phase and one barrier less than it had
    whist.clear_wave(w, lane);
    // global digit starts = exclusive scan of this pass's histogram (identical in every block)
    uint32_t gstart = 0;
    {
        uint32_t g = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int c = 0; c < HS_COPIES; c++) g += ghist[(size_t)c * (SORT_MAX_PASSES * SORT_BINS) + tid];
        }
        scan_excl<RADIX>(g, s_scan);
        gstart = g;
    }

#ifdef SORT_PROF
    unsigned long long sp_t = __builtin_readcyclecounter();
#endif
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    lds_barrier();
    while (true) {
        SP_STAMP(7);                                        // scatter + end barrier (and the prologue, once)
        const uint32_t tile = s_tile;
        if (tile >= ntiles) break;
        SP_STAMP(0);
#ifdef SORT_PROF
        if (tid == 0) atomicAdd(&g_sort_prof[15], 1ull);
#endif
        const uint32_t bbase = tile * TILE;
        const uint32_t wbase = bbase + w * (64 * KPT);

        uint64_t keys[KPT];
        uint32_t rnk[KPT / 2];                          // 16-bit ranks, two per register
        // what pads the stream's last tile must land behind every real key of the tile: the LAST digit — with a biased digit
        // that is not the all-ones key ((~0 >> shift) - bias lands in the middle of the bins)
        const uint64_t pad = (uint64_t)(dmask + bias) << shift;
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            uint32_t idx = wbase + j * 64 + lane;
            keys[j] = idx < n ? in[chunked ? chunk_phys(M, idx) : idx] : pad;      // padding: last digit, last in stream order, never written
        }
#ifdef SORT_PROF
        if (keys[KPT - 1] == 0x123456789ull) atomicAdd(&g_sort_prof[14], 1ull);      // forces the loads to have landed
        SP_STAMP(1);                                        // key loads
#endif
        // ---- stable rank of every key among the same-digit keys of its wave ----------------------------------
        // per row: the lowest peer lane adds the class size to the wave's LDS digit counter, then every lane reads
        // the counter back (LDS operations of one wave retire in order): rank = counter - class size + lanes below.
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            const uint32_t dg = key_digit<HI>(keys[j], shift, dmask, bias);
            // (peeling the few distinct digits of a row off leader by leader — readlane, compare, mbcnt per class — was
            //  measured slower than this fixed 8-ballot form: 71 vs 61 us per pass, the dependent scalar chain does not pipeline;
            //  so was a run-structured rank — head lanes add their run length with a returning LDS atomic, the others fetch it
            //  with ds_bpermute, rows where a digit owns two runs fall back to this form: ~25 VALU per row instead of ~47, and
            //  64.4 against 62.2 us per pass — NOTES.md)
            uint32_t mlo, mhi;
            match_any<BITS>(dg, mlo, mhi);
            const uint32_t below = lanes_below(mlo, mhi);
            const uint32_t cnt = (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
            const uint32_t after = whist.add_and_read(w, dg, below == 0, cnt);
            const uint32_t r = after - cnt + below;
            if (j & 1) rnk[j >> 1] |= r << 16; else rnk[j >> 1] = r;
        }
        lds_barrier();
        SP_STAMP(2);                                        // rank + barrier
        // ---- digit totals of the tile, per-wave bases, look-back ----------------------------------------
        uint32_t tot = 0, lbase = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int i = 0; i < OS_WAVES; i++) tot += whist.get(i, tid);
            if (tile > 0) lb_st32(&status[(size_t)tile * RADIX + tid], (LB_AGG << 30) | tot);
        }
        {
            uint32_t a = tot;
            scan_excl<RADIX>(a, s_scan);
            lbase = a;
        }
        if (tid < RADIX) {
            uint32_t acc = lbase;
#pragma unroll
            for (int i = 0; i < OS_WAVES; i++) { uint32_t c = whist.get(i, tid); whist.set(i, tid, acc); acc += c; }
        }
        lds_barrier();
        SP_STAMP(3);                                        // totals, scan, bases
        // ---- stage in digit order (needs only tile-local positions), BEFORE the look-back: the key registers die here
        //      and the staging of waves 4..7 overlaps the global round trips of the look-back lanes ---------------------
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            const uint32_t dg = key_digit<HI>(keys[j], shift, dmask, bias);
            staged[whist.get(w, dg) + ((rnk[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu)] = keys[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // compiler barrier; this wave's LDS operations retire in order
        whist.clear_wave(w, lane);
        SP_STAMP(4);                                        // staging (issue)
        // ---- look-back: one lane per digit walks its own chain of status words, OS_LBW predecessors per probe.
        //      (A one-at-a-time walk moves ~1 tile per L2 round trip, which is about the rate at which tiles retire:
        //      the window of aggregate-only predecessors then never drains and the walk becomes the pass.)
        uint32_t excl = 0;
        if (tid < RADIX && tile > 0) {
            int p = (int)tile - 1;
            uint32_t spins = 0;
            bool done = false;
            while (!done) {
                uint32_t v[OS_LBW];
#pragma unroll
                for (int i = 0; i < OS_LBW; i++)
                    v[i] = p - i >= 0 ? lb_ld32(&status[(size_t)(p - i) * RADIX + tid]) : (LB_PREFIX << 30);
                int used = 0;
#pragma unroll
                for (int i = 0; i < OS_LBW; i++) {
                    if (!done && used == i) {
                        const uint32_t f = v[i] >> 30;
                        if (f != 0) { excl += v[i] & ST_VALMASK; used = i + 1; if (f == LB_PREFIX) done = true; }
                    }
                }
                p -= used;
                if (used == 0) {
                    if (++spins > LB_SPIN_LIMIT) { atomicOr(err, 4u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        if (tid < RADIX) {
            lb_st32(&status[(size_t)tile * RADIX + tid], (LB_PREFIX << 30) | (excl + tot));
            s_gdelta[tid] = gstart + excl - lbase;
        }
        lds_barrier();
        SP_STAMP(5);                                        // look-back + barrier
        // the next ticket's round trip (~1 us) runs under the scatter; every thread read s_tile barriers ago
        if (tid == OS_THREADS - 1) s_tile = atomicAdd(ticket, 1u);
        // ---- coalesced stores: every digit run leaves the CU as one contiguous piece --------------------------------
        const uint32_t nvalid = min((uint32_t)TILE, n - bbase);
        // (the compiler otherwise hoists the sixteen `j * 1024 + tid` out of the tile loop and, at the register cap, SPILLS them:
        //  29 scratch round trips per tile in the 512-bin instantiation — recomputing one OR per key is free)
        uint32_t tid_here = (uint32_t)tid;
        asm volatile("" : "+v"(tid_here));
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            uint32_t i = j * OS_THREADS + tid_here;
            if (i < nvalid) {
                uint64_t key = staged[i];
                uint32_t dg = key_digit<HI>(key, shift, dmask, bias);
                out[i + s_gdelta[dg]] = key;
            }
        }
        lds_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static SortPlan plan_with_width(uint64_t live_mask, int lo_bit, int hi_bit, int db) {
    SortPlan p;
    p.n_passes = 0;
    int b = lo_bit;
    while (b < hi_bit && p.n_passes < SORT_MAX_PASSES) {
        if (!((live_mask >> b) & 1ull)) { b++; continue; }
        int width = hi_bit - b < db ? hi_bit - b : db;
        // trim dead bits at the top of the window
        while (width > 1 && !((live_mask >> (b + width - 1)) & 1ull)) width--;
        p.shift[p.n_passes] = b;
        p.mask[p.n_passes] = (1u << width) - 1u;
        p.bias[p.n_passes] = 0; p.fmask[p.n_passes] = 0;
        p.n_passes++;
        b += width;
    }
    return p;
}

// digit_bits: 4 = the 16-bin instantiation throughout; 8 = 256 bins; 0 = 256 bins, or 512 where nine-bit digits save a
// whole pass (17 or 18 live key bits — an 8192 x 8192 canvas has 9 + 9 — sort in two passes instead of three)
SortPlan make_sort_plan(uint64_t live_mask, int lo_bit, int hi_bit, int digit_bits) {
    if (digit_bits == 4 || digit_bits == 9) return plan_with_width(live_mask, lo_bit, hi_bit, digit_bits);
    const SortPlan p8 = plan_with_width(live_mask, lo_bit, hi_bit, 8);
    if (digit_bits == 8) return p8;
    const SortPlan p9 = plan_with_width(live_mask, lo_bit, hi_bit, 9);
    return p9.n_passes < p8.n_passes ? p9 : p8;
}

static int bit_length(uint32_t v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

SortPlan make_segment_sort_plan(uint64_t live44, bool layer_sorted, int digit_bits, const KeyRange* range, bool* biased) {
    if (biased) *biased = false;
    uint64_t live = live44;
    if (layer_sorted) live &= ~0x1FFFFFull;               // a stream already non-decreasing in layer: stable sort by tile alone
    const SortPlan plain = make_sort_plan(live << 20, 20, 64, digit_bits);
    if (!range || !range->valid || digit_bits == 4) return plain;
    if (range->min_x > range->max_x || range->min_y > range->max_y) return plain;
    // layer digits as before (bits 20..40), then ONE digit per tile field: (field - min) in bit_length(max - min) bits
    const int bx = bit_length(range->max_x - range->min_x), by = bit_length(range->max_y - range->min_y);
    const int widest = digit_bits == 8 ? 8 : 9;
    if (bx > widest || by > widest) return plain;
    SortPlan p = make_sort_plan((live & 0x1FFFFFull) << 20, 20, 41, digit_bits);
    if (p.n_passes + 2 > SORT_MAX_PASSES) return plain;
    if (bx > 0) { const int i = p.n_passes++; p.shift[i] = 41; p.mask[i] = (1u << bx) - 1u; p.bias[i] = range->min_x; p.fmask[i] = 0xFFFu; }
    if (by > 0) { const int i = p.n_passes++; p.shift[i] = 53; p.mask[i] = (1u << by) - 1u; p.bias[i] = range->min_y; p.fmask[i] = 0x7FFu; }
    if (p.n_passes >= plain.n_passes) return plain;       // (only where it saves a pass: plain digits do not depend on the span)
    if (biased) *biased = true;
    return p;
}

uint32_t sort_hist_blocks(size_t n) {
    uint32_t hb = (uint32_t)((n + HS_TILE - 1) / HS_TILE);
    return hb > 2048 ? 2048u : hb;                        // few workgroups: the final flush is bins x passes global atomics each
}

// [histograms] [tickets: 16 words] [tile-field spans: one 4-word record per k_sort_hist workgroup, <= 2048]
static inline size_t sort_fixed_words() { return (size_t)HS_COPIES * SORT_MAX_PASSES * SORT_BINS + 16 + 2048 * 4; }
const uint32_t* sort_range_words(const uint32_t* scratch) { return scratch + (size_t)HS_COPIES * SORT_MAX_PASSES * SORT_BINS + 16; }
size_t sort_scratch_words(size_t n) {
    size_t ntiles = (n + os_tile_min(n) - 1) / os_tile_min(n);
    // [hist: HS_COPIES x MAX_PASSES x SORT_BINS] [tickets: MAX_PASSES, pad to 16] [tile-field spans: 2048 x 4] [status: MAX_PASSES * ntiles * SORT_BINS]
    return sort_fixed_words() + (size_t)SORT_MAX_PASSES * (ntiles + 1) * SORT_BINS;
}
// the words of the scratch a sort of `n` keys with this plan expects to be zero when it starts (histograms, tickets, the
// status rows of the passes that run): launch_radix_sort clears them itself unless the caller says an earlier kernel of the
// frame already did (api.cpp folds the clearing into the frame's first kernel)
size_t sort_zero_words(size_t n, const SortPlan& plan) {
    const size_t ntiles = (n + os_tile_min(n) - 1) / os_tile_min(n);
    return sort_fixed_words() + (size_t)plan.n_passes * ntiles * SORT_BINS;
}

RasHist make_ras_hist(const SortPlan& plan, uint32_t* sort_scratch) {
    RasHist R;
    memset(&R, 0, sizeof R);
    if (plan.n_passes < 1 || plan.n_passes > RH_MAX_PASSES) return R;
    for (int p = 0; p < plan.n_passes; p++) {
        if (plan.mask[p] >= SORT_BINS) return R;
        R.shift[p] = (uint32_t)plan.shift[p]; R.mask[p] = plan.mask[p]; R.bias[p] = plan.bias[p]; R.fmask[p] = plan.fmask[p];
    }
    R.n_passes = (uint32_t)plan.n_passes;
    R.hist = sort_scratch;                                   // [HS_COPIES][SORT_MAX_PASSES][SORT_BINS] at the head of the scratch
    return R;
}

const uint64_t* launch_radix_sort(hipStream_t s, const uint64_t* in, uint64_t* a, uint64_t* b, DevCount nc,
                                  const SortPlan& plan, int digit_bits, uint32_t* scratch, uint32_t* err,
                                  const ChunkedSrc* chunked, FrameInfo* info,
                                  bool scratch_is_zero, bool hist_ready, uint32_t max_workgroups) {
    const size_t n = nc.bound;                        // provisioning (grid, scratch); the kernels use the device count
    if (n <= 1 || plan.n_passes == 0) return in;
    const uint32_t ntiles = (uint32_t)((n + os_tile_min(n) - 1) / os_tile_min(n));     // (row stride of the status words: sort_zero_words)
    const int small_kpt = digit_bits == 4 ? 0 : os_kpt_small(n);
    uint32_t* hist = scratch;
    uint32_t* tickets = scratch + (size_t)HS_COPIES * SORT_MAX_PASSES * SORT_BINS;
    uint32_t* status = scratch + sort_fixed_words();
    const int P = plan.n_passes;
    // zero hist + tickets + the status words of the passes that run (re-initialised every call)
    if (!scratch_is_zero) (void)hipMemsetAsync(scratch, 0, sort_zero_words(n, plan) * 4, s);
    const uint32_t hb = sort_hist_blocks(n);
    ChunkedSrc C0;
    memset(&C0, 0, sizeof C0);
    const ChunkedSrc C = chunked ? *chunked : C0;
    if (chunked) FORMA_LAUNCH(k_sort_hist<true>, dim3(hb), dim3(HS_THREADS), 0, s, in, nc, plan, hist, C, info);
    else if (!(hist_ready && scratch_is_zero)) FORMA_LAUNCH(k_sort_hist<false>, dim3(hb), dim3(HS_THREADS), 0, s, in, nc, plan, hist, C, info);
    uint32_t cap = 512u * 512u / OS_THREADS;              // persistent: 16 waves per CU
    if (max_workgroups && max_workgroups < cap) cap = max_workgroups;   // (frames in flight: leave CUs to the other frames' kernels)
    const uint64_t* src = in;
    uint64_t* dst = a;
    for (int p = 0; p < P; p++) {
        uint32_t* st = status + (size_t)p * ntiles * SORT_BINS;
        const bool ch = p == 0 && C.n_chunks > 1;                      // only the first pass reads the received buckets in place
        const bool hi = plan.shift[p] >= 32;
        // (on a timed frame the launch carries its own events, FORMA_LAUNCH: the dispatch's start and end timestamps)
        const int bits = digit_bits == 4 ? 4 : (plan.mask[p] > 255u ? 9 : 8);
        const uint32_t tile = (uint32_t)OS_THREADS * (uint32_t)(small_kpt ? small_kpt : os_kpt(bits));
        const uint32_t ptiles = (uint32_t)((n + tile - 1) / tile);
        const uint32_t grid = ptiles < cap ? ptiles : cap;
#define OS_LAUNCH(B, CH, HI_, K_) FORMA_LAUNCH((k_onesweep<B, CH, HI_, K_>), dim3(grid), dim3(OS_THREADS), 0, s, src, dst, nc, \
                                           plan.shift[p], plan.mask[p], plan.bias[p], (const uint32_t*)(hist + p * SORT_BINS), st, tickets + p, err, ch ? C : C0)
#define OS_LAUNCH_K(B, K_) do { if (ch) { if (hi) OS_LAUNCH(B, true, true, K_); else OS_LAUNCH(B, true, false, K_); } \
                                else { if (hi) OS_LAUNCH(B, false, true, K_); else OS_LAUNCH(B, false, false, K_); } } while (0)
#ifdef OS_FORCE_KPT
#define OS_LAUNCH_B(B) OS_LAUNCH_K(B, OS_FORCE_KPT)                   // (tools: any even count)
#else
#define OS_LAUNCH_B(B) do { if (small_kpt == 8) OS_LAUNCH_K(B, 8); else if (small_kpt == 12) OS_LAUNCH_K(B, 12); else OS_LAUNCH_K(B, 0); } while (0)
#endif
        if (digit_bits == 4) OS_LAUNCH_K(4, 0);
        else if (plan.mask[p] > 255u) OS_LAUNCH_B(9);
        else OS_LAUNCH_B(8);
#undef OS_LAUNCH_B
#undef OS_LAUNCH_K
#undef OS_LAUNCH
        src = dst;
        dst = (dst == a) ? b : a;
    }
    return src;
}
