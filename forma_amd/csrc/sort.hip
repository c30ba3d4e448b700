// sort.hip — stage 3: device-side stable LSB radix sort of the packed u64 pixel segments
// (replaces `segments.par_crumsort()`, reference forma/src/cpu/rasterizer.rs:161-164, ordering =
// `PixelSegment::cmp` on bits 20..63, cpu/pixel_segment.rs:161-171).
//
// 4-bit digits.  Per pass: k_hist (per-block digit counts via wavefront ballots) ->
// k_scan_counts (one block, digit-major exclusive scan) -> k_scatter (ballot/popcount ranking inside
// each wave, LDS staging so every digit run leaves the CU as one contiguous, coalesced store).
// Constant digits (single-bin histograms) are skipped: the caller passes the mask of key bits that
// vary at all, computed for free by the rasterizer.  HBM-bound byte shuffling: no MFMA.
#include "common.h"

#define SORT_THREADS 256
#define SORT_WAVES   (SORT_THREADS / 64)
#define SORT_KPT     16                              // keys per lane
#define SORT_WKEYS   (64 * SORT_KPT)                 // keys per wave
#define SORT_TILE    (SORT_THREADS * SORT_KPT)       // keys per block (4096 -> 32 KiB of LDS staging)
#define RADIX_BITS   4
#define RADIX        16

// mask of lanes (among `valid`) whose 4-bit digit equals dv, from the four digit-bit ballots
__device__ __forceinline__ uint64_t digit_peers(uint64_t valid, uint64_t b0, uint64_t b1, uint64_t b2, uint64_t b3,
                                                uint32_t dv) {
    uint64_t m = valid;
    m &= (dv & 1u) ? b0 : ~b0;
    m &= (dv & 2u) ? b1 : ~b1;
    m &= (dv & 4u) ? b2 : ~b2;
    m &= (dv & 8u) ? b3 : ~b3;
    return m;
}

__global__ __launch_bounds__(SORT_THREADS) void k_hist(const uint64_t* __restrict__ in, uint32_t n, int shift,
                                                       uint32_t nblocks, uint32_t* __restrict__ counts) {
    __shared__ uint32_t wave_hist[SORT_WAVES][RADIX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t wbase = blockIdx.x * SORT_TILE + w * SORT_WKEYS;
    const uint32_t dv = lane & 15;
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < SORT_KPT; j++) {
        uint32_t idx = wbase + j * 64 + lane;
        bool valid = idx < n;
        uint64_t key = valid ? in[idx] : 0ull;
        uint32_t dg = (uint32_t)(key >> shift) & 15u;
        uint64_t vm = __ballot(valid);
        uint64_t b0 = __ballot(dg & 1u), b1 = __ballot(dg & 2u), b2 = __ballot(dg & 4u), b3 = __ballot(dg & 8u);
        cnt += __popcll(digit_peers(vm, b0, b1, b2, b3, dv));
    }
    if (lane < RADIX) wave_hist[w][lane] = cnt;
    __syncthreads();
    if (threadIdx.x < RADIX) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; i++) t += wave_hist[i][threadIdx.x];
        counts[threadIdx.x * nblocks + blockIdx.x] = t;
    }
}

// single block: exclusive scan in place over RADIX * nblocks counters (digit-major)
__global__ __launch_bounds__(1024) void k_scan_counts(uint32_t* __restrict__ counts, uint32_t n) {
    __shared__ uint32_t lds[1024 / 64];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n; base += 1024 * 4) {
        uint32_t idx = base + threadIdx.x * 4;
        uint32_t v[4], s = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = (idx + i < n) ? counts[idx + i] : 0; s += v[i]; }
        uint32_t inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) lds[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { uint32_t t = lds[i]; if (i < w) wbase += t; tot += t; }
        uint32_t ex = s_carry + wbase + inc - s;
#pragma unroll
        for (int i = 0; i < 4; i++) { if (idx + i < n) counts[idx + i] = ex; ex += v[i]; }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SORT_THREADS) void k_scatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                          uint32_t n, int shift, uint32_t nblocks,
                                                          const uint32_t* __restrict__ offsets) {
    __shared__ uint64_t staged[SORT_TILE];
    __shared__ uint32_t wave_hist[SORT_WAVES][RADIX];
    __shared__ uint32_t pos[SORT_WAVES][RADIX];
    __shared__ uint32_t gdelta[RADIX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t bbase = blockIdx.x * SORT_TILE;
    const uint32_t wbase = bbase + w * SORT_WKEYS;
    const uint32_t dv = lane & 15;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    uint64_t keys[SORT_KPT];
    uint32_t rnk[SORT_KPT];          // rank among same-digit keys of this wave, in stream order
    uint32_t running = 0;            // lanes 0..15: keys of digit `lane` seen so far in this wave
#pragma unroll
    for (int j = 0; j < SORT_KPT; j++) {
        uint32_t idx = wbase + j * 64 + lane;
        keys[j] = idx < n ? in[idx] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < SORT_KPT; j++) {
        uint32_t idx = wbase + j * 64 + lane;
        bool valid = idx < n;
        uint32_t dg = (uint32_t)(keys[j] >> shift) & 15u;
        uint64_t vm = __ballot(valid);
        uint64_t b0 = __ballot(dg & 1u), b1 = __ballot(dg & 2u), b2 = __ballot(dg & 4u), b3 = __ballot(dg & 8u);
        uint64_t own = digit_peers(vm, b0, b1, b2, b3, dg);
        uint32_t before = __shfl(running, (int)dg, 64);
        rnk[j] = before + __popcll(own & lt_mask);
        running += __popcll(digit_peers(vm, b0, b1, b2, b3, dv));
    }
    if (lane < RADIX) wave_hist[w][lane] = running;
    __syncthreads();
    if (threadIdx.x < RADIX) {
        const uint32_t d = threadIdx.x;
        uint32_t wh[SORT_WAVES], tot = 0;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; i++) { wh[i] = wave_hist[i][d]; tot += wh[i]; }
        uint32_t inc = tot;                        // exclusive scan over the 16 digits (lanes 0..15)
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) { uint32_t t = __shfl_up(inc, s, 16); if ((int)d >= s) inc += t; }
        uint32_t local_base = inc - tot, acc = local_base;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; i++) { pos[i][d] = acc; acc += wh[i]; }
        gdelta[d] = offsets[d * nblocks + blockIdx.x] - local_base;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_KPT; j++) {
        uint32_t idx = wbase + j * 64 + lane;
        if (idx < n) {
            uint32_t dg = (uint32_t)(keys[j] >> shift) & 15u;
            staged[pos[w][dg] + rnk[j]] = keys[j];
        }
    }
    __syncthreads();
    const uint32_t nvalid = min((uint32_t)SORT_TILE, n - bbase);
#pragma unroll
    for (int j = 0; j < SORT_KPT; j++) {
        uint32_t i = j * SORT_THREADS + threadIdx.x;
        if (i < nvalid) {
            uint64_t key = staged[i];
            uint32_t dg = (uint32_t)(key >> shift) & 15u;
            out[i + gdelta[dg]] = key;
        }
    }
}

size_t sort_counter_words(size_t n, int digit_bits) {
    (void)digit_bits;
    size_t nb = (n + SORT_TILE - 1) / SORT_TILE;
    return RADIX * nb + 16;
}

uint64_t* launch_radix_sort(hipStream_t s, const uint64_t* in, uint64_t* a, uint64_t* b, size_t n, uint64_t live_mask,
                            int lo_bit, int hi_bit, int digit_bits, uint32_t* counters, uint32_t* scan_tmp,
                            int* passes_out, hipEvent_t* pass_ev0, hipEvent_t* pass_ev1) {
    (void)digit_bits; (void)scan_tmp;
    int passes = 0;
    const uint64_t* src = in;
    uint64_t* dst = a;
    if (n > 1) {
        const uint32_t nb = (uint32_t)((n + SORT_TILE - 1) / SORT_TILE);
        for (int shift = lo_bit; shift < hi_bit; shift += RADIX_BITS) {
            int width = hi_bit - shift < RADIX_BITS ? hi_bit - shift : RADIX_BITS;
            uint64_t dmask = ((1ull << width) - 1ull) << shift;
            if ((live_mask & dmask) == 0) continue;          // single-bin histogram: pass is the identity
            if (pass_ev0) hipEventRecord(pass_ev0[passes], s);
            hipLaunchKernelGGL(k_hist, dim3(nb), dim3(SORT_THREADS), 0, s, src, (uint32_t)n, shift, nb, counters);
            hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, counters, RADIX * nb);
            hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(SORT_THREADS), 0, s, src, dst, (uint32_t)n, shift, nb,
                               (const uint32_t*)counters);
            if (pass_ev1) hipEventRecord(pass_ev1[passes], s);
            src = dst;
            dst = (dst == a) ? b : a;
            passes++;
        }
    }
    if (passes_out) *passes_out = passes;
    if (passes == 0) {
        if (n) hipMemcpyAsync(a, in, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
        return a;
    }
    return (uint64_t*)src;
}
