// lookback.h — chained-scan ("decoupled look-back") helpers shared by the HIP translation units.
//
// Tile status words carry their own flag, so the data IS the flag (MI355X guide, Guideline 16 "R2 granule"):
// one naturally aligned relaxed agent-scope store publishes, relaxed agent-scope loads poll, no fences.
//   u32 status = [flag 2 | value 30]            u64 status = [flag 2 | count 30 | sum 32]
//   flag 0 = not published, 1 = this tile's aggregate, 2 = inclusive prefix of all tiles up to this one
// Tiles are handed out by an atomic ticket, so a tile only ever waits for tiles that are already running.
//
// The look-back itself is WAVE-PARALLEL: 64 predecessors per probe.  A serial walk (one predecessor per
// L2 round trip) cannot keep up once tiles retire faster than one per round trip — the window of
// not-yet-prefixed predecessors then grows until every tile walks the whole in-flight set.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LB_AGG    1u
#define LB_PREFIX 2u
#define LB_SPIN_LIMIT (1u << 22)

typedef __attribute__((address_space(1))) unsigned int       lb_gu32;
typedef __attribute__((address_space(1))) unsigned long long lb_gu64;

__device__ __forceinline__ uint32_t lb_ld32(const uint32_t* p) {
    return __hip_atomic_load((lb_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Same word through the caches (no coherence bits): may be STALE.  Only for hints whose staleness costs a redundant
// operation, never correctness.
__device__ __forceinline__ uint32_t lb_ld32_cached(const uint32_t* p) {
    return __hip_atomic_load((lb_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void lb_st32(uint32_t* p, uint32_t v) {
    __hip_atomic_store((lb_gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t lb_ld64(const uint64_t* p) {
    return __hip_atomic_load((lb_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_st64(uint64_t* p, uint64_t v) {
    __hip_atomic_store((lb_gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t lb_wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Called by ALL 64 lanes of ONE wave.  Returns the exclusive prefix of tile `tile` (sum of the values of tiles
// 0..tile-1).  Sets bit 2 of *err and returns garbage if a predecessor never shows up (bounded spin).
__device__ __forceinline__ uint32_t lb_lookback_u32(const uint32_t* status, uint32_t tile, uint32_t* err) {
    const int lane = threadIdx.x & 63;
    uint32_t excl = 0, spins = 0;
    int p0 = (int)tile - 1;
    while (p0 >= 0) {
        const int p = p0 - lane;
        const uint32_t v = p >= 0 ? lb_ld32(&status[p]) : (LB_PREFIX << 30);      // virtual prefix 0 before tile 0
        const uint32_t f = v >> 30;
        const uint64_t ready = __ballot(f != 0), pref = __ballot(f == LB_PREFIX);
        const int fp = pref ? __builtin_ctzll(pref) : 64;                          // nearest inclusive prefix
        const uint64_t need = fp >= 63 ? ~0ull : ((2ull << fp) - 1ull);            // lanes 0..fp
        if ((ready & need) != need) {
            if (++spins > LB_SPIN_LIMIT) { if (lane == 0) atomicOr(err, 4u); break; }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        excl += lb_wave_sum(lane <= fp ? (v & 0x3FFFFFFFu) : 0u);
        if (fp < 64) break;
        p0 -= 64;
    }
    return excl;
}

// u64 variant: two sums at once.  count (30 bits) and sum (32 bits, wrapping like the reference's u32 prefix sum).
__device__ __forceinline__ void lb_lookback_u64(const uint64_t* status, uint32_t tile, uint32_t* err, uint32_t* out_cnt,
                                                uint32_t* out_sum) {
    const int lane = threadIdx.x & 63;
    uint32_t cnt = 0, sum = 0, spins = 0;
    int p0 = (int)tile - 1;
    while (p0 >= 0) {
        const int p = p0 - lane;
        const uint64_t v = p >= 0 ? lb_ld64(&status[p]) : ((uint64_t)LB_PREFIX << 62);
        const uint32_t f = (uint32_t)(v >> 62);
        const uint64_t ready = __ballot(f != 0), pref = __ballot(f == LB_PREFIX);
        const int fp = pref ? __builtin_ctzll(pref) : 64;
        const uint64_t need = fp >= 63 ? ~0ull : ((2ull << fp) - 1ull);
        if ((ready & need) != need) {
            if (++spins > LB_SPIN_LIMIT) { if (lane == 0) atomicOr(err, 4u); break; }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        const bool take = lane <= fp;
        cnt += lb_wave_sum(take ? ((uint32_t)(v >> 32) & 0x3FFFFFFFu) : 0u);
        sum += lb_wave_sum(take ? (uint32_t)v : 0u);
        if (fp < 64) break;
        p0 -= 64;
    }
    *out_cnt = cnt; *out_sum = sum;
}
