// radix_rank.h — wavefront ranking primitives shared by the radix sort (sort.hip) and the per-row run sort of the carry
// pre-pass (paint.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// lanes of this wave whose BITS-bit digit equals mine, as two 32-bit halves.  Per digit bit: one sign-extending
// bit-field extract, one compare (the ballot) and one 3-input boolean op per half — `m & ~(ballot ^ -bit)`.
template <int BITS>
__device__ __forceinline__ void match_any(uint32_t dg, uint32_t& mlo, uint32_t& mhi) {
    mlo = ~0u; mhi = ~0u;
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const uint32_t nb = (uint32_t)(-(int32_t)((dg >> b) & 1u));       // 0 or 0xFFFFFFFF
        const uint64_t bal = __ballot(nb != 0u);
        mlo &= ~((uint32_t)bal ^ nb);
        mhi &= ~((uint32_t)(bal >> 32) ^ nb);
    }
}
// The 8-bit match-any, hand-scheduled: per digit bit one v_bfe_i32 (0 / -1), one v_cmp (the ballot, into an SGPR pair)
// and one v_bitop3 per half (m & ~(ballot ^ -bit), truth table 0x90) = 32 VALU instructions, against ~100 from the
// compiler for the loop above.  Four SGPR pairs rotate so that every ballot is >= 3 instructions old when it is read
// (gfx950 needs 2 wait states between a VALU SGPR write and a VALU read of it).
template <>
__device__ __forceinline__ void match_any<8>(uint32_t dg, uint32_t& mlo, uint32_t& mhi) {
    uint32_t t0, t1, t2, t3;
    asm volatile(
        "v_bfe_i32 %2, %6, 0, 1\n\t"
        "v_bfe_i32 %3, %6, 1, 1\n\t"
        "v_bfe_i32 %4, %6, 2, 1\n\t"
        "v_bfe_i32 %5, %6, 3, 1\n\t"
        "v_cmp_ne_u32_e64 s[92:93], 0, %2\n\t"
        "v_cmp_ne_u32_e64 s[94:95], 0, %3\n\t"
        "v_cmp_ne_u32_e64 s[96:97], 0, %4\n\t"
        "v_cmp_ne_u32_e64 s[98:99], 0, %5\n\t"
        "v_xnor_b32 %0, s92, %2\n\t"
        "v_xnor_b32 %1, s93, %2\n\t"
        "v_bitop3_b32 %0, %0, s94, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s95, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, s96, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s97, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, s98, %5 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s99, %5 bitop3:0x90\n\t"
        "v_bfe_i32 %2, %6, 4, 1\n\t"
        "v_bfe_i32 %3, %6, 5, 1\n\t"
        "v_bfe_i32 %4, %6, 6, 1\n\t"
        "v_bfe_i32 %5, %6, 7, 1\n\t"
        "v_cmp_ne_u32_e64 s[92:93], 0, %2\n\t"
        "v_cmp_ne_u32_e64 s[94:95], 0, %3\n\t"
        "v_cmp_ne_u32_e64 s[96:97], 0, %4\n\t"
        "v_cmp_ne_u32_e64 s[98:99], 0, %5\n\t"
        "v_bitop3_b32 %0, %0, s92, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s93, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, s94, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s95, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, s96, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s97, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, s98, %5 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s99, %5 bitop3:0x90\n\t"
        : "=&v"(mlo), "=&v"(mhi), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(dg)
        : "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
}
// nine bits: the hand-scheduled eight, then the ninth like the generic loop
template <>
__device__ __forceinline__ void match_any<9>(uint32_t dg, uint32_t& mlo, uint32_t& mhi) {
    match_any<8>(dg, mlo, mhi);
    const uint32_t nb = (uint32_t)(-(int32_t)((dg >> 8) & 1u));
    const uint64_t bal = __ballot(nb != 0u);
    mlo &= ~((uint32_t)bal ^ nb);
    mhi &= ~((uint32_t)(bal >> 32) ^ nb);
}
__device__ __forceinline__ uint32_t lanes_below(uint32_t mlo, uint32_t mhi) {          // popcount(m & lanemask_lt)
    return __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
}

