// ubench_issue.hip — the chip's VALU issue peak in wave64 instructions per second: what `roofline_painter.peak` (bench.py) is priced against.
// One workgroup per CU (100 KB of LDS keeps a second one off it) of W waves per SIMD, every wave 16 independent chains of one
// instruction; wall time by HIP events -> G wave-instructions/s chip-wide and clocks per instruction per SIMD (at the clock the run reports).
//     hipcc -O2 --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue && tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
#define CHAINS 16
#define KERNEL(NAME, TYPE, INIT, ASM)                                                               \
    __global__ void NAME(uint64_t* out, float seed) {                                               \
        __shared__ uint32_t pad[25000];                                                             \
        TYPE a[CHAINS];                                                                             \
        for (int i = 0; i < CHAINS; i++) a[i] = INIT;                                               \
        const uint64_t t0 = __builtin_readcyclecounter();                                           \
        for (int it = 0; it < ITERS; it++) {                                                        \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) asm volatile(ASM : "+v"(a[i]));      \
        }                                                                                           \
        const uint64_t t1 = __builtin_readcyclecounter();                                           \
        TYPE s = 0; for (int i = 0; i < CHAINS; i++) s += a[i];                                     \
        if (s == (TYPE)12345) pad[threadIdx.x] = 1;                                                 \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                            \
        if (s == (TYPE)54321) out[1000] = pad[(threadIdx.x * 7) % 25000];                           \
    }
KERNEL(k_fma_f32, float, seed + i, "v_fma_f32 %0, %0, %0, %0")
KERNEL(k_fma_f64, double, (double)seed + i, "v_fma_f64 %0, %0, %0, %0")
KERNEL(k_and_b32, uint32_t, (uint32_t)seed + i, "v_and_b32 %0, 0x7ff000, %0")
KERNEL(k_pk_fma_f32, double, (double)seed + i, "v_pk_fma_f32 %0, %0, %0, %0")
KERNEL(k_ceil_f64, double, (double)seed + i, "v_ceil_f64 %0, %0")
KERNEL(k_lshl_b64, uint64_t, (uint64_t)seed + i, "v_lshlrev_b64 %0, 3, %0")
KERNEL(k_mul_lo_u32, uint32_t, (uint32_t)seed + i, "v_mul_lo_u32 %0, %0, %0")
KERNEL(k_cvt_i32_f32, float, seed + i, "v_cvt_i32_f32 %0, %0")
KERNEL(k_bfe_u32, uint32_t, (uint32_t)seed + i, "v_bfe_u32 %0, %0, 3, 9")
KERNEL(k_cndmask, uint32_t, (uint32_t)seed + i, "v_cndmask_b32 %0, %0, %0, vcc")
KERNEL(k_cndmask_e64, uint32_t, (uint32_t)seed + i, "v_cndmask_b32_e64 %0, %0, %0, s[20:21]")
KERNEL(k_cmp_cnd, uint32_t, (uint32_t)seed + i, "v_cmp_lt_u32 vcc, 5, %0\n\tv_cndmask_b32 %0, %0, %0, vcc")
KERNEL(k_cmp_cnd_e64, uint32_t, (uint32_t)seed + i, "v_cmp_lt_u32_e64 s[20:21], 5, %0\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %0, s[20:21]")
KERNEL(k_cmp_4_cnd, uint32_t, (uint32_t)seed + i, "v_cmp_lt_u32 vcc, 5, %0\n\tv_add_u32 %0, 3, %0\n\tv_add_u32 %0, 3, %0\n\tv_add_u32 %0, 3, %0\n\tv_add_u32 %0, 3, %0\n\tv_cndmask_b32 %0, %0, %0, vcc")
KERNEL(k_cmp_4_cnd_e64, uint32_t, (uint32_t)seed + i, "v_cmp_lt_u32_e64 s[20:21], 5, %0\n\tv_add_u32 %0, 3, %0\n\tv_add_u32 %0, 3, %0\n\tv_add_u32 %0, 3, %0\n\tv_add_u32 %0, 3, %0\n\tv_cndmask_b32_e64 %0, %0, %0, s[20:21]")
KERNEL(k_addc, uint32_t, (uint32_t)seed + i, "v_add_co_u32 %0, vcc, 3, %0")
KERNEL(k_mov_b32, uint32_t, (uint32_t)seed + i, "v_mov_b32 %0, %0")
KERNEL(k_lshl_b32, uint32_t, (uint32_t)seed + i, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_lshr_b32, uint32_t, (uint32_t)seed + i, "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_or_b32, uint32_t, (uint32_t)seed + i, "v_or_b32 %0, 5, %0")
KERNEL(k_sub_u32, uint32_t, (uint32_t)seed + i, "v_sub_u32 %0, %0, %0")
KERNEL(k_max_i32, uint32_t, (uint32_t)seed + i, "v_max_i32 %0, 5, %0")
KERNEL(k_max_f32, float, seed + i, "v_max_f32 %0, 0, %0")
KERNEL(k_mul_f32, float, seed + i, "v_mul_f32 %0, %0, %0")
KERNEL(k_add_f32, float, seed + i, "v_add_f32 %0, 0.5, %0")
KERNEL(k_and_or, uint32_t, (uint32_t)seed + i, "v_and_or_b32 %0, %0, 15, %0")
KERNEL(k_or3, uint32_t, (uint32_t)seed + i, "v_or3_b32 %0, %0, %0, %0")
KERNEL(k_bitop3, uint32_t, (uint32_t)seed + i, "v_bitop3_b32 %0, %0, %0, %0 bitop3:0x90")
KERNEL(k_mbcnt, uint32_t, (uint32_t)seed + i, "v_mbcnt_lo_u32_b32 %0, %0, %0")
KERNEL(k_mad_u32_u24, uint32_t, (uint32_t)seed + i, "v_mad_u32_u24 %0, %0, %0, %0")
KERNEL(k_alignbit, uint32_t, (uint32_t)seed + i, "v_alignbit_b32 %0, %0, %0, 20")
KERNEL(k_perm, uint32_t, (uint32_t)seed + i, "v_perm_b32 %0, %0, %0, %0")
KERNEL(k_cvt_f32_i32, float, seed + i, "v_cvt_f32_i32 %0, %0")
KERNEL(k_rcp_f32, float, seed + i, "v_rcp_f32 %0, %0")
KERNEL(k_add_f64, double, (double)seed + i, "v_add_f64 %0, %0, %0")
KERNEL(k_mov_dpp, uint32_t, (uint32_t)seed + i, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_cmp_e64, uint32_t, (uint32_t)seed + i, "v_cmp_lt_u32_e64 s[20:21], 5, %0")
KERNEL(k_min_u32, uint32_t, (uint32_t)seed + i, "v_min_u32 %0, 77, %0")
KERNEL(k_bfi, uint32_t, (uint32_t)seed + i, "v_bfi_b32 %0, %0, %0, %0")
KERNEL(k_lshl_add, uint32_t, (uint32_t)seed + i, "v_lshl_add_u32 %0, %0, 2, %0")
KERNEL(k_add_u32, uint32_t, (uint32_t)seed + i, "v_add_u32 %0, 3, %0")
KERNEL(k_floor_f32, float, seed + i, "v_floor_f32 %0, %0")

int main() {
    uint64_t* d; (void)hipMalloc(&d, 2048 * 8);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clockRate %d kHz\n", p.gcnArchName, cus, p.clockRate);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#define RUN(K, W)                                                                                   \
    do {                                                                                            \
        K<<<cus, 256 * W>>>(d, 1.5f); (void)hipDeviceSynchronize();                                 \
        (void)hipEventRecord(e0); K<<<cus, 256 * W>>>(d, 1.5f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                           \
        uint64_t h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);                               \
        const double insts = (double)cus * 4 * W * ITERS * CHAINS;                                  \
        printf("%-14s %d waves/SIMD: %8.1f G wave-instr/s chip-wide (%.1f us), %.2f shader clocks per instruction per SIMD\n", \
               #K, W, insts / (ms * 1e-3) * 1e-9, ms * 1e3, (double)h / ((double)W * ITERS * CHAINS));               \
    } while (0)
    for (int w = 4; w <= 4; w *= 4) { RUN(k_fma_f32, w); RUN(k_fma_f64, w); RUN(k_and_b32, w); RUN(k_pk_fma_f32, w); RUN(k_ceil_f64, w); RUN(k_lshl_b64, w); RUN(k_mul_lo_u32, w); RUN(k_cvt_i32_f32, w); RUN(k_bfe_u32, w); RUN(k_cndmask, w); RUN(k_cndmask_e64, w); RUN(k_cmp_cnd, w); RUN(k_cmp_cnd_e64, w); RUN(k_cmp_4_cnd, w); RUN(k_cmp_4_cnd_e64, w); RUN(k_addc, w); RUN(k_mov_b32, w); RUN(k_lshl_b32, w); RUN(k_lshr_b32, w); RUN(k_or_b32, w); RUN(k_sub_u32, w); RUN(k_max_i32, w); RUN(k_max_f32, w); RUN(k_mul_f32, w); RUN(k_add_f32, w); RUN(k_and_or, w); RUN(k_or3, w); RUN(k_bitop3, w); RUN(k_mbcnt, w); RUN(k_mad_u32_u24, w); RUN(k_alignbit, w); RUN(k_perm, w); RUN(k_cvt_f32_i32, w); RUN(k_rcp_f32, w); RUN(k_add_f64, w); RUN(k_mov_dpp, w); RUN(k_cmp_e64, w); RUN(k_min_u32, w); RUN(k_bfi, w); RUN(k_lshl_add, w); RUN(k_add_u32, w); RUN(k_floor_f32, w); }
    return 0;
}
