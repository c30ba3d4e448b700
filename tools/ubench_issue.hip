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
    for (int w = 1; w <= 4; w *= 2) { RUN(k_fma_f32, w); RUN(k_fma_f64, w); RUN(k_and_b32, w); RUN(k_pk_fma_f32, w); RUN(k_ceil_f64, w); }
    return 0;
}
