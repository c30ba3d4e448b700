// ubench_valu.hip — issue cost of the VALU instructions the rasterizer's `find` is made of (f64 fma / ceil / conversions against
// f32 fma): one wavefront per SIMD, 16 independent chains, shader clocks per instruction.
//     hipcc -O2 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu && tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define OP_LOOP(NAME, DECL, BODY, SINK)                                                             \
    __global__ void NAME(uint64_t* out, double seed) {                                              \
        DECL;                                                                                       \
        uint64_t t0 = __builtin_readcyclecounter();                                                 \
        for (int it = 0; it < ITERS; it++) { BODY; }                                                \
        uint64_t t1 = __builtin_readcyclecounter();                                                 \
        SINK;                                                                                       \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                            \
    }

OP_LOOP(k_fma_f32, float a[16]; for (int i = 0; i < 16; i++) a[i] = (float)seed + i,
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i])),
        float s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345.f) out[1] = 1)
OP_LOOP(k_fma_f64, double a[16]; for (int i = 0; i < 16; i++) a[i] = seed + i,
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i])),
        double s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345.) out[1] = 1)
OP_LOOP(k_ceil_f64, double a[16]; for (int i = 0; i < 16; i++) a[i] = seed + i,
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_ceil_f64 %0, %0" : "+v"(a[i])),
        double s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345.) out[1] = 1)
OP_LOOP(k_cvt_f32_f64, double a[16]; float b[16]; for (int i = 0; i < 16; i++) { a[i] = seed + i; b[i] = 0; },
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(b[i]) : "v"(a[i])),
        float s = 0; for (int i = 0; i < 16; i++) s += b[i]; if (s == 12345.f) out[1] = 1)
OP_LOOP(k_cvt_f64_f32, double a[16]; float b[16]; for (int i = 0; i < 16; i++) { b[i] = (float)seed + i; a[i] = 0; },
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(b[i])),
        double s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345.) out[1] = 1)
OP_LOOP(k_floor_f32, float a[16]; for (int i = 0; i < 16; i++) a[i] = (float)seed + i,
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i])),
        float s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345.f) out[1] = 1)
OP_LOOP(k_cvt_i32_f32, float a[16]; int b[16]; for (int i = 0; i < 16; i++) { a[i] = (float)seed + i; b[i] = 0; },
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(b[i]) : "v"(a[i])),
        int s = 0; for (int i = 0; i < 16; i++) s += b[i]; if (s == 12345) out[1] = 1)
OP_LOOP(k_lshl_b64, uint64_t a[16]; for (int i = 0; i < 16; i++) a[i] = (uint64_t)seed + i,
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i])),
        uint64_t s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345) out[1] = 1)
OP_LOOP(k_and_b32, uint32_t a[16]; for (int i = 0; i < 16; i++) a[i] = (uint32_t)seed + i,
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_and_b32 %0, 0x7ff000, %0" : "+v"(a[i])),
        uint32_t s = 0; for (int i = 0; i < 16; i++) s += a[i]; if (s == 12345) out[1] = 1)

int main() {
    uint64_t* d; hipMalloc(&d, 4096 * 8);
    uint64_t h[8];
#define RUN(K, WAVES)                                                                              \
    do {                                                                                           \
        K<<<1024, 64 * WAVES>>>(d, 1.5); K<<<1024, 64 * WAVES>>>(d, 1.5); hipDeviceSynchronize(); \
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);                                                \
        printf("%-16s %d wave(s)/SIMD-ish: %.2f clocks per instruction per wave\n", #K, WAVES, (double)h[0] / (ITERS * 16.0)); \
    } while (0)
    RUN(k_fma_f32, 1); RUN(k_fma_f64, 1); RUN(k_ceil_f64, 1); RUN(k_cvt_f32_f64, 1); RUN(k_cvt_f64_f32, 1); RUN(k_floor_f32, 1);
    RUN(k_cvt_i32_f32, 1); RUN(k_lshl_b64, 1); RUN(k_and_b32, 1);
    RUN(k_fma_f32, 4); RUN(k_fma_f64, 4); RUN(k_ceil_f64, 4); RUN(k_cvt_f32_f64, 4); RUN(k_cvt_f64_f32, 4); RUN(k_lshl_b64, 4);
    RUN(k_fma_f32, 8); RUN(k_fma_f64, 8); RUN(k_ceil_f64, 8); RUN(k_cvt_f32_f64, 8);
    return 0;
}
