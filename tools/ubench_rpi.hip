// Is gfx950's v_cvt_rpi_i32_f32 ("round to plus infinity") the reference's `(x + 0.5).floor() as i32` (rasterizer.rs:78-80)?  All 2^32 inputs.
// Answer (MI355X): NO — 25 165 823 mismatches: 0.49999997 (the f32 sum rounds up to 1.0, the instruction rounds the exact sum), the odd
// integers of magnitude 2^23..2^24 (ties to even in the f32 sum) and every NaN.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_rpi.hip -o tools/ubench_rpi
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ int rpi(float x) { int r; asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__global__ void k(unsigned long long* bad, uint32_t* first) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long nb = 0;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
        const float x = __uint_as_float((uint32_t)u);
        const int a = (int)floorf(x + 0.5f);
        const int b = rpi(x);
        if (a != b) { nb++; atomicMin(first, (uint32_t)u); }
    }
    if (nb) atomicAdd(bad, nb);
}
int main() {
    unsigned long long* bad; uint32_t* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0xFF, 4);
    k<<<4096, 256>>>(bad, first);
    unsigned long long hb; uint32_t hf;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("mismatches %llu first 0x%08x\n", hb, hf);
    return 0;
}
