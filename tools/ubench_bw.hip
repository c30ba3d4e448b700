// Micro-benchmark: streaming read / write / copy rates on one MI355X for the buffer sizes of the hot path (110 MB stream).
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int KPT> __global__ void k_read(const uint4* __restrict__ a, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t base = (size_t)blockIdx.x * blockDim.x * KPT; base < n; base += (size_t)gridDim.x * blockDim.x * KPT) {
        uint4 v[KPT];
#pragma unroll
        for (int j = 0; j < KPT; j++) { size_t i = base + (size_t)j * blockDim.x + threadIdx.x; v[j] = i < n ? a[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int j = 0; j < KPT; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345u) *sink = acc;
}
template <int KPT> __global__ void k_write(uint4* __restrict__ b, size_t n) {
    for (size_t base = (size_t)blockIdx.x * blockDim.x * KPT; base < n; base += (size_t)gridDim.x * blockDim.x * KPT)
#pragma unroll
        for (int j = 0; j < KPT; j++) { size_t i = base + (size_t)j * blockDim.x + threadIdx.x; if (i < n) b[i] = make_uint4(i, j, 2, 3); }
}
template <int KPT> __global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t base = (size_t)blockIdx.x * blockDim.x * KPT; base < n; base += (size_t)gridDim.x * blockDim.x * KPT) {
        uint4 v[KPT];
#pragma unroll
        for (int j = 0; j < KPT; j++) { size_t i = base + (size_t)j * blockDim.x + threadIdx.x; v[j] = i < n ? a[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int j = 0; j < KPT; j++) { size_t i = base + (size_t)j * blockDim.x + threadIdx.x; if (i < n) b[i] = v[j]; }
    }
}
int main() {
    uint4 *a, *b; uint32_t* sink;
    const size_t cap = (size_t)1 << 30;
    hipMalloc(&a, cap); hipMalloc(&b, cap); hipMalloc(&sink, 4);
    hipMemset(a, 1, cap); hipMemset(b, 2, cap);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[3] = {(size_t)110 << 20, (size_t)256 << 20, (size_t)1 << 30};
    for (int si = 0; si < 3; si++) {
        const size_t n = sizes[si] / 16;
        for (int grid : {1024, 4096, 16384}) {
            float best[3] = {1e9f, 1e9f, 1e9f};
            for (int rep = 0; rep < 6; rep++)
                for (int k = 0; k < 3; k++) {
                    hipEventRecord(e0);
                    if (k == 0) hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, a, n, sink);
                    if (k == 1) hipLaunchKernelGGL(k_write<8>, dim3(grid), dim3(256), 0, 0, b, n);
                    if (k == 2) hipLaunchKernelGGL(k_copy<8>, dim3(grid), dim3(256), 0, 0, a, b, n);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best[k]) best[k] = ms;
                }
            printf("%5zu MB grid %5d: read %6.2f TB/s (%5.1f us)  write %6.2f TB/s (%5.1f us)  copy %6.2f TB/s r+w (%5.1f us)\n", sizes[si] >> 20, grid,
                   sizes[si] / best[0] / 1e9, best[0] * 1e3, sizes[si] / best[1] / 1e9, best[1] * 1e3, 2.0 * sizes[si] / best[2] / 1e9, best[2] * 1e3);
        }
    }
    return 0;
}
