// Micro-benchmark: what a chained-scan look-back pays for its status-row reads on gfx950.
// Every wave repeatedly issues W loads (one 1 KB row each: 64 lanes x 4 B, or 16 lanes x 16 B of 4 rows per load) and waits.
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: dword agent-coherent (sc1)   1: dword plain   2: dwordx4 coherent (volatile)   3: dwordx4 plain
template <int MODE, int W> __global__ void k_probe(const uint32_t* buf, uint32_t rows, uint32_t window, uint64_t* out, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t acc = 0, r = (wave * 977u) % window;
    const uint64_t t0 = clock64();
    for (int it = 0; it < 32; it++) {
        if (MODE < 2) {
            uint32_t v[W];
#pragma unroll
            for (int i = 0; i < W; i++) {
                const uint32_t row = (r + i) % window;
                const uint32_t* p = buf + (size_t)row * 256 + lane;
                v[i] = MODE == 0 ? __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
#pragma unroll
            for (int i = 0; i < W; i++) acc += v[i];
        } else {
            u32x4 v[W];
#pragma unroll
            for (int i = 0; i < W; i++) {
                const uint32_t row = (r + i * 4 + (lane >> 4)) % window;
                const uint32_t* p = buf + (size_t)row * 256 + (lane & 15) * 4;
                v[i] = MODE == 2 ? *(const volatile __attribute__((address_space(1))) u32x4*)p : *(const __attribute__((address_space(1))) u32x4*)p;
            }
#pragma unroll
            for (int i = 0; i < W; i++) acc += v[i].x + v[i].y + v[i].z + v[i].w;
        }
        r = (r + 37 + (acc & 1)) % window;
    }
    const uint64_t t1 = clock64();
    if (lane == 0) out[wave] = t1 - t0;
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_stream(const uint4* a, uint4* b, size_t n, int reps) {
    for (int r = 0; r < reps; r++)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
template <int MODE, int W> static double run(hipStream_t s, int blocks, int threads, const uint32_t* buf, uint32_t rows, uint32_t window, uint64_t* out, uint32_t* sink) {
    static uint64_t h[8192];
    hipLaunchKernelGGL((k_probe<MODE, W>), dim3(blocks), dim3(threads), 0, s, buf, rows, window, out, sink);
    hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); exit(1); }
    const int nw = blocks * threads / 64;
    hipMemcpy(h, out, 8 * nw, hipMemcpyDeviceToHost);
    double a = 0; for (int i = 0; i < nw; i++) a += (double)h[i];
    return a / nw / 32;
}
int main() {
    const uint32_t rows = 4096;
    uint32_t* buf; uint64_t* out; uint32_t* sink; uint4 *sa, *sb;
    const size_t sn = (size_t)256 << 20 >> 4;
    hipMalloc(&buf, (size_t)rows * 1024); hipMemset(buf, 1, (size_t)rows * 1024); hipMalloc(&out, 8 * 8192); hipMalloc(&sink, 4);
    hipMalloc(&sa, sn * 16); hipMalloc(&sb, sn * 16); hipMemset(sa, 3, sn * 16);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const char* names[4] = {"dword coherent", "dword plain", "dwordx4 coherent", "dwordx4 plain"};
    for (int loaded = 0; loaded < 2; loaded++)
        for (int cfg = 0; cfg < 3; cfg++) {
            const int blocks = cfg == 0 ? 256 : 512, threads = cfg == 0 ? 64 : 256;       // 1 wave per CU, or 8
            for (int hot = 0; hot < 2; hot++) {
                const uint32_t window = hot ? 64 : rows;
                if (cfg == 2 && !hot) continue;
                printf("%s, %d waves per CU, %s:\n", loaded ? "beside a streaming copy" : "alone", cfg == 0 ? 1 : 8, hot ? "64-row hot window" : "rows spread over 4 MB");
                for (int mode = 0; mode < 4; mode++) {
                    if (loaded) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s2, sa, sb, sn, 6);
                    double c8 = mode == 0 ? run<0, 8>(s1, blocks, threads, buf, rows, window, out, sink) : mode == 1 ? run<1, 8>(s1, blocks, threads, buf, rows, window, out, sink)
                              : mode == 2 ? run<2, 8>(s1, blocks, threads, buf, rows, window, out, sink) : run<3, 8>(s1, blocks, threads, buf, rows, window, out, sink);
                    if (loaded) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s2, sa, sb, sn, 6);
                    double c1 = mode == 0 ? run<0, 1>(s1, blocks, threads, buf, rows, window, out, sink) : mode == 1 ? run<1, 1>(s1, blocks, threads, buf, rows, window, out, sink)
                              : mode == 2 ? run<2, 1>(s1, blocks, threads, buf, rows, window, out, sink) : run<3, 1>(s1, blocks, threads, buf, rows, window, out, sink);
                    printf("    %-18s 1 load in flight: %6.0f cycles   8 in flight: %6.0f cycles (%s rows)\n", names[mode], c1, c8, mode < 2 ? "8" : "32");
                }
            }
            if (cfg == 1) cfg = 2;
        }
    return 0;
}
