// tools/ubench_atomic.hip — what do global (device-scope, non-returning) atomicAdds cost on MI355X?
// N atomics from a grid of 256-thread workgroups into a table of M words, addresses pseudo-random or run-coherent.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_atomic tools/ubench_atomic.hip && tools/ubench_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_atomic(uint32_t* tab, uint32_t mask, uint32_t per_thread, uint32_t run) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = 0; i < per_thread; i++) {
        uint32_t k = (g * per_thread + i) / run;                 // `run` consecutive atomics share an address
        uint32_t a = (k * 2654435761u) >> 7;
        atomicAdd(&tab[a & mask], 1u);
    }
}
__global__ void k_stream(const uint4* in, uint4* out, size_t n) {   // background copy
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
    uint32_t* tab; hipMalloc(&tab, 64u << 20); hipMemset(tab, 0, 64u << 20);
    uint4 *a, *b; size_t nb = 110u << 20; hipMalloc(&a, nb); hipMalloc(&b, nb); hipMemset(a, 1, nb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t s2; hipStreamCreate(&s2);
    const uint32_t total = 1u << 20;
    for (int streaming = 0; streaming < 2; streaming++)
    for (uint32_t run : {1u, 8u, 64u})
    for (uint32_t words : {128u, 4096u, 32768u, 1u << 20}) {
        const uint32_t per_thread = 4, threads = total / per_thread;
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            if (streaming) for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s2, a, b, nb / 16);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_atomic, dim3(threads / 256), dim3(256), 0, 0, tab, words - 1, per_thread, run);
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s run %2u table %8u words: %8.1f us for %u atomics = %6.2f ns each\n", streaming ? "beside a copy" : "idle         ",
               run, words, best * 1e3f, total, best * 1e6f / total);
    }
    return 0;
}
