// ubench_hostwrite.hip — can the painter write its pixels STRAIGHT into (registered, mapped) caller memory at the link's rate?  A 4K RGBA8 image
// written by one wavefront per 16 x 16 tile exactly as k_paint_wave stores it (lane = x + 16 * row group, four rows per lane: a store instruction is four
// 64-byte row pieces), into device memory, into hipHostMalloc'd memory and into malloc'd + hipHostRegister'ed memory; against hipMemcpy2DAsync of the image.
//     hipcc -O2 --offload-arch=gfx950 tools/ubench_hostwrite.hip -o tools/ubench_hostwrite && tools/ubench_hostwrite
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define W 3840
#define H 2160
__global__ __launch_bounds__(64) void k_tiles(uint32_t* __restrict__ img, uint32_t seed) {
    const uint32_t tw = W / 16, tile = blockIdx.x, tx = tile % tw, ty = tile / tw;
    const uint32_t lx = threadIdx.x & 15u, rg = threadIdx.x >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t py = ty * 16u + rg * 4u + r, px = tx * 16u + lx;
        img[(size_t)py * W + px] = seed + py * 7u + px;
    }
}
// the same pixels, a lane = four pixels of a row (16-byte stores: a store instruction is sixteen 64-byte row pieces... of ONE tile: all 16 rows)
__global__ __launch_bounds__(64) void k_tiles16(uint32_t* __restrict__ img, uint32_t seed) {
    const uint32_t tw = W / 16, tile = blockIdx.x, tx = tile % tw, ty = tile / tw;
    const uint32_t q = threadIdx.x & 3u, row = threadIdx.x >> 2;
    const uint32_t py = ty * 16u + row, px = tx * 16u + q * 4u;
    uint4 v = make_uint4(seed + py * 7u + px, seed + py * 7u + px + 1, seed + py * 7u + px + 2, seed + py * 7u + px + 3);
    *reinterpret_cast<uint4*>(img + (size_t)py * W + px) = v;
}
static float run(void (*k)(uint32_t*, uint32_t), uint32_t* p, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<(W / 16) * (H / 16), 64>>>(p, 1u); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++) k<<<(W / 16) * (H / 16), 64>>>(p, (uint32_t)i);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}
int main() {
    const size_t bytes = (size_t)W * H * 4;
    uint32_t *dev, *hm, *reg, *reg_d;
    (void)hipMalloc(&dev, bytes);
    (void)hipHostMalloc(&hm, bytes, hipHostMallocMapped);
    reg = (uint32_t*)aligned_alloc(4096, bytes); for (size_t i = 0; i < bytes / 4; i += 1024) reg[i] = 0;
    if (hipHostRegister(reg, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { printf("register failed\n"); return 1; }
    (void)hipHostGetDevicePointer((void**)&reg_d, reg, 0);
    struct { const char* name; uint32_t* p; } T[] = {{"device memory", dev}, {"hipHostMalloc", hm}, {"hipHostRegister", reg_d}};
    for (auto& t : T) {
        const float a = run(k_tiles, t.p, 10), b = run(k_tiles16, t.p, 10);
        printf("%-16s 4-byte stores %8.1f us = %6.1f GB/s    16-byte stores %8.1f us = %6.1f GB/s\n", t.name, a, bytes / a * 1e-3, b, bytes / b * 1e-3);
    }
    // check what landed
    (void)hipDeviceSynchronize();
    size_t bad = 0; for (uint32_t py = 0; py < H; py += 37) for (uint32_t px = 0; px < W; px += 11) bad += reg[(size_t)py * W + px] != 9u + py * 7u + px;
    printf("registered memory holds the last launch's pixels: %s\n", bad ? "NO" : "yes");
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemcpy2DAsync(reg, W * 4, dev, W * 4, W * 4, H, hipMemcpyDeviceToHost, 0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; i++) (void)hipMemcpy2DAsync(reg, W * 4, dev, W * 4, W * 4, H, hipMemcpyDeviceToHost, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpy2DAsync D2H into the registered image %8.1f us = %6.1f GB/s\n", ms * 100.f, bytes / (ms * 100.f) * 1e-3);
    return 0;
}
