// Development probe (run on the GPU box): sustained global -> LDS DMA rate (global_load_lds_dwordx4) per CU as a
// function of the footprint the workgroups of an XCD read — L2-resident (small) vs Infinity-Cache / HBM (large).
// hipcc --offload-arch=gfx950 -O2 -o dma_probe lds_dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const void *g, uint32_t lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_byte_addr) : "memory");
}

// every wave streams `iters` x 8 KiB; workgroup w of XCD x (w & 7) reads region x of `region_bytes`, offset by its index
__global__ __launch_bounds__(512) void k(const uint8_t *src, size_t region_bytes, uint32_t iters, int in_flight16)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const uint8_t *base = src + (size_t)xcd * region_bytes;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void *)smem);
    size_t off = ((size_t)loc * 8 + wave) * 8192 % region_bytes;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) glds16(base + off + j * 1024 + lane * 16, lds0 + (wave * 16 + (it & 1) * 8 + j) * 1024);
        off += 8 * 8192 * 32 / 32;             // next 64 KiB block of this wave's stripe
        if (off + 8192 > region_bytes) off -= region_bytes - ((region_bytes % 8192) ? 0 : 0), off %= region_bytes;
        if (in_flight16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main()
{
    const size_t max_region = 256ull << 20;
    uint8_t *d;
    hipMalloc(&d, max_region * 8);
    hipMemset(d, 1, max_region * 8);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const size_t regions[] = {1ull << 20, 2ull << 20, 3ull << 20, 8ull << 20, 24ull << 20, 256ull << 20};
    for (size_t r : regions)
        for (int fl = 0; fl < 2; fl++) {
            const uint32_t iters = 2000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 128 * 1024, 0, d, r, 200u, fl);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 128 * 1024, 0, d, r, iters, fl);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = 256.0 * 8 * iters * 8192;
            printf("region/XCD %4zu MiB, %s: %.2f ms  %.2f TB/s  = %.1f B/clk/CU at 2.4 GHz (%.1f at 1.6)\n", r >> 20,
                   fl ? "8 KiB/wave in flight across waits" : "drain each 8 KiB", ms, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9,
                   bytes / ms / 1e-3 / 256 / 1.6e9);
        }
    return 0;
}
