// Where do the waves of an eight-wave workgroup land?  (round 6: k_attention_s<8> runs 7 query tiles on 8 waves at 197 tokens; if wave i of
// every workgroup goes to SIMD i mod 4, the idle wave and the 5-query tail wave of the co-resident workgroups (two per CU for the real kernel's 120 VGPRs) share two SIMDs and the
// other two carry a third more work.)  Launches k_attention_s's geometry (grid (12, n, 1), 512 threads, 40 KiB dynamic LDS; this probe has few registers, so four workgroups
// fit a CU) with a kernel that spins for a while and records HW_ID / XCC_ID of every wave and its start time.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/wave_placement_probe.hip -o /tmp/wpp && /tmp/wpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>

__global__ __launch_bounds__(512, 4) void k_probe(uint32_t *__restrict__ out, uint32_t spin)
{
    extern __shared__ uint8_t smem[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_readcyclecounter();
    float acc = (float)lane;
    for (uint32_t i = 0; i < spin; i++) acc = fmaf(acc, 1.0001f, 0.5f);
    smem[threadIdx.x] = (uint8_t)acc;
    __syncthreads();
    if (lane == 0) {
        const uint32_t L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        uint32_t *o = out + ((size_t)L * 8 + wave) * 4;
        o[0] = hw;
        o[1] = xcc;
        o[2] = (uint32_t)(t0 >> 8);
        o[3] = smem[0];
    }
}

int main()
{
    const uint32_t heads = 12, n = 256;
    const size_t n_wg = (size_t)heads * n;
    uint32_t *d;
    hipMalloc(&d, n_wg * 8 * 4 * 4);
    hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
    hipLaunchKernelGGL(k_probe, dim3(heads, n, 1), dim3(512), 40 * 1024, 0, d, 20000u);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(n_wg * 8 * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    size_t hist[8][4] = {};
    std::map<uint32_t, std::vector<std::pair<uint32_t, uint32_t>>> per_cu;      // (xcc, se, sh, cu) -> (start time, L)
    for (size_t L = 0; L < n_wg; L++)
        for (int w = 0; w < 8; w++) {
            const uint32_t hw = h[(L * 8 + w) * 4], xcc = h[(L * 8 + w) * 4 + 1] & 15u;
            hist[w][(hw >> 4) & 3]++;
            if (w == 0) per_cu[(xcc << 16) | (hw & 0xff00u)].push_back({h[(L * 8 + w) * 4 + 2], (uint32_t)L});
        }
    printf("wave index -> SIMD histogram over %zu workgroups\n", n_wg);
    for (int w = 0; w < 8; w++) printf("  wave %d: simd0 %zu simd1 %zu simd2 %zu simd3 %zu\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("CUs seen: %zu\n", per_cu.size());
    int shown = 0;
    for (auto &kv : per_cu) {
        auto v = kv.second;
        std::sort(v.begin(), v.end());
        if (shown++ < 6) {
            printf("  cu key %06x: %zu workgroups; first (time>>8, linear id):", kv.first, v.size());
            for (size_t i = 0; i < v.size() && i < 8; i++) printf(" (%u, %u)", v[i].first - v[0].first, v[i].second);
            printf("\n");
        }
    }
    // the first four workgroups of a CU (the first generation): how do their linear ids relate?
    std::map<uint32_t, size_t> diffs;
    for (auto &kv : per_cu) {
        auto v = kv.second;
        std::sort(v.begin(), v.end());
        for (size_t i = 1; i < v.size() && i < 4; i++) diffs[v[i].second - v[0].second]++;
    }
    printf("linear-id distance of a CU's first-generation workgroups from its first one (distance: count), top entries:\n");
    std::vector<std::pair<size_t, uint32_t>> dv;
    for (auto &kv : diffs) dv.push_back({kv.second, kv.first});
    std::sort(dv.rbegin(), dv.rend());
    for (size_t i = 0; i < dv.size() && i < 10; i++) printf("  %u: %zu\n", dv[i].second, dv[i].first);
    return 0;
}
