// Where the waves of a 5-wave workgroup land: HW_ID (SIMD, CU, SE) per (workgroup, wave), for a grid of 512 / 768 / 1024 workgroups
// with 32 KB of LDS each (k_traj_pipe's shape at B = 4096 / 6144 / 8192).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/wave_placement_probe.hip -o tools/probes/wave_placement_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(320) k(unsigned* out, int spin) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    lds[threadIdx.x] = (float)hw;
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);   // keep the workgroups co-resident
    if (lane == 0) { out[(blockIdx.x * 5 + wave) * 2] = hw; out[(blockIdx.x * 5 + wave) * 2 + 1] = xcc; }
    if (lds[0] == -1.f) out[0] = 0;
}

int main() {
    unsigned* d;
    CK(hipMalloc(&d, 1024 * 5 * 2 * 4));
    for (int nwg : {256, 512, 768, 1024}) {
        CK(hipMemset(d, 0, 1024 * 5 * 2 * 4));
        hipLaunchKernelGGL(k, dim3(nwg), dim3(320), 32 * 1024, 0, d, 400000);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(nwg * 10);
        CK(hipMemcpy(h.data(), d, nwg * 10 * 4, hipMemcpyDeviceToHost));
        // per (xcc, se, cu): which SIMD each workgroup's wave 0 got, and the SIMDs of waves 0..4 of the first workgroups
        std::map<unsigned, std::vector<int>> w0;          // CU key -> SIMD ids of the wave-0s it hosts
        int hist[4] = {0, 0, 0, 0};
        for (int b = 0; b < nwg; ++b) {
            const unsigned hw = h[(b * 5) * 2], xcc = h[(b * 5) * 2 + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, se = (hw >> 13) & 7;
            w0[(xcc << 8) | (se << 4) | cu].push_back((int)simd);
            hist[simd]++;
        }
        printf("\n%d workgroups of 5 waves: wave 0 on SIMD 0/1/2/3: %d %d %d %d; CUs used %zu\n", nwg, hist[0], hist[1], hist[2], hist[3], w0.size());
        int shown = 0, clash = 0, multi = 0;
        for (auto& kv : w0) {
            if (kv.second.size() > 1) {
                ++multi;
                bool same = false;
                for (size_t i = 0; i < kv.second.size(); ++i) for (size_t j = i + 1; j < kv.second.size(); ++j) same |= kv.second[i] == kv.second[j];
                clash += same;
            }
            if (shown < 6) { printf("  CU %03x: wave-0 SIMDs:", kv.first); for (int s : kv.second) printf(" %d", s); printf("\n"); ++shown; }
        }
        printf("  CUs hosting several workgroups: %d, of which two wave-0s share a SIMD: %d\n", multi, clash);
        printf("  first workgroups, SIMD of waves 0..4:");
        for (int b = 0; b < 6; ++b) { printf("  [wg %d:", b); for (int w = 0; w < 5; ++w) printf(" %u", (h[(b * 5 + w) * 2] >> 4) & 3); printf(" cu %u]", (h[b * 10] >> 8) & 15); }
        printf("\n");
    }
    return 0;
}
