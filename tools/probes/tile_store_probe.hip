// The store pattern of the serial-recurrence trajectory kernels (k_traj_duo / k_traj_quad) without any compute, and what
// buffering more row tiles per flush would buy: a wave owns NQ consecutive episode pairs (2 x T x D floats = one contiguous
// run per output array) and writes them in flushes of TL row tiles (TL x 16 x 2 D floats per pair and array), three arrays.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tile_store_probe.hip -o tools/probes/tile_store_probe.bin
//   ./tools/probes/tile_store_probe.bin [B=65536] [T=100] [D=7]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Arr { float* p[3]; };

// persistent = 1: units u = wave id, u += all waves; 0: one unit per wave.  nap: s_sleep ticks between flushes (stands in for
// the contraction + recurrence of the next tiles)
__global__ void __launch_bounds__(256) k_tiles(Arr a, int narr, long npairs, int pair4, int tile4, int NQ, int TL, int NRT, int persistent,
                                               int nap) {
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4* mine = lds + wave * 64;
    mine[lane] = f32x4{1.f, 2.f, 3.f, (float)lane};
    const long nunits = (npairs + NQ - 1) / NQ;
    const long stride = persistent ? (long)gridDim.x * 4 : nunits;
    for (long u = (long)blockIdx.x * 4 + wave; u < nunits; u += stride) {
        for (int rt0 = 0; rt0 < NRT; rt0 += TL) {
            const int tiles = min(TL, NRT - rt0);
            const int lo4 = rt0 * tile4;
            int hi4 = (rt0 + tiles) * tile4;
            if (hi4 > pair4) hi4 = pair4;
            for (int j = 0; j < NQ; ++j) {
                const long pr = u * NQ + j;
                if (pr >= npairs) break;
                for (int k = 0; k < narr; ++k) {
                    f32x4* dst = reinterpret_cast<f32x4*>(a.p[k]) + pr * pair4;
                    for (int i = lo4 + lane; i < hi4; i += 64) {
                        const f32x4 v = mine[(i + k) & 63];
                        __builtin_nontemporal_store(v, dst + i);
                    }
                }
            }
            for (int s = 0; s < nap; ++s) __builtin_amdgcn_s_sleep(8);
        }
    }
}

__global__ void __launch_bounds__(256) k_fill(float* p, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) reinterpret_cast<f32x4*>(p)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}

int main(int argc, char** argv) {
    const long B = argc > 1 ? atol(argv[1]) : 65536;
    const int T = argc > 2 ? atoi(argv[2]) : 100, D = argc > 3 ? atoi(argv[3]) : 7;
    const int narr = argc > 4 ? atoi(argv[4]) : 3;
    const long npairs = B / 2;
    const int pair4 = 2 * T * D / 4, tile4 = 16 * 2 * D / 4, NRT = (T + 15) / 16;
    Arr a;
    const size_t bytes = (size_t)B * T * D * 4;
    for (int k = 0; k < 3; ++k) CK(hipMalloc(&a.p[k], bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](auto&& launch) {
        std::vector<float> ts;
        for (int r = 0; r < 7; ++r) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        return ts[ts.size() / 2];
    };
    const double gb = (double)narr * bytes / 1e9;
    printf("B = %ld, T = %d, D = %d, %d arrays, %.0f MB per launch; run per (pair, array, flush) = TL x %d B\n", B, T, D, narr, gb * 1e3, tile4 * 16);
    printf("| pattern | us | TB/s |\n|---|---|---|\n");
    for (int w = 0; w < 3; ++w) {
        const float t = timed([&] { for (int k = 0; k < narr; ++k) hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, a.p[k], (long)(bytes / 16)); });
        if (w == 2) printf("| fill, one array per launch | %.1f | %.2f |\n", t, gb / t * 1e-3 * 1e3);
    }
    int dev_cu = 256;
    for (int persistent = 1; persistent >= 0; --persistent)
        for (int NQ : {2, 4})
            for (int TL : {1, 2, 4, 7})
                for (int wgs : {2, 4}) {             // resident workgroups per CU (dynamic LDS pads the occupancy)
                    if (TL > NRT) continue;
                    const size_t lds = wgs == 2 ? 70 * 1024 : 36 * 1024;
                    CK(hipFuncSetAttribute((const void*)k_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    const long nunits = (npairs + NQ - 1) / NQ;
                    const long blocks = persistent ? std::min<long>((nunits + 3) / 4, (long)dev_cu * wgs) : (nunits + 3) / 4;
                    const float t = timed([&] { hipLaunchKernelGGL(k_tiles, dim3((unsigned)blocks), dim3(256), lds, 0, a, narr, npairs, pair4, tile4, NQ, TL, NRT, persistent, 0); });
                    CK(hipGetLastError());
                    printf("| %s, %d pairs per wave, %d tile(s) per flush (%d B runs), %d workgroups per CU | %.1f | %.2f |\n",
                           persistent ? "persistent" : "short-lived", NQ, TL, TL * tile4 * 16 > pair4 * 16 ? pair4 * 16 : TL * tile4 * 16, wgs, t, gb / t);
                }
    return 0;
}
