// Store-pattern ablation on MI355X: which global-store shapes reach the HBM write ceiling?
// build: hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o gpurun_out/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// V0: fill-like: every lane stores 16 B, wave covers 1 KiB contiguous, aligned
__global__ void k_v0(float4* out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = make_float4(1, 2, 3, 4);
}
// V1: trajectory-shaped: item = (group of 2 episodes, row tile): 2 pieces of 448 B at b*2800 + rt*448, 56 lanes active,
//     NARR arrays, wave walks items with stride (items of one episode are spread over waves)
template <int NARR, bool CONTIG>
__global__ void k_v1(float* o0, float* o1, float* o2, int G) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wave, Wn = gridDim.x * 4;
    const int sseg = lane / 28, w4 = (lane - sseg * 28) * 4;
    const bool act = sseg < 2;
    const float4 v = make_float4(1, 2, 3, 4);
    if (!CONTIG) {
        const int rt = wid % 7, gstride = Wn / 7;
        const int rows = rt == 6 ? 4 : 16;
        for (int g = wid / 7; g < G; g += gstride) {
            const size_t gb = ((size_t)g * 2 * 100 + rt * 16) * 7 + (size_t)sseg * 700 + w4;
            if (act && w4 < rows * 7) {
                *reinterpret_cast<float4*>(o0 + gb) = v;
                if (NARR > 1) *reinterpret_cast<float4*>(o1 + gb) = v;
                if (NARR > 2) *reinterpret_cast<float4*>(o2 + gb) = v;
            }
        }
    } else {
        for (int g = wid; g < G; g += Wn) {
            for (int rt = 0; rt < 7; ++rt) {
                const int rows = rt == 6 ? 4 : 16;
                const size_t gb = ((size_t)g * 2 * 100 + rt * 16) * 7 + (size_t)sseg * 700 + w4;
                if (act && w4 < rows * 7) {
                    *reinterpret_cast<float4*>(o0 + gb) = v;
                    if (NARR > 1) *reinterpret_cast<float4*>(o1 + gb) = v;
                    if (NARR > 2) *reinterpret_cast<float4*>(o2 + gb) = v;
                }
            }
        }
    }
}
// V2: each wave writes whole episodes as a flat contiguous stream: 175 float4 per episode per array
template <int NARR>
__global__ void k_v2(float* o0, float* o1, float* o2, int Bn) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wave, Wn = gridDim.x * 4;
    const float4 v = make_float4(1, 2, 3, 4);
    for (int b = wid; b < Bn; b += Wn) {
        for (int i = lane; i < 175; i += 64) {
            const size_t gb = (size_t)b * 700 + i * 4;
            *reinterpret_cast<float4*>(o0 + gb) = v;
            if (NARR > 1) *reinterpret_cast<float4*>(o1 + gb) = v;
            if (NARR > 2) *reinterpret_cast<float4*>(o2 + gb) = v;
        }
    }
}
// V3: a block (256 threads) writes 16 consecutive episodes of one array as one flat stream (fully coalesced, 44.8 KB)
template <int NARR>
__global__ void k_v3(float* o0, float* o1, float* o2, int Bn) {
    const float4 v = make_float4(1, 2, 3, 4);
    const int nchunk = (Bn + 15) / 16;
    for (int c = blockIdx.x; c < nchunk; c += gridDim.x) {
        const size_t base = (size_t)c * 16 * 700;
        const int n4 = min(16, Bn - c * 16) * 175;
        for (int i = threadIdx.x; i < n4; i += 256) {
            *reinterpret_cast<float4*>(o0 + base + (size_t)i * 4) = v;
            if (NARR > 1) *reinterpret_cast<float4*>(o1 + base + (size_t)i * 4) = v;
            if (NARR > 2) *reinterpret_cast<float4*>(o2 + base + (size_t)i * 4) = v;
        }
    }
}

// V4: block = episode pair (group), wave w writes row tiles {2w, 2w+1}; optional XCD-contiguous block remap
template <int NARR, bool XCD>
__global__ void k_v4(float* o0, float* o1, float* o2, int G) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sseg = lane / 28, w4 = (lane - sseg * 28) * 4;
    const bool act = sseg < 2;
    const float4 v = make_float4(1, 2, 3, 4);
    const int nb8 = gridDim.x >> 3;
    const int vb = XCD ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    for (int g = vb; g < G; g += gridDim.x) {
        for (int rt = 2 * wave; rt < min(7, 2 * wave + 2); ++rt) {
            const int rows = rt == 6 ? 4 : 16;
            const size_t gb = ((size_t)g * 2 * 100 + rt * 16) * 7 + (size_t)sseg * 700 + w4;
            if (act && w4 < rows * 7) {
                *reinterpret_cast<float4*>(o0 + gb) = v;
                if (NARR > 1) *reinterpret_cast<float4*>(o1 + gb) = v;
                if (NARR > 2) *reinterpret_cast<float4*>(o2 + gb) = v;
            }
        }
    }
}

template <typename F>
float timeit(F f, int n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / n * 1e-3f;
}

int main(int argc, char** argv) {
    const int Bn = argc > 1 ? atoi(argv[1]) : 1048576;
    const size_t n = (size_t)Bn * 700;
    float *o0, *o1, *o2;
    CK(hipMalloc(&o0, n * 4)); CK(hipMalloc(&o1, n * 4)); CK(hipMalloc(&o2, n * 4));
    const int G = Bn / 2;
    const int reps = Bn > 100000 ? 10 : 200;
    auto rep = [&](const char* name, double bytes, float t) { printf("%-44s B=%8d  %8.1f us  %7.0f GB/s\n", name, Bn, t * 1e6, bytes / t / 1e9); };
    const int blocks7 = 2044 / 7 * 7;
    rep("V0 fill float4 1 array", n * 4.0, timeit([&] { hipLaunchKernelGGL(k_v0, dim3(2048), dim3(256), 0, 0, (float4*)o0, n / 4); }, reps));
    rep("V1 tile-strided 1 array", n * 4.0, timeit([&] { hipLaunchKernelGGL((k_v1<1, false>), dim3(blocks7), dim3(256), 0, 0, o0, o1, o2, G); }, reps));
    rep("V1 tile-strided 3 arrays", n * 12.0, timeit([&] { hipLaunchKernelGGL((k_v1<3, false>), dim3(blocks7), dim3(256), 0, 0, o0, o1, o2, G); }, reps));
    rep("V1 tile-contig(wave=episode pair) 3 arrays", n * 12.0, timeit([&] { hipLaunchKernelGGL((k_v1<3, true>), dim3(2048), dim3(256), 0, 0, o0, o1, o2, G); }, reps));
    rep("V4 block=episode pair, wave=2 tiles, 3 arr", n * 12.0, timeit([&] { hipLaunchKernelGGL((k_v4<3, false>), dim3(2048), dim3(256), 0, 0, o0, o1, o2, G); }, reps));
    rep("V4 same + XCD-contiguous remap, 3 arr", n * 12.0, timeit([&] { hipLaunchKernelGGL((k_v4<3, true>), dim3(2048), dim3(256), 0, 0, o0, o1, o2, G); }, reps));
    rep("V4 same + XCD remap, 2 arr", n * 8.0, timeit([&] { hipLaunchKernelGGL((k_v4<2, true>), dim3(2048), dim3(256), 0, 0, o0, o1, o2, G); }, reps));
    rep("V2 wave=episode flat 1 array", n * 4.0, timeit([&] { hipLaunchKernelGGL((k_v2<1>), dim3(2048), dim3(256), 0, 0, o0, o1, o2, Bn); }, reps));
    rep("V2 wave=episode flat 3 arrays", n * 12.0, timeit([&] { hipLaunchKernelGGL((k_v2<3>), dim3(2048), dim3(256), 0, 0, o0, o1, o2, Bn); }, reps));
    rep("V3 block=16 episodes flat 1 array", n * 4.0, timeit([&] { hipLaunchKernelGGL((k_v3<1>), dim3(2048), dim3(256), 0, 0, o0, o1, o2, Bn); }, reps));
    rep("V3 block=16 episodes flat 3 arrays", n * 12.0, timeit([&] { hipLaunchKernelGGL((k_v3<3>), dim3(2048), dim3(256), 0, 0, o0, o1, o2, Bn); }, reps));
    return 0;
}
