// Pure-store floor of the headline launch (cfg2, B = 4096: 3 arrays x 4096 x 2800 B = 34.4 MB), timed like bench.py:
// 200 launches captured in one hipGraph, HIP events around the replay (eager back-to-back launches of a ~7 us kernel
// measure the host's launch rate, not the kernel).  Same geometry as k_traj_tiles: 7168 waves, two items per wave,
// a wave owns row tile wid % 7 of groups wid / 7 and wid / 7 + gstride, XCD-contiguous block remap, 56 lanes x 16 B per
// array and item.   build: hipcc --offload-arch=gfx950 -O3 tools/store_probe2.hip -o /tmp/store_probe2 && /tmp/store_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 plain, 1 sc1, 2 sc0 sc1, 3 nt
__device__ __forceinline__ void st(float* p, const f32x4& v) {
    if (MODE == 0) *reinterpret_cast<f32x4*>(p) = v;
    else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE, bool XCD>
__global__ void __launch_bounds__(256) k_tiles(float* o0, float* o1, float* o2, int G, int gstride) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb8 = gridDim.x >> 3;
    const int vb = XCD && (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int wid = vb * 4 + wave;
    const int sseg = lane / 28, w4 = (lane - sseg * 28) * 4;
    const int rt = wid % 7, rows = rt == 6 ? 4 : 16;
    const f32x4 v = {1.f, 2.f, 3.f, (float)wid};
    for (int g = wid / 7; g < G; g += gstride) {
        const size_t gb = ((size_t)g * 2 * 100 + rt * 16) * 7 + (size_t)sseg * 700 + w4;
        if (sseg < 2 && w4 < rows * 7) { st<MODE>(o0 + gb, v); st<MODE>(o1 + gb, v); st<MODE>(o2 + gb, v); }
    }
}

// the tile pattern with the kernel's input round trip in front of the stores: every lane gathers 2 parameter words, the
// boundary state (2 floats, 2 doubles) and 4 basis-fragment words -- the loads k_traj_tiles issues -- and the stored
// values depend on them, but there is no MFMA / epilogue / LDS transpose
template <int MODE>
__global__ void __launch_bounds__(256) k_tiles_loads(float* o0, float* o1, float* o2, const float* params, const float* ip,
                                                     const float* iv, const double* cp, const double* cv, const float* A,
                                                     int G, int gstride) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int wid = vb * 4 + wave;
    const int sseg = lane / 28, w4 = (lane - sseg * 28) * 4;
    const int rt = wid % 7, rows = rt == 6 ? 4 : 16;
    const int col = lane & 15, q = lane >> 4, bl = col >> 3, d = (col & 7) < 7 ? (col & 7) : 6;
    const float a0 = A[(q) * 112 + rt * 16 + col], a1 = A[(4 + q) * 112 + rt * 16 + col];
    const float a2 = A[(8 + q) * 112 + rt * 16 + col], a3 = A[(12 + q) * 112 + rt * 16 + col];
    for (int g = wid / 7; g < G; g += gstride) {
        const size_t b = (size_t)2 * g + bl;
        const float p0 = params[b * 42 + d * 6 + (q < 2 ? q : 0)], p1 = params[b * 42 + d * 6 + 4 + (q & 1)];
        const float x = ip[b * 7 + d] + iv[b * 7 + d] + (float)(cp[b * 7 + d] + cv[b * 7 + d]);
        const f32x4 v = {p0 * a0, p1 * a1, x * a2, a3};
        const size_t gb = ((size_t)g * 2 * 100 + rt * 16) * 7 + (size_t)sseg * 700 + w4;
        if (sseg < 2 && w4 < rows * 7) { st<MODE>(o0 + gb, v); st<MODE>(o1 + gb, v); st<MODE>(o2 + gb, v); }
    }
}

// workgroup-cooperative variant: a workgroup owns (group g, row tiles 0..3) or (group g, row tiles 4..6) and its 256 lanes
// store each (episode, array) run of 4 x 448 = 1792 B (or 1008 B for the second half) as consecutive 16-byte chunks
template <int MODE>
__global__ void __launch_bounds__(256) k_coop(float* o0, float* o1, float* o2, int G, int wgstride) {
    const int half = blockIdx.x & 1;                      // 0: rows 0..63, 1: rows 64..99
    const int chunks = half ? 63 : 112;                   // float4 per (episode, array) run: 36 x 7 / 4, 64 x 7 / 4
    const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    float* const arr[3] = {o0, o1, o2};
    for (int g = blockIdx.x >> 1; g < G; g += wgstride) {
        for (int c = threadIdx.x; c < 6 * chunks; c += 256) {
            const int run = c / chunks, k = c - run * chunks;            // run = array * 2 + episode
            const size_t base = ((size_t)(2 * g + (run & 1)) * 100 + half * 64) * 7;
            st<MODE>(arr[run >> 1] + base + 4 * k, v);
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_fill(float* o, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) st<MODE>(o + 4 * i, v);
}

__global__ void k_empty() {}

template <typename F>
static float graph_time(F launch, hipStream_t s, int n) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 5; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best / n * 1e-3f;
}

int main() {
    const int Bn = 4096, G = Bn / 2;
    const size_t n = (size_t)Bn * 700;
    float* o; CK(hipMalloc(&o, 3 * n * 4));
    float *o0 = o, *o1 = o + n, *o2 = o + 2 * n;
    hipStream_t s; CK(hipStreamCreate(&s));
    const int blocks = 1792, gstride = blocks * 4 / 7;
    const double bytes = 3.0 * n * 4;
    auto rep = [&](const char* name, double by, float t) { printf("| %-58s | %6.2f us | %5.0f GB/s |\n", name, t * 1e6, by / t / 1e9); };
    rep("empty kernel (launch boundary only)", 0, graph_time([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, s, 200));
#define T(MODE, XCD, NAME) rep(NAME, bytes, graph_time([&] { hipLaunchKernelGGL((k_tiles<MODE, XCD>), dim3(blocks), dim3(256), 0, s, o0, o1, o2, G, gstride); }, s, 200))
    T(0, false, "tile pattern, plain stores");
    T(0, true,  "tile pattern, plain stores, XCD remap");
    T(1, false, "tile pattern, sc1 stores");
    T(1, true,  "tile pattern, sc1 stores, XCD remap  (= k_traj_tiles' stores)");
    T(2, true,  "tile pattern, sc0 sc1 stores, XCD remap");
    T(3, true,  "tile pattern, nt stores, XCD remap");
    {
        float *params, *ipb, *ivb, *Ab; double *cpb, *cvb;
        CK(hipMalloc(&params, (size_t)Bn * 42 * 4)); CK(hipMalloc(&ipb, (size_t)Bn * 7 * 4)); CK(hipMalloc(&ivb, (size_t)Bn * 7 * 4));
        CK(hipMalloc(&cpb, (size_t)Bn * 7 * 8)); CK(hipMalloc(&cvb, (size_t)Bn * 7 * 8)); CK(hipMalloc(&Ab, 16 * 112 * 4));
        CK(hipMemset(params, 0, (size_t)Bn * 42 * 4)); CK(hipMemset(ipb, 0, (size_t)Bn * 7 * 4)); CK(hipMemset(ivb, 0, (size_t)Bn * 7 * 4));
        CK(hipMemset(cpb, 0, (size_t)Bn * 7 * 8)); CK(hipMemset(cvb, 0, (size_t)Bn * 7 * 8)); CK(hipMemset(Ab, 0, 16 * 112 * 4));
        rep("tile pattern, sc1, XCD remap + the kernel's input loads first", bytes, graph_time([&] { hipLaunchKernelGGL((k_tiles_loads<1>), dim3(blocks), dim3(256), 0, s, o0, o1, o2, params, ipb, ivb, cpb, cvb, Ab, G, gstride); }, s, 200));
    }
    rep("workgroup-cooperative runs of 1792 / 1008 B, sc1 (2048 WGs)", bytes, graph_time([&] { hipLaunchKernelGGL((k_coop<1>), dim3(2048), dim3(256), 0, s, o0, o1, o2, G, 1024); }, s, 200));
    rep("workgroup-cooperative runs of 1792 / 1008 B, sc1 (4096 WGs)", bytes, graph_time([&] { hipLaunchKernelGGL((k_coop<1>), dim3(4096), dim3(256), 0, s, o0, o1, o2, G, 2048); }, s, 200));
    rep("flat fill of the same 34.4 MB, plain", bytes, graph_time([&] { hipLaunchKernelGGL((k_fill<0>), dim3(2048), dim3(256), 0, s, o, 3 * n / 4); }, s, 200));
    rep("flat fill of the same 34.4 MB, sc1", bytes, graph_time([&] { hipLaunchKernelGGL((k_fill<1>), dim3(2048), dim3(256), 0, s, o, 3 * n / 4); }, s, 200));
    return 0;
}
