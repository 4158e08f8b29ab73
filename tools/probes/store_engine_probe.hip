// Pure-store probe for the HBM-streaming launches (round 4): what does a PERSISTENT store engine reach against short-lived fill
// workgroups, by waves per CU, run length and source of the data (registers / LDS)?  No compute, no loads.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_engine_probe.hip -o /tmp/sep && /tmp/sep [B]
// Three output arrays of B x 700 floats (cfg2: 7 DoF x 100 steps), as the fused step writes them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Arr { float* p[3]; long n4; };   // n4: float4 per array

// A: short-lived workgroups, each thread PER float4 per array (the shape of an elementwise fill): block b covers chunk b
template <int PER>
__global__ void __launch_bounds__(256) k_short(Arr a, int narr) {
    const long base = (long)blockIdx.x * 256 * PER + threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (int j = 0; j < narr; ++j)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const long i = base + (long)u * 256;
            if (i < a.n4) reinterpret_cast<f32x4*>(a.p[j])[i] = v;
        }
}

// B: persistent grid-stride
__global__ void __launch_bounds__(256) k_stride(Arr a, int narr) {
    const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += stride)
        for (int j = 0; j < narr; ++j) reinterpret_cast<f32x4*>(a.p[j])[i] = v;
}

// C: persistent store engine: one workgroup of NS waves per CU; batch = run4 float4 per array (contiguous); batch b -> workgroup
// b % gridDim.x (compact front) or contiguous ranges per workgroup (chunked = 1); wave s takes the 1 KB chunks s, s + NS, ...
// SRC: 0 registers, 1 LDS (ds_read_b128 of a resident image), U = reads / stores per round
template <int SRC, int U>
__global__ void __launch_bounds__(1024) k_engine(Arr a, int narr, int run4, int chunked, int array_major, int prio) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NS = blockDim.x >> 6;
    if (SRC == 1) {
        for (int i = threadIdx.x; i < run4 * 4 * 3; i += blockDim.x) lds[i] = (float)i;
        __syncthreads();
    }
    if (prio) __builtin_amdgcn_s_setprio(3);
    const long nb = (a.n4 + run4 - 1) / run4;
    const long per = (nb + gridDim.x - 1) / gridDim.x;
    const f32x4 c = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (long it = 0; it < per; ++it) {
        const long b = chunked ? (long)blockIdx.x * per + it : it * gridDim.x + blockIdx.x;
        if (b >= nb) break;
        const long i0 = b * run4;
        const int n4 = (int)((a.n4 - i0) < run4 ? (a.n4 - i0) : run4);
        if (array_major) {
            for (int j = 0; j < narr; ++j) {
                f32x4* out = reinterpret_cast<f32x4*>(a.p[j]) + i0;
                const f32x4* src = reinterpret_cast<const f32x4*>(lds + (size_t)j * run4 * 4);
                for (int k0 = wave * 64; k0 < n4; k0 += 64 * NS * U) {
                    f32x4 v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int idx = k0 + u * 64 * NS + lane;
                        v[u] = c;
                        if (SRC == 1 && idx < n4) v[u] = src[idx];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int idx = k0 + u * 64 * NS + lane;
                        if (idx < n4) out[idx] = v[u];
                    }
                }
            }
        } else {   // arrays interleaved per 1 KB chunk
            for (int k0 = wave * 64; k0 < n4; k0 += 64 * NS * U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = k0 + u * 64 * NS + lane;
                    if (idx < n4)
                        for (int j = 0; j < narr; ++j) {
                            f32x4 v = c;
                            if (SRC == 1) v = reinterpret_cast<const f32x4*>(lds + (size_t)j * run4 * 4)[idx];
                            (reinterpret_cast<f32x4*>(a.p[j]) + i0)[idx] = v;
                        }
                }
            }
        }
    }
}

// D: persistent workgroups that take batches from ONE device-wide counter (the dispatcher's in-order assignment, emulated):
// whatever the waves' relative progress, batches are started in address order
template <int SRC, int U>
__global__ void __launch_bounds__(1024) k_dynamic(Arr a, int narr, int run4, unsigned* ctr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ long sb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NS = blockDim.x >> 6;
    if (SRC == 1) {
        for (int i = threadIdx.x; i < run4 * 4 * 3; i += blockDim.x) lds[i] = (float)i;
        __syncthreads();
    }
    const long nb = (a.n4 + run4 - 1) / run4;
    const f32x4 c = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (;;) {
        if (threadIdx.x == 0) sb = (long)atomicAdd(ctr, 1u);
        __syncthreads();
        const long b = sb;
        __syncthreads();
        if (b >= nb) break;
        const long i0 = b * run4;
        const int n4 = (int)((a.n4 - i0) < run4 ? (a.n4 - i0) : run4);
        for (int j = 0; j < narr; ++j) {
            f32x4* out = reinterpret_cast<f32x4*>(a.p[j]) + i0;
            const f32x4* src = reinterpret_cast<const f32x4*>(lds + (size_t)j * run4 * 4);
            for (int k0 = wave * 64; k0 < n4; k0 += 64 * NS * U) {
                f32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = k0 + u * 64 * NS + lane;
                    v[u] = c;
                    if (SRC == 1 && idx < n4) v[u] = src[idx];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = k0 + u * 64 * NS + lane;
                    if (idx < n4) out[idx] = v[u];
                }
            }
        }
    }
}

template <class F>
static double timeit(F f, int reps = 10) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1e-3 / reps;
}

int main(int argc, char** argv) {
    const long B = argc > 1 ? atol(argv[1]) : 262144;
    Arr a;
    a.n4 = B * 700 / 4;
    for (int j = 0; j < 3; ++j) CK(hipMalloc(&a.p[j], a.n4 * 16));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double bytes3 = 3.0 * a.n4 * 16, bytes1 = 1.0 * a.n4 * 16;
    printf("# B = %ld, %d CUs, %.2f GB per 3-array pass\n", B, cus, bytes3 / 1e9);
    printf("| variant | us | TB/s | of 8 TB/s |\n|---|---|---|---|\n");
    auto row = [&](const std::string& name, double t, double bytes) {
        printf("| %s | %.1f | %.2f | %.1f %% |\n", name.c_str(), t * 1e6, bytes / t / 1e12, bytes / t / 8e12 * 100);
        fflush(stdout);
    };
    // clocks
    timeit([&] { hipLaunchKernelGGL(k_short<1>, dim3((a.n4 + 255) / 256), dim3(256), 0, 0, a, 3); }, 30);
    const bool more = argc > 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_engine<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_engine<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        row("hipMemsetAsync x 3", timeit([&] { for (int j = 0; j < 3; ++j) CK(hipMemsetAsync(a.p[j], 0, a.n4 * 16, 0)); }), bytes3);
        row("short-lived, 1 float4 / thread, 1 array x 3 launches", timeit([&] { for (int j = 0; j < 3; ++j) { Arr b = a; b.p[0] = a.p[j]; hipLaunchKernelGGL(k_short<1>, dim3((a.n4 + 255) / 256), dim3(256), 0, 0, b, 1); } }), bytes3);
        row("short-lived, 1 float4 / thread, 3 arrays in one launch", timeit([&] { hipLaunchKernelGGL(k_short<1>, dim3((a.n4 + 255) / 256), dim3(256), 0, 0, a, 3); }), bytes3);
        if (!more) {
        row("short-lived, 4 float4 / thread (16 KB per block and array), 3 arrays", timeit([&] { hipLaunchKernelGGL(k_short<4>, dim3((a.n4 + 1023) / 1024), dim3(256), 0, 0, a, 3); }), bytes3);
        row("short-lived, 8 float4 / thread (32 KB per block and array), 3 arrays", timeit([&] { hipLaunchKernelGGL(k_short<8>, dim3((a.n4 + 2047) / 2048), dim3(256), 0, 0, a, 3); }), bytes3);
        row("short-lived, 4 float4 / thread, 1 array", timeit([&] { hipLaunchKernelGGL(k_short<4>, dim3((a.n4 + 1023) / 1024), dim3(256), 0, 0, a, 1); }), bytes1);
        for (int wgs : {8, 4, 2})
            row("persistent grid-stride, " + std::to_string(wgs) + " x 256-thread workgroups per CU, 3 arrays",
                timeit([&] { hipLaunchKernelGGL(k_stride, dim3(cus * wgs), dim3(256), 0, 0, a, 3); }), bytes3);
        }
        const int run4 = 1400;   // 22.4 KB per array: four cfg2 episode pairs
        const size_t lds = (size_t)run4 * 16 * 3;
        if (!more) {
        for (int ns : {1, 2, 4, 8, 16})
            row("engine, registers, " + std::to_string(ns) + " waves / CU, 22.4 KB runs, array-major, compact front",
                timeit([&] { hipLaunchKernelGGL((k_engine<0, 8>), dim3(cus), dim3(64 * ns), 0, 0, a, 3, run4, 0, 1, 0); }), bytes3);
        for (int ns : {1, 2, 4, 8, 16})
            row("engine, LDS, " + std::to_string(ns) + " waves / CU, 22.4 KB runs, array-major, compact front",
                timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(64 * ns), lds, 0, a, 3, run4, 0, 1, 0); }), bytes3);
        row("engine, LDS, 4 waves / CU, prio 3", timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(256), lds, 0, a, 3, run4, 0, 1, 1); }), bytes3);
        row("engine, LDS, 4 waves / CU, U = 4", timeit([&] { hipLaunchKernelGGL((k_engine<1, 4>), dim3(cus), dim3(256), lds, 0, a, 3, run4, 0, 1, 0); }), bytes3);
        row("engine, LDS, 4 waves / CU, arrays interleaved per chunk", timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(256), lds, 0, a, 3, run4, 0, 0, 0); }), bytes3);
        row("engine, LDS, 4 waves / CU, chunked ranges per workgroup", timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(256), lds, 0, a, 3, run4, 1, 1, 0); }), bytes3);
        row("engine, LDS, 4 waves / CU, 2 workgroups per CU", timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus * 2), dim3(256), lds, 0, a, 3, run4, 0, 1, 0); }), bytes3);
        row("engine, registers, 4 waves / CU, 8 workgroups per CU", timeit([&] { hipLaunchKernelGGL((k_engine<0, 8>), dim3(cus * 8), dim3(256), 0, 0, a, 3, run4, 0, 1, 0); }), bytes3);
        row("engine, registers, 4 waves / CU, 5.6 KB runs", timeit([&] { hipLaunchKernelGGL((k_engine<0, 8>), dim3(cus), dim3(256), 0, 0, a, 3, 350, 0, 1, 0); }), bytes3);
        row("engine, registers, 8 waves / CU, 89.6 KB runs", timeit([&] { hipLaunchKernelGGL((k_engine<0, 8>), dim3(cus), dim3(512), 0, 0, a, 3, 5600, 0, 1, 0); }), bytes3);
        } else if (argv[2][0] == 'd') {
        unsigned* ctr;
        CK(hipMalloc(&ctr, 4));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dynamic<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        row("chunked, LDS, 4 waves / CU, 22.4 KB runs, array-major (static ranges)",
            timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(256), lds, 0, a, 3, run4, 1, 1, 0); }), bytes3);
        row("compact static (b % nWG), LDS, 4 waves / CU, 22.4 KB runs",
            timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(256), lds, 0, a, 3, run4, 0, 1, 0); }), bytes3);
        for (int r4 : {350, 1400, 2800})
            for (int ns : {1, 4})
                for (int w : {1, 2})
                    row("DYNAMIC (one counter), LDS, " + std::to_string(ns) + " waves x " + std::to_string(w) + " workgroups / CU, " + std::to_string(r4 * 16 / 1000.0).substr(0, 4) + " KB runs",
                        timeit([&] { CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL((k_dynamic<1, 8>), dim3(cus * w), dim3(64 * ns), (size_t)r4 * 48, 0, a, 3, r4, ctr); }), bytes3);
        for (int w : {4, 8, 16})
            row("DYNAMIC, registers, 1 wave x " + std::to_string(w) + " workgroups / CU, 5.6 KB runs",
                timeit([&] { CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL((k_dynamic<0, 8>), dim3(cus * w), dim3(64), 0, 0, a, 3, 350, ctr); }), bytes3);
        for (int w : {8, 16, 32})
            row("DYNAMIC, registers, 1 wave x " + std::to_string(w) + " workgroups / CU, 1 KB runs",
                timeit([&] { CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL((k_dynamic<0, 1>), dim3(cus * w), dim3(64), 0, 0, a, 3, 64, ctr); }), bytes3);
        } else {
        // CHUNKED ranges (workgroup w owns batches [w * per, (w + 1) * per)): by run length, waves, workgroups per CU, order
        for (int r4 : {350, 700, 1400, 2800})
            for (int ns : {2, 4, 8})
                row("chunked, LDS, " + std::to_string(ns) + " waves / CU, " + std::to_string(r4 * 16 / 1000.0).substr(0, 4) + " KB runs, array-major",
                    timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(64 * ns), (size_t)r4 * 48, 0, a, 3, r4, 1, 1, 0); }), bytes3);
        for (int r4 : {350, 1400})
            row("chunked, LDS, 4 waves / CU, " + std::to_string(r4 * 16 / 1000.0).substr(0, 4) + " KB runs, arrays interleaved per chunk",
                timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(256), (size_t)r4 * 48, 0, a, 3, r4, 1, 0, 0); }), bytes3);
        for (int w : {2, 4, 8})
            row("chunked, registers, 4 waves, " + std::to_string(w) + " workgroups / CU, 22.4 KB runs, array-major",
                timeit([&] { hipLaunchKernelGGL((k_engine<0, 8>), dim3(cus * w), dim3(256), 0, 0, a, 3, run4, 1, 1, 0); }), bytes3);
        for (int w : {2, 3})
            row("chunked, LDS, 4 waves, " + std::to_string(w) + " workgroups / CU, 22.4 KB runs, array-major",
                timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus * w), dim3(256), lds, 0, a, 3, run4, 1, 1, 0); }), bytes3);
        row("chunked, LDS, 4 waves / CU, 22.4 KB runs, 2 arrays (trajectory only)",
            timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(256), lds, 0, a, 2, run4, 1, 1, 0); }), bytes1 * 2);
        row("chunked, LDS, 4 waves / CU, 22.4 KB runs, 1 array",
            timeit([&] { hipLaunchKernelGGL((k_engine<1, 8>), dim3(cus), dim3(256), lds, 0, a, 1, run4, 1, 1, 0); }), bytes1);
        }
    }
    return 0;
}
