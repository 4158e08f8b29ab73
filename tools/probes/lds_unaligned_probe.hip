// LDS vector stores at 4-byte-aligned (not 16 / 8-byte-aligned) addresses: correct on this box (alignment mode)?  And what do they
// cost against aligned ones and against four ds_write_b32?   hipcc --offload-arch=gfx950 -O3 ... && ./a.out
// Layout under test: the [t][d] trajectory image with D = 7 floats per row -- lane l writes 4 (or 3) floats at float offset 7 l.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ void wr(unsigned ad, float b) {
    if (MODE == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(f32x4{b, b + 1, b + 2, b + 3}));
    else if (MODE == 1) asm volatile("ds_write_b96 %0, %1" ::"v"(ad), "v"(f32x3{b, b + 1, b + 2}));
    else if (MODE == 2) asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(f32x2{b, b + 1}));
    else {
        asm volatile("ds_write_b32 %0, %1" ::"v"(ad), "v"(b));
        asm volatile("ds_write_b32 %0, %1 offset:4" ::"v"(ad), "v"(b + 1));
        asm volatile("ds_write_b32 %0, %1 offset:8" ::"v"(ad), "v"(b + 2));
        asm volatile("ds_write_b32 %0, %1 offset:12" ::"v"(ad), "v"(b + 3));
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int pitch, int shift, long long* cyc, int reps) {
    __shared__ __attribute__((aligned(16))) float s[4 * (64 * 8 + 64)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* sw = s + wave * (64 * 8 + 64);
    for (int i = lane; i < 64 * 8 + 64; i += 64) sw[i] = -1.f;
    __syncthreads();
    const unsigned ad = (unsigned)(size_t)(sw + lane * pitch + shift);
    const float b = 100.f * lane;
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int u = 0; u < 8; ++u) wr<MODE>(ad, b);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0) *cyc = t1 - t0;
    if (wave == 0) for (int i = lane; i < 64 * 8 + 64; i += 64) out[i] = sw[i];
}

template <int MODE>
static int run(const char* name, int n, float* d, long long* c) {
    float h[64 * 8 + 64];
    for (int pitch : {7, 8})
        for (int shift = 0; shift < 4; shift += (pitch == 8 ? 4 : 1)) {
            const int reps = 512;
            hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, d, pitch, shift, c, reps);
            CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
            long long cy; CK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < n && e < pitch; ++e) {
                    // a later lane's write may cover the tail of an earlier one only if n > pitch: never here
                    if (h[l * pitch + shift + e] != 100.f * l + e) ++bad;
                }
            printf("| %s | pitch %d floats, shift %d | %s | %.1f cycles per wave-instruction%s (4 waves on the CU) |\n", name, pitch, shift,
                   bad ? "WRONG" : "ok", (double)cy / (reps * 8), MODE == 3 ? " group of four" : "");
        }
    return 0;
}

int main() {
    float* d; long long* c;
    CK(hipMalloc(&d, 4 * (64 * 8 + 64))); CK(hipMalloc(&c, 8));
    printf("| store | address pattern | bytes | cost |\n|---|---|---|---|\n");
    run<0>("ds_write_b128", 4, d, c);
    run<1>("ds_write_b96", 3, d, c);
    run<2>("ds_write_b64", 2, d, c);
    run<3>("4 x ds_write_b32", 4, d, c);
    return 0;
}
