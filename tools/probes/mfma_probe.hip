// v_mfma_f32_16x16x4_f32 issue-rate probe: NACC independent accumulators per wave, W waves per SIMD, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_probe tools/probes/mfma_probe.hip && build/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = a0 + i * 1e-6f;
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    const float* pa = sm + (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            f32x4 af[(NACC + 3) / 4];
#pragma unroll
            for (int c = 0; c < (NACC + 3) / 4; ++c) af[c] = *reinterpret_cast<const f32x4*>(pa + 4 * c + ((it & 7) << 10) % 2048);
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i >> 2][i & 3], b, acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(int wg_per_cu, const char* name) {
    int cus = 256;
    float* out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, LDS>), dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)cus * wg_per_cu * 4 * iters * NACC * 2048.0;
    printf("| %s | %d | %d | %.2f ms | %.1f TF |\n", name, NACC, wg_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    printf("| operands | accumulators per wave | waves per SIMD | time | fp32 MFMA rate (16x16x4) |\n|---|---|---|---|---|\n");
    run<13, false>(1, "registers"); run<13, false>(2, "registers"); run<4, false>(1, "registers"); run<4, false>(2, "registers");
    run<16, false>(2, "registers"); run<1, false>(2, "registers"); run<1, false>(4, "registers");
    run<13, true>(1, "A from LDS (ds_read_b128 per 4)"); run<13, true>(2, "A from LDS (ds_read_b128 per 4)");
    return 0;
}
