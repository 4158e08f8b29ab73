// One wave running the closed-loop tile recurrence in the forms the kernels use, cycles per step (s_memtime):
//   v1: desired pos / vel as fp32 in LDS (32 reads up front), convert per step, action converted and written per step
//   v2: desired pos / vel as float64 in LDS, four steps fetched ahead, float64 action written per step
//   v0: the bare chain on registers (no LDS, no conversions)
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void step(double dp, double dv, double pg, double dg, double lo, double hi, double dt, double& qs,
                                     double& qds, double& u) {
    u = pg * (dp - qs) + dg * (dv - qds);
    u = fmin(fmax(u, lo), hi);
    qds = qds + dt * u;
    qs = qs + dt * qds;
}

template <int V>
__global__ void k(double* out, long long* cyc, int ntiles, double pg, double dg, double dt, int nlanes) {
    __shared__ float sF[3 * 256];
    __shared__ double sD[2 * 256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 768; i += 64) sF[i] = 0.001f * i;
    for (int i = lane; i < 512; i += 64) sD[i] = 0.001 * i;
    __syncthreads();
    double qs = out[lane], qds = out[lane + 64];
    const double lo = -1.0, hi = 1.0;
    const int col = lane & 15;
    long long t0 = __builtin_readcyclecounter();
    if (lane < nlanes) {
        for (int it = 0; it < ntiles; ++it) {
            if (V == 0) {
#pragma unroll
                for (int tl = 0; tl < 16; ++tl) { double u; step(0.3, 0.1, pg, dg, lo, hi, dt, qs, qds, u); }
            } else if (V == 1) {
                float pr[16], vr[16];
#pragma unroll
                for (int tl = 0; tl < 16; ++tl) { pr[tl] = sF[col + tl * 7]; vr[tl] = sF[256 + col + tl * 7]; }
#pragma unroll
                for (int tl = 0; tl < 16; ++tl) {
                    double u; step((double)pr[tl], (double)vr[tl], pg, dg, lo, hi, dt, qs, qds, u);
                    sF[512 + col + tl * 7] = (float)u;
                }
            } else {
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    double dp[4], dv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { dp[i] = sD[col + (4 * ch + i) * 16]; dv[i] = sD[256 + col + (4 * ch + i) * 16]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double u; step(dp[i], dv[i], pg, dg, lo, hi, dt, qs, qds, u);
                        sD[col + (4 * ch + i) * 16] = u;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[lane] = qs + qds;
    if (lane == 0) cyc[0] = t1 - t0;
}

template <int V>
void run(const char* name, int nlanes) {
    double* d; long long* c;
    hipMalloc(&d, 128 * 8); hipMalloc(&c, 8);
    hipMemset(d, 0, 128 * 8);
    const int n = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d, c, n, 1.2, 0.1, 0.02, nlanes);
    hipDeviceSynchronize();
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%-44s lanes %2d: %7.1f ticks per step (%6.0f per 16-step tile)\n", name, nlanes, cy / (16.0 * n), (double)cy / n);
}

int main() {
    for (int nl : {16, 32, 64}) {
        run<0>("v0 bare chain", nl);
        run<1>("v1 fp32 LDS in, cvt per step, fp32 LDS out", nl);
        run<2>("v2 f64 LDS in (4 ahead), f64 LDS out", nl);
    }
    return 0;
}
