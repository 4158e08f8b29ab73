// Dependent-chain latency of the float64 / float32 vector ops the serial recurrences are made of (one wave, wall_clock64 ticks
// at 100 MHz and s_memtime shader cycles).   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dev/dp_latency.hip -o /tmp/dp && /tmp/dp
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void chain(double* out, long long* cyc, double a, double b, int n) {
    double x = out[threadIdx.x], y = out[threadIdx.x + 64];
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (OP == 0) x = x + a;                                  // v_add_f64
            else if (OP == 1) x = x * a;                             // v_mul_f64
            else if (OP == 2) x = __builtin_fma(x, a, b);            // v_fma_f64
            else if (OP == 3) x = fmin(fmax(x, a), b);               // v_max_f64 + v_min_f64
            else if (OP == 4) {                                      // one PD + plant step (7 dependent DP ops on qd)
                double u = a * (b - y) + a * (b - x);
                u = fmin(fmax(u, -1.0), 1.0);
                x = x + 0.02 * u;
                y = y + 0.02 * x;
            } else if (OP == 5) {                                    // fp32 add chain
                float f = (float)x; f = f + (float)a; x = f;
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name, int ops_per_iter) {
    double* d; long long* c;
    hipMalloc(&d, 128 * 8); hipMalloc(&c, 8);
    hipMemset(d, 0, 128 * 8);
    const int n = 4096;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, d, c, 1.0000001, 0.5, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, d, c, 1.0000001, 0.5, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double nops = (double)n * 16 * ops_per_iter;
    printf("%-28s %8.2f ns per dependent op   (%lld counter ticks, %.3f ms, %.2f ticks/op)\n", name, ms * 1e6 / nops, cy, ms,
           cy / nops);
}

int main() {
    run<0>("v_add_f64", 1);
    run<1>("v_mul_f64", 1);
    run<2>("v_fma_f64", 1);
    run<3>("v_max_f64+v_min_f64 (pair)", 1);
    run<4>("PD+plant step (whole step)", 1);
    run<5>("f32 add via cvt", 1);
    return 0;
}
