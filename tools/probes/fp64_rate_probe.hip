// What the float64 instructions of the closed-loop step cost on one SIMD of gfx950: issue rate (32 independent instructions in a
// row, one wave alone on its SIMD) and dependent latency (a chain of 32), in shader cycles per instruction (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fp64_rate_probe.hip -o tools/probes/fp64_rate_probe.bin && ./tools/probes/fp64_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// MODE: 0 v_add_f64, 1 v_mul_f64, 2 v_fma_f64, 3 v_max_f64, 4 v_cvt_f64_f32, 5 v_cvt_f32_f64, 6 v_cndmask_b32 pair (a double select),
//       7 v_add_f32, 8 the whole controller + plant step (9-deep chain + 2 + conversions) as the kernels have it
template <int MODE, bool DEP, int LANES = 64>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, int reps, double seed) {
    const int lane = threadIdx.x;
    double a = seed + lane, b = seed * 0.5 + 1.0, c = 1.0000001;
    double r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    float f0 = (float)a, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    long long t0 = 0, t1 = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (lane < LANES)                       // EXEC = the low LANES lanes for the whole measured stream
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 0) {
            if (DEP) { REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(r0) : "v"(c));) }
            else { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));) }
        } else if (MODE == 1) {
            if (DEP) { REP32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r0) : "v"(c));) }
            else { REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));) }
        } else if (MODE == 2) {
            if (DEP) { REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r0) : "v"(c), "v"(b));) }
            else { REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c), "v"(b));) }
        } else if (MODE == 3) {
            if (DEP) { REP32(asm volatile("v_max_f64 %0, %0, %1" : "+v"(r0) : "v"(c));) }
            else { REP8(asm volatile("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));) }
        } else if (MODE == 4) {
            if (DEP) { REP32(asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0" : "+v"(r0), "+v"(f0));) }
            else { REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));) }
        } else if (MODE == 5) {
            REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));)
        } else if (MODE == 6) {
            REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f4) : );)
        } else if (MODE == 7) {
            if (DEP) { REP32(asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f4));) }
            else { REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f4));) }
        } else {
            // 32 steps of: u = pg (dp - q) + dg (dv - qd); clip; qd += dt u; q += dt qd   (r0 = q, r1 = qd; DEP: without the conversions)
            const double pg = 1.2, dg = 0.1, lo = -1.0, hi = 1.0, dt = 0.02;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                double dp, dv;
                if (DEP) { dp = r4; dv = r5; }
                else { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dp) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dv) : "v"(f1)); }
                double e0, e1, u;
                asm volatile("v_add_f64 %0, %1, -%2" : "=v"(e0) : "v"(dp), "v"(r0));
                asm volatile("v_add_f64 %0, %1, -%2" : "=v"(e1) : "v"(dv), "v"(r1));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(e0) : "v"(e0), "v"(pg));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(e1) : "v"(e1), "v"(dg));
                asm volatile("v_add_f64 %0, %1, %2" : "=v"(u) : "v"(e0), "v"(e1));
                asm volatile("v_max_f64 %0, %1, %2" : "=v"(u) : "v"(u), "v"(lo));
                asm volatile("v_min_f64 %0, %1, %2" : "=v"(u) : "v"(u), "v"(hi));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(e0) : "v"(u), "v"(dt));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(r1) : "v"(e0));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(e1) : "v"(r1), "v"(dt));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(r0) : "v"(e1));
                if (!DEP) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f2) : "v"(u));
            }
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 64 + lane] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, bool DEP, int LANES = 64>
static int run(const char* name, double* d, long long* c, int per_rep) {
    const int reps = 64;
    hipLaunchKernelGGL((k<MODE, DEP, LANES>), dim3(1), dim3(64), 0, 0, d, c, reps, 1.0);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<MODE, DEP, LANES>), dim3(1), dim3(64), 0, 0, d, c, reps, 1.0);
    CK(hipDeviceSynchronize());
    long long h = 0;
    CK(hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost));
    printf("| %s | %.2f |\n", name, (double)h / ((double)reps * per_rep));
    return 0;
}

int main() {
    double* d; long long* c;
    CK(hipMalloc(&d, 64 * 8 * 64)); CK(hipMalloc(&c, 8 * 64));
    printf("| instruction stream (one wave alone on its SIMD) | shader cycles per instruction (s_memtime ticks x clock ratio not applied) |\n|---|---|\n");
    run<0, false>("v_add_f64, independent", d, c, 32); run<0, true>("v_add_f64, dependent chain", d, c, 32);
    run<1, false>("v_mul_f64, independent", d, c, 32); run<1, true>("v_mul_f64, dependent chain", d, c, 32);
    run<2, false>("v_fma_f64, independent", d, c, 32); run<2, true>("v_fma_f64, dependent chain", d, c, 32);
    run<3, false>("v_max_f64, independent", d, c, 32); run<3, true>("v_max_f64, dependent chain", d, c, 32);
    run<4, false>("v_cvt_f64_f32, independent", d, c, 32); run<4, true>("v_cvt_f64_f32 + v_cvt_f32_f64 round trip, dependent (per pair)", d, c, 32);
    run<5, false>("v_cvt_f32_f64, independent", d, c, 32);
    run<6, false>("v_cndmask_b32, independent", d, c, 32);
    run<7, false>("v_add_f32, independent", d, c, 32); run<7, true>("v_add_f32, dependent chain", d, c, 32);
    run<8, true>("controller + plant step, float64 inputs (11 instructions, 9 dependent), per STEP", d, c, 32);
    run<8, false>("controller + plant step with its 3 conversions (14 instructions), per STEP", d, c, 32);
    run<0, true, 32>("v_add_f64, dependent chain, EXEC = lanes 0 - 31", d, c, 32);
    run<0, true, 16>("v_add_f64, dependent chain, EXEC = lanes 0 - 15", d, c, 32);
    run<0, false, 32>("v_add_f64, independent, EXEC = lanes 0 - 31", d, c, 32);
    run<0, false, 16>("v_add_f64, independent, EXEC = lanes 0 - 15", d, c, 32);
    run<8, true, 32>("controller + plant step, float64 inputs, EXEC = lanes 0 - 31, per STEP", d, c, 32);
    run<8, true, 16>("controller + plant step, float64 inputs, EXEC = lanes 0 - 15, per STEP", d, c, 32);
    run<7, true, 32>("v_add_f32, dependent chain, EXEC = lanes 0 - 31", d, c, 32);
    run<7, true, 16>("v_add_f32, dependent chain, EXEC = lanes 0 - 15", d, c, 32);
    return 0;
}
