// integer replanning state, episode reset, boundary-condition gather, validity reduction, inspection / self-test kernels
#include "mpk_dev.h"

namespace mpk {


// ------------------------------------------------------------------------------------------------------------
// integer replanning state
// ------------------------------------------------------------------------------------------------------------
// valid (optional): the verdict of the validity gate on the plan just produced -- an invalid plan finishes its episode without a step
// (done |= !valid in front of the rule: replan_write)
__global__ void __launch_bounds__(256) k_replan_advance(const ReplanDev rp, const int T, const int B, const uint8_t* __restrict__ valid) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const ReplanVals v = replan_eval(rp, b, T);
    (void)replan_write(rp, b, v, valid ? valid[b] != 0 : true);
}

#ifndef MPK_DEVICE_ONLY
int launch_replan_advance(int32_t* traj_steps, int32_t* plan_steps, int32_t* seg_len, uint8_t* done, int every,
                          int max_planning_times, int horizon, int T, int B, void* stream, const uint8_t* valid) {
    ReplanDev rp;
    rp.traj_steps = traj_steps; rp.plan_steps = plan_steps; rp.seg_len = seg_len; rp.done = done;
    rp.every = every; rp.max_planning_times = max_planning_times; rp.horizon = horizon;
    hipLaunchKernelGGL(k_replan_advance, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, rp, T, B, valid);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

// mpk_gate_flags: terminated = !valid && !was_done, truncated = done && valid (black_box_wrapper.py:169-172,198-203)
__global__ void __launch_bounds__(256) k_gate_flags(const uint8_t* __restrict__ valid, const uint8_t* __restrict__ was_done,
                                                    const uint8_t* __restrict__ done, uint8_t* __restrict__ terminated,
                                                    uint8_t* __restrict__ truncated, const int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const bool v = valid[b] != 0, w = was_done ? was_done[b] != 0 : false, d = done[b] != 0;
    terminated[b] = (!v && !w) ? 1 : 0;
    truncated[b] = (d && v) ? 1 : 0;
}

#ifndef MPK_DEVICE_ONLY
int launch_gate_flags(const uint8_t* valid, const uint8_t* was_done, const uint8_t* done, uint8_t* terminated, uint8_t* truncated, int B,
                      void* stream) {
    hipLaunchKernelGGL(k_gate_flags, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, valid, was_done, done, terminated, truncated, B);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

// BlackBoxWrapper.reset (black_box_wrapper.py:222-229) for B episodes: counters to zero, plant state from the caller's
// initial state (NULL = zeros) and its fp32 image, the boundary condition of the first plan (black_box_wrapper.py:110-111)
__global__ void __launch_bounds__(256) k_episode_reset(const double* __restrict__ init_q, const double* __restrict__ init_qd,
                                                       double* __restrict__ q, double* __restrict__ qd,
                                                       float* __restrict__ cond_pos, float* __restrict__ cond_vel,
                                                       int32_t* __restrict__ traj_steps, int32_t* __restrict__ plan_steps,
                                                       uint8_t* __restrict__ done, const int B, const int D) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < (long)B * D) {
        const double a = init_q ? init_q[e] : 0.0, b = init_qd ? init_qd[e] : 0.0;
        q[e] = a; qd[e] = b;
        if (cond_pos) { cond_pos[e] = (float)a; cond_vel[e] = (float)b; }
    }
    if (e < B) {
        traj_steps[e] = 0; plan_steps[e] = 0; done[e] = 0;
    }
}

#ifndef MPK_DEVICE_ONLY
int launch_episode_reset(const double* init_q, const double* init_qd, double* q, double* qd, float* cond_pos,
                         float* cond_vel, int32_t* traj_steps, int32_t* plan_steps, uint8_t* done, int B, int D,
                         void* stream) {
    hipLaunchKernelGGL(k_episode_reset, dim3((unsigned)(((long)B * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       init_q, init_qd, q, qd, cond_pos, cond_vel, traj_steps, plan_steps, done, B, D);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

// condition_on_desired (black_box_wrapper.py:199-201): the desired state at the last executed step of this plan
__global__ void __launch_bounds__(256) k_condition_gather(const float* __restrict__ pos, const float* __restrict__ vel,
                                                          const int32_t* __restrict__ seg_len,
                                                          float* __restrict__ cond_pos, float* __restrict__ cond_vel,
                                                          const int B, const int T, const int D) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)B * D) return;
    const int b = (int)(e / D), d = (int)(e - (long)b * D);
    int t = seg_len[b] - 1;
    t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    const size_t src = ((size_t)b * T + t) * D + d;
    cond_pos[e] = pos[src];
    cond_vel[e] = vel[src];
}

#ifndef MPK_DEVICE_ONLY
int launch_condition_gather(const float* pos, const float* vel, const int32_t* seg_len, float* cond_pos, float* cond_vel,
                            int B, int T, int D, void* stream) {
    hipLaunchKernelGGL(k_condition_gather, dim3((unsigned)(((long)B * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pos, vel, seg_len, cond_pos, cond_vel, B, T, D);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

// ------------------------------------------------------------------------------------------------------------
// validity reduction: one wave per episode
// ------------------------------------------------------------------------------------------------------------
struct ValidArgs {
    double lo[kMaxDofArgs], hi[kMaxDofArgs];
    double tb[2], db[2];
    int check_td, P, D, B, T;
};

__global__ void __launch_bounds__(256) k_validity(const ValidArgs v, const float* __restrict__ pos,
                                                  const float* __restrict__ params, uint8_t* __restrict__ valid,
                                                  double* __restrict__ penalty) {
    __shared__ double s_lo[kMaxDofArgs], s_hi[kMaxDofArgs];   // a lane-dependent index into the kernarg arrays would
    if (threadIdx.x < (unsigned)v.D) {                         // push the whole struct to scratch
        const double* lo = v.lo;
        const double* hi = v.hi;
        s_lo[threadIdx.x] = lo[threadIdx.x];
        s_hi[threadIdx.x] = hi[threadIdx.x];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= v.B) return;
    const int n = v.T * v.D;
    const float* p = pos + (size_t)b * n;
    bool ok = true;
    double over = 0.0, under = 0.0;
    for (int e = lane; e < n; e += 64) {
        const int d = e % v.D;
        const double x = (double)p[e];
        ok = ok && (x >= s_lo[d]) && (x <= s_hi[d]);
        over += fmax(x - s_hi[d], 0.0);
        under += fmax(s_lo[d] - x, 0.0);
    }
    double tpen = 0.0;
    if (v.check_td) {
        const double tau = (double)params[(size_t)b * v.P], delay = (double)params[(size_t)b * v.P + 1];
        if (lane == 0) ok = ok && tau >= v.tb[0] && tau <= v.tb[1] && delay >= v.db[0] && delay <= v.db[1];
        tpen = 3.0 * (fmax(0.0, tau - v.tb[1]) + fmax(0.0, v.tb[0] - tau)) +
               3.0 * (fmax(0.0, delay - v.db[1]) + fmax(0.0, v.db[0] - delay));
    }
    const bool all_ok = __all(ok);
    if (lane == 0) valid[b] = all_ok ? 1 : 0;
    if (penalty) {
        for (int m = 32; m >= 1; m >>= 1) {
            over += __shfl_xor(over, m);
            under += __shfl_xor(under, m);
        }
        // table_tennis_env.py:282-289: -(3*tau excess + 3*delay excess + mean(max(pos - high, 0)) + mean(max(low - pos, 0)))
        if (lane == 0) penalty[b] = -(tpen + over / (double)n + under / (double)n);
    }
}

#ifndef MPK_DEVICE_ONLY
int launch_validity(const float* pos, const float* params, int P, int D, const double* lo, const double* hi,
                    int check_td, const double* tb, const double* db, uint8_t* valid, double* penalty, int B, int T,
                    void* stream) {
    ValidArgs v{};
    for (int d = 0; d < D; ++d) { v.lo[d] = lo[d]; v.hi[d] = hi[d]; }
    if (check_td) { v.tb[0] = tb[0]; v.tb[1] = tb[1]; v.db[0] = db[0]; v.db[1] = db[1]; }
    v.check_td = check_td; v.P = P; v.D = D; v.B = B; v.T = T;
    hipLaunchKernelGGL(k_validity, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, v, pos, params, valid,
                       penalty);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

// ------------------------------------------------------------------------------------------------------------
// k_scaled_basis: traj_gen.show_scaled_basis (examples/mp_params_tuning.py:7) -- the basis functions times their
// parameter scale at arbitrary times, evaluated by the row functions the trajectory kernels use
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scaled_basis(const DevCfg c, const float* __restrict__ times, const int n,
                                                      float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = times[i];
    if (c.mp_type == MPK_MP_PRODMP) {
        const int N = c.n_pc, K = c.nb + 1;
        const double* PB = c.tab + 4 * (size_t)N;
        const double* S = PB + 2 * (size_t)N * K;
        const float s = scaled_time(t, c.delay, c.tau);
        const int idx = min(prodmp_index(s, c.scaled_dt), N - 1);
        for (int k = 0; k < K; ++k) out[(size_t)i * K + k] = (float)PB[(size_t)idx * K + k] * (float)S[k];
    } else {
        const double x = phase_f64(c, t, c.tau, c.delay, ExpLiteral());
        rbf_cols(c, x, (double)c.ws, out + (size_t)i * c.nb, 1);
    }
}

#ifndef MPK_DEVICE_ONLY
int launch_scaled_basis(const DevCfg& c, const float* times, int n, float* out, void* stream) {
    hipLaunchKernelGGL(k_scaled_basis, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, c, times, n, out);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

// ------------------------------------------------------------------------------------------------------------
// self-test of div_exact (the table-index arithmetic): every fp32 numerator bit pattern in [first, first + count) against
// the IEEE division, for one divisor
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_div_sweep(const float d, const uint32_t first, const uint64_t count,
                                                   unsigned long long* __restrict__ mismatches) {
    const ExactDiv x = make_exact_div(d);
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        const float z = __uint_as_float(first + (uint32_t)i);
        const float q0 = z / d, q1 = div_exact(z, x);
        // identical bits, or both NaN (numerators that are NaN / inf are outside any time grid but harmless)
        if (__float_as_uint(q0) != __float_as_uint(q1) && !(q0 != q0 && q1 != q1)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

#ifndef MPK_DEVICE_ONLY
int launch_div_sweep(float d, uint32_t first, uint64_t count, unsigned long long* mismatches, void* stream) {
    hipLaunchKernelGGL(k_div_sweep, dim3(4096), dim3(256), 0, (hipStream_t)stream, d, first, count, mismatches);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

// ------------------------------------------------------------------------------------------------------------
// k_reward_aggregate: reward_aggregation(rewards[:t + 1]) of black_box_wrapper.py:216 for the verbose = 2 path, in the order of
// k_episode_return (mpk_episode.hip) -- per step slot t mod 16 the sum over the row tiles in time order, then the sixteen slots
// left to right -- so that the two paths agree bit for bit.  16 lanes per episode.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_reward_aggregate(const double* __restrict__ rewards, const int32_t* __restrict__ seg_len,
                                                          const int agg, double* __restrict__ out, const int B, const int T) {
    const long e = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int tl = threadIdx.x & 15, lane = threadIdx.x & 63, base = lane & 48;
    const bool on = e < B;
    const int n = on ? min(seg_len[on ? e : 0], T) : 0;
    double acc = 0.0;
    if (on) {
        const double* r = rewards + (size_t)e * T;
        for (int t = tl; t < T; t += 16) {
            const double v = t < n ? r[t] : 0.0;
            if (agg == 2) acc = t == n - 1 ? v : acc;
            else acc = acc + v;
        }
    }
    double sum = __shfl(acc, base);
#pragma unroll
    for (int i = 1; i < 16; ++i) sum = sum + __shfl(acc, base + i);
    if (on && tl == 0) out[e] = agg == 1 ? (n > 0 ? sum / (double)n : 0.0) : sum;
}

#ifndef MPK_DEVICE_ONLY
int launch_reward_aggregate(const double* rewards, const int32_t* seg_len, int agg, double* out, int B, int T, void* stream) {
    hipLaunchKernelGGL(k_reward_aggregate, dim3((unsigned)((B + 15) / 16)), dim3(256), 0, (hipStream_t)stream, rewards, seg_len, agg,
                       out, B, T);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

}  // namespace mpk

#if defined(MPK_TRACE) && !defined(MPK_TRACE_UNIT)
#define MPK_TRACE_READER_HERE
#include "mpk_trace_reader.h"
#endif
