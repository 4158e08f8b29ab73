// k_traj_rows / k_traj_phase / k_traj_phase_dmp / k_dmp_prestep: per-episode phase (learned tau / delay, per-episode init_time)
#include "mpk_phase.h"

namespace mpk {

// ------------------------------------------------------------------------------------------------------------
// k_traj_rows: per-episode phase (learned tau/delay or per-episode init_time), one workgroup per episode
// ------------------------------------------------------------------------------------------------------------
struct RowArgs {
    DevCfg c;
    const float* params;
    const float* init_pos;
    const float* init_vel;
    const float* init_time;
    float init_time_shared;
    float* pos;
    float* vel;
    int32_t* flag;
    int B;
};

// One explicit Euler step of the reference's DMP (a = alpha (beta (g - y) - z) + f; z += ds a; y += ds z) in the per-episode-phase
// kernels.  Round 5 (MPK_DMP_PHASE_FMA 1): contracted -- five dependent fused operations instead of nine separately rounded ones.  The
// chain's latency is what bounds these kernels at a few thousand episodes (one wave runs a chunk's 200 dependent steps) and a fifth
// of their instructions at any size.  The fused form is the MORE accurate evaluation of the same step; against the separately rounded
// one (the shared-phase serial kernels, the float32 oracle) it moves results by ~1e-7 of the scale -- bit identity across kernel
// families is not the contract, 1e-5 against the reference is; all three per-episode DMP kernels take this
// function, so they still agree with each other bit for bit.  0: one rounding per operation, as rounds 1 - 4.
#ifndef MPK_DMP_PHASE_FMA
#define MPK_DMP_PHASE_FMA 1
#endif
__device__ __forceinline__ void dmp_phase_step(float& y, float& z, const float g, const float f, const float ds, const float alpha,
                                               const float beta) {
#if MPK_DMP_PHASE_FMA
    const float t1 = g - y;
    const float t3 = fmaf(beta, t1, -z);
    const float acc = fmaf(alpha, t3, f);
    z = fmaf(ds, acc, z);
    y = fmaf(ds, z, y);
#else
    const float t1 = g - y;
    const float t2 = beta * t1;
    const float t3 = t2 - z;
    const float t4 = alpha * t3;
    const float acc = t4 + f;
    z = z + ds * acc;
    y = y + ds * z;
#endif
}

template <int MP>
__global__ void __launch_bounds__(256) k_traj_rows(const RowArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevCfg& c = a.c;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int D = c.D, T = c.T, KT = c.KT, P = c.P;
    constexpr int NROW = MP == MPK_MP_PRODMP ? 2 : 1;
    float* sX = smem;                       // [D][KT]
    float* sH = sX + D * KT;                // [NROW][T][KT]
    float* sP = sH + NROW * T * KT;         // [T*D]   pos (promp) / force (dmp)
    float* sV = sP + T * D;                 // [T*D]   dmp only
    float* sT = sV + (MP == MPK_MP_DMP ? T * D : 0);  // [T] times (promp) / ds (dmp)

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float* prm = a.params + (size_t)b * P;
        float tau = c.tau, delay = c.delay;
        int o = 0;
        // np.clip(action, low, high): only tau / delay carry finite bounds (black_box_wrapper.py:104-105)
        if (c.learn_tau) { tau = fminf(fmaxf(prm[o], c.tau_lo), c.tau_hi); ++o; }
        if (c.learn_delay) { delay = fminf(fmaxf(prm[o], c.delay_lo), c.delay_hi); ++o; }
        const float it = a.init_time ? a.init_time[b] : a.init_time_shared;
        __syncthreads();  // previous episode's LDS fully consumed
        for (int e = tid; e < D * KT; e += nt) {
            const int dd = e / KT, k = e - dd * KT;
            float v = 0.0f;
            // RAW parameters / boundary conditions: every scale lives in the basis rows (see prodmp_col)
            if (MP == MPK_MP_PRODMP) {
                const int nb = c.nb;
                if (k < nb) {
                    if (!c.disable_weights) v = prm[c.off + dd * c.Kloc + k];
                } else if (k == nb) {
                    if (!c.disable_goal) v = prm[c.off + dd * c.Kloc + (c.disable_weights ? 0 : nb)];
                } else if (k == nb + 1) {
                    v = a.init_pos[(size_t)b * D + dd];
                } else if (k == nb + 2) {
                    v = a.init_vel[(size_t)b * D + dd];
                } else {
                    v = 1.0f;                      // goal-offset column (MPK_GOAL_OFFSET_ADD)
                }
            } else if (MP == MPK_MP_PROMP) {
                if (k < c.nb) v = prm[c.off + dd * c.Kloc + k];
                else v = a.init_pos[(size_t)b * D + dd];
            } else {
                v = prm[c.off + dd * c.Kloc + k];
            }
            sX[e] = v;
        }
        // basis rows for this episode's phase
        if (MP == MPK_MP_PRODMP) {
            const float sb = scaled_time(it, delay, tau);
            const int idxb = min(prodmp_index(sb, c.scaled_dt), c.n_pc - 1);
            ProdmpBC bc;
            prodmp_bc(c, idxb, bc);
            for (int t = tid; t < T; t += nt) {
                const float time = c.base_times[t] + it;
                const float s = scaled_time(time, delay, tau);
                if (s > (float)c.len_factor) atomicOr(a.flag, 1);
                const int idx = min(prodmp_index(s, c.scaled_dt), c.n_pc - 1);
                double xi[4];
                prodmp_xi(c, bc, idx, xi);
                for (int k = 0; k < KT; ++k) {
                    float h, hv;
                    prodmp_col(c, bc, idx, xi, k, (double)tau, div_pos(1.0, (double)tau), &h, &hv);
                    sH[t * KT + k] = h;
                    sH[(T + t) * KT + k] = hv;
                }
            }
        } else {
            for (int t = tid; t < T; t += nt) {
                const float time = c.base_times[t] + it;
                const double x = phase_f64(c, time, tau, delay, ExpLiteral());
                rbf_cols(c, x, MP == MPK_MP_PROMP ? (double)c.ws : x * (double)c.ws, sH + t * KT, 1);
                if (MP == MPK_MP_PROMP) {
                    if (KT > c.nb) sH[t * KT + c.nb] = 1.0f;
                    sT[t] = time;
                } else if (t < T - 1) {
                    sT[t] = scaled_time(c.base_times[t + 1] + it, delay, tau) - scaled_time(time, delay, tau);
                }
            }
        }
        __syncthreads();
        // contraction: fp32 fmaf chain in ascending k (the order of the MFMA accumulation)
        for (int e = tid; e < T * D; e += nt) {
            const int t = e / D, dd = e - t * D;
            const float* x = sX + dd * KT;
            float accp = 0.0f, accv = 0.0f;
            for (int k = 0; k < KT; ++k) {
                accp = fmaf(sH[t * KT + k], x[k], accp);
                if (MP == MPK_MP_PRODMP) accv = fmaf(sH[(T + t) * KT + k], x[k], accv);
            }
            if (MP == MPK_MP_PRODMP) {
                a.pos[(size_t)b * T * D + e] = accp;
                a.vel[(size_t)b * T * D + e] = accv;
            } else {
                sP[e] = accp;
                if (MP == MPK_MP_PROMP) a.pos[(size_t)b * T * D + e] = accp;
            }
        }
        if (MP == MPK_MP_PROMP) {
            __syncthreads();
            for (int e = tid; e < T * D; e += nt) {
                const int t = e / D, dd = e - t * D;
                const int th = t < T - 1 ? t + 1 : T - 1, tl = t < T - 1 ? t : T - 2;
                a.vel[(size_t)b * T * D + e] = (sP[th * D + dd] - sP[tl * D + dd]) * (1.0f / (sT[th] - sT[tl]));
            }
        } else if (MP == MPK_MP_DMP) {
            __syncthreads();
            if (tid < D) {
                const int dd = tid;
                float y = a.init_pos[(size_t)b * D + dd];
                float z = a.init_vel[(size_t)b * D + dd] * tau;
                const float g = prm[c.off + dd * c.Kloc + c.nb] * c.gs;
                const TauDiv td = make_tau_div(tau);
                for (int t = 0; t < T; ++t) {
                    const float f = sP[t * D + dd];
                    sP[t * D + dd] = y;
                    sV[t * D + dd] = div_tau(z, td);
                    if (t < T - 1) {
                        const float ds = sT[t];
                        dmp_phase_step(y, z, g, f, ds, c.dmp_alpha, c.dmp_beta);
                    }
                }
            }
            __syncthreads();
            for (int e = tid; e < T * D; e += nt) {
                a.pos[(size_t)b * T * D + e] = sP[e];
                a.vel[(size_t)b * T * D + e] = sV[e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_traj_phase: per-episode phase (learned tau / delay, per-episode init_time), one WAVE per episode, no workgroup
// barriers in the episode loop (D * KS <= 256 with KS = 8 or 16 contraction columns).
//   promp / prodmp -- lane <-> time step, 64 steps a round:
//     1. the lane gets ITS basis row in registers.  promp: fp64 phase + RBF evaluation with the device functions of
//        k_build_shared (a per-episode phase equal to the shared one gives identical bits).  prodmp: the reference's
//        own form  pos = c1*y1 + c2*y2 + Psi.wg   -- the row is a plain gather of [Psi | y1 y2] at the
//        lane's table index from an fp32 row table (one 64-byte line), the boundary conditions enter through
//        (c1, c2), solved per (episode, DoF) in float64; folding them into the rows (what the shared-phase kernels
//        do, because there it is free) would cost a float64 update per (episode, step, column)
//     2. for every DoF the raw parameter column X[d][:] is broadcast from LDS and the fmaf chain runs in ascending k
//        (the MFMA accumulation order); promp takes its forward difference from the next lane (lane 63 of a
//        non-final round only feeds lane 62)
//     3. the round's [64][D] block of pos / vel -- one contiguous run in HBM -- is staged in the wave's LDS slice at
//        the run's 16-byte phase and leaves as float4 stores
//     the next episode's header and parameter columns are fetched while the rows are built (before this episode's
//     stores enter the in-order memory queue)
//   dmp -- rows to LDS, forcing by lane <-> element, the Euler recurrence on D lanes, coalesced copy-out
// ------------------------------------------------------------------------------------------------------------
struct PhaseArgs {
    DevCfg c;
    const float* params;
    const float* init_pos;
    const float* init_vel;
    const float* init_time;
    float init_time_shared;
    float* pos;
    float* vel;
    int32_t* flag;
    int B, wave_floats, t_pad, x_pad, o_pad, c_pad, tab_pad, chunk, img_pad;
    int wt;     // write-through stores while the outputs are cache resident
    int vec_ok; // dmp: outputs 16-byte aligned and T * D a multiple of 4 (float4 stores)
    int h_pad;  // dmp: floats of the interpolation table of the forcing rows behind the centres (0: rows evaluated exactly)
};

// ---- per-episode-phase DMP: the forcing rows by INTERPOLATION (round 5) -------------------------------------------------------
// A row x phi_k(x) weights_scale depends on ONE variable, the left-bounded scaled time s = max((t - delay) / tau, 0) -- whatever an
// episode's tau, delay and init_time are.  Evaluated exactly it costs one float64 exponential for the phase and one per radial basis
// function per (episode, step): six for cfg3, ~90 float64 operations, two thirds of these kernels' arithmetic (DESIGN section 9:
// "issue bound on six float64 exponentials").  Here every workgroup tabulates the rows once, at kFastN + 3 equally spaced nodes of
// s in [-h, kFastS + h] with the builders' own float64 functions (rbf_row: the table IS the exact rows at its nodes), and an item takes
// the four nodes around its s through the cubic Lagrange polynomial: ~45 fp32 operations and eight 16-byte LDS reads.  Error
// 0.0234 h^4 |d4F/ds4| with h = 1 / 256: the rows of <= 5 basis functions (the only shape the table is built for: KS == 8, the
// reference's DMP configurations) have a fourth derivative <= ~6.4e3 (widths >= 0.14 in s), i.e. <= 6e-8 -- below the fp32 rounding
// of the row itself; items beyond s = kFastS (a tau far below the horizon) take the exact path.  Bit identity with the exact rows
// is given up (the 1e-5 contract is what holds across kernel families: DESIGN section 3); "phase_table" 0 restores the exact rows.
constexpr int kFastN = 448;            // (512 until the pipeline kernel needed 2 KB of a CU's LDS back: four of its workgroups per CU)
constexpr float kFastS = 2.0f;
constexpr int kFastRows = kFastN + 3;            // node j <-> s = (j - 1) kFastS / kFastN

template <int KS>
__device__ __forceinline__ void fast_rows_build(const DevCfg& c, const double* cen, float* tab, const int tid, const int nthreads) {
    for (int j = tid; j < kFastRows; j += nthreads) {
        const double s = (double)(j - 1) * (double)(kFastS / kFastN);
        // the phase's SMOOTH continuation on both sides of its range (node -1 below 0; linear phase: beyond its clip at 1 -- the items look
        // up min(s, 1), whose neighbours must not carry the clip's kink: with them clipped the rows within one node of s = 1 were off by
        // 2.5e-4 of their slope, 2.5e-6 of the velocity scale on cfg3 with a linear phase)
        double x;
        if (c.phase_type == MPK_PHASE_LINEAR) x = s;
        else x = s >= 0.0 ? exp_nonpos(-(double)c.alpha_phase * s) : div_pos(1.0, exp_nonpos((double)c.alpha_phase * s));
        float h[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) h[k] = 0.0f;
        rbf_row<KS>(c, cen, cen + c.n_total, x, x * (double)c.ws, h, ExpLiteral());
#pragma unroll
        for (int k = 0; k < KS; ++k) tab[j * KS + k] = h[k];
    }
}

// The table depends on the handle's configuration only (centres, bandwidths, phase constants): built ONCE, at mpk_create, into
// device memory (DevCfg::rows32 of a DMP handle, row stride 8) and copied into a workgroup's LDS -- 16.5 KB from L2 -- where every
// workgroup used to evaluate the nodes (515 then, kFastRows = 451 now) itself: as much float64 arithmetic as the 800 items of the one chunk a workgroup of
// k_traj_phase_dmp_wg owns (which therefore ran on the exact rows), 2 % of a launch of 65 536 episodes.
__global__ void __launch_bounds__(256) k_fast_rows_table(const DevCfg c, float* __restrict__ out) {
    fast_rows_build<8>(c, c.tab, out, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}
// the scaled time an item looks up: a linear phase is clipped at 1 (the table holds the smooth continuation beyond it, see fast_rows_build)
__device__ __forceinline__ float fast_rows_arg(const DevCfg& c, const float s) { return c.phase_type == MPK_PHASE_LINEAR ? fminf(s, 1.0f) : s; }
__device__ __forceinline__ void fast_rows_stage(const float* __restrict__ src, float* dst, const int n, const int tid, const int nthreads) {
    // (four loads in flight per thread before the first LDS write waits for one: a loop of load -> write is a memory round trip per pass)
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(__builtin_assume_aligned(dst, 16));
    const int n4 = n >> 2;
    for (int i0 = tid; i0 < n4; i0 += 4 * nthreads) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u * nthreads < n4 ? s4[i0 + u * nthreads] : float4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * nthreads < n4) d4[i0 + u * nthreads] = v[u];
    }
}

// the row at scaled time s in [0, kFastS): cubic Lagrange interpolation over the nodes i - 1 .. i + 2, i = floor(s / h)
template <int KS>
__device__ __forceinline__ void fast_rows_eval(const float* tab, const float s, float (&h)[KS]) {
    const float u = s * (float)(kFastN / kFastS);      // (224: the item's place between two nodes moves by 1e-7 of their distance)
    const int i = (int)u;
    const float t = u - (float)i, tm1 = t - 1.0f, tp1 = t + 1.0f, tm2 = t - 2.0f;
    const float w0 = (t * tm1) * tm2 * (-1.0f / 6.0f), w1 = (tp1 * tm1) * tm2 * 0.5f;
    const float w2 = (tp1 * t) * tm2 * -0.5f, w3 = (tp1 * t) * tm1 * (1.0f / 6.0f);
    const float* r = tab + i * KS;                      // rows i .. i + 3 = nodes i - 1 .. i + 2
#pragma unroll
    for (int j = 0; j < KS / 4; ++j) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(r + 4 * j), a1 = *reinterpret_cast<const f32x4*>(r + KS + 4 * j);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(r + 2 * KS + 4 * j), a3 = *reinterpret_cast<const f32x4*>(r + 3 * KS + 4 * j);
#pragma unroll
        for (int q = 0; q < 4; ++q) h[4 * j + q] = fmaf(w3, a3[q], fmaf(w2, a2[q], fmaf(w1, a1[q], w0 * a0[q])));
    }
}

template <int MP>
__device__ __forceinline__ float phase_x_value(const DevCfg& c, const float* __restrict__ prm,
                                               const float* __restrict__ ip, const float* __restrict__ iv, int dd, int k,
                                               int ks) {
    // RAW parameters / boundary conditions: every scale lives in the basis rows (see prodmp_col)
    if (MP == MPK_MP_DMP) {
        // the chain sees the weights only; goal, y0, ydot0 travel in the last three (otherwise zero) columns
        if (k < c.nb) return prm[c.off + dd * c.Kloc + k];
        if (k == ks - 3) return prm[c.off + dd * c.Kloc + c.nb];
        if (k == ks - 2) return ip[dd];
        return k == ks - 1 ? iv[dd] : 0.0f;
    }
    if (k >= c.KT) return 0.0f;
    if (MP == MPK_MP_PRODMP) {
        const int nb = c.nb;
        if (k < nb) return c.disable_weights ? 0.0f : prm[c.off + dd * c.Kloc + k];
        if (k == nb) return c.disable_goal ? 0.0f : prm[c.off + dd * c.Kloc + (c.disable_weights ? 0 : nb)];
        return k == nb + 1 ? ip[dd] : iv[dd];
    }
    return k < c.nb ? prm[c.off + dd * c.Kloc + k] : ip[dd];
}

template <int KQ>
__device__ __forceinline__ float row_chain(const float* __restrict__ row, const float (&x)[KQ * 4]) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        const float4 h = *reinterpret_cast<const float4*>(row + 4 * j);
        acc = fmaf(h.x, x[4 * j + 0], acc);
        acc = fmaf(h.y, x[4 * j + 1], acc);
        acc = fmaf(h.z, x[4 * j + 2], acc);
        acc = fmaf(h.w, x[4 * j + 3], acc);
    }
    return acc;
}

// episode header + parameter columns, one episode ahead (shared by both per-episode-phase kernels)
template <int MP, int KS>
struct PhaseFetch {
    static constexpr int NX = 4;                        // D * KS <= 256 values, one per lane and round
    float tau_raw, delay_raw, it;
    float xv[NX];
    __device__ __forceinline__ void issue(const PhaseArgs& a, int bb, int lane) {
        const DevCfg& c = a.c;
        const float* prm = a.params + (size_t)bb * c.P;
        const float* ip = a.init_pos + (size_t)bb * c.D;
        const float* iv = a.init_vel + (size_t)bb * c.D;
        int o = 0;
        tau_raw = c.tau; delay_raw = c.delay;
        if (c.learn_tau) { tau_raw = prm[o]; ++o; }
        if (c.learn_delay) delay_raw = prm[o];
        it = a.init_time ? a.init_time[bb] : a.init_time_shared;
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            const int e = lane + 64 * r;
            const int dd = e / KS, k = e - dd * KS;
            xv[r] = e < c.D * KS ? phase_x_value<MP>(c, prm, ip, iv, dd, k, KS) : 0.0f;
        }
    }
    __device__ __forceinline__ void park(float* sx, int n, int lane) const {
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            const int e = lane + 64 * r;
            if (e < n) sx[e] = xv[r];
        }
    }
};

// n floats staged at s0 / s1 [sh ...] (sh = 16-byte phase of the destinations: pos and vel share it) -> o0 / o1 [0 .. n):
// float4 body, dword head / tail, both arrays in one pass (shared chunk arithmetic)
// WT: write-through stores, every one of them (cache-resident batches; see store16)
template <bool WT>
__device__ __forceinline__ void flush_span2(const float* __restrict__ s0, const float* __restrict__ s1, float* __restrict__ o0,
                                            float* __restrict__ o1, int n, int sh, int lane) {
    const int end = sh + n;
    const int q0 = (sh + 3) >> 2, q1 = end >> 2;
    for (int q = q0 + lane; q < q1; q += 64) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(s0 + 4 * q), b = *reinterpret_cast<const f32x4*>(s1 + 4 * q);
        store16<WT>(o0 - sh + 4 * q, a);
        store16<WT>(o1 - sh + 4 * q, b);
    }
    const int head_end = 4 * q0 < end ? 4 * q0 : end;
    if (lane < head_end - sh) { store4<WT>(o0 + lane, s0[sh + lane]); store4<WT>(o1 + lane, s1[sh + lane]); }
    const int tail = 4 * q1 > head_end ? 4 * q1 : head_end;
    if (lane < end - tail) { store4<WT>(o0 - sh + tail + lane, s0[tail + lane]); store4<WT>(o1 - sh + tail + lane, s1[tail + lane]); }
}


// TL (prodmp): the fp32 row table is staged in the workgroup's LDS (row stride 2*KS + 4 floats: 16-byte aligned rows
// spread over the banks) and the workgroup is up to 16 waves, so row and boundary gathers never enter the memory queue
// FL (prodmp): the rounds run over the FLATTENED (episode, step) items of a chunk -- 64 consecutive items a round, whatever
// episode they belong to (a chunk's outputs are one contiguous run of HBM) -- instead of over each episode's steps: T = 100
// fills 100 of 128 lanes per episode the other way.  Everything per episode (clipped tau / delay, init_time, 1 / tau, the
// boundary-condition factors) is then per LANE, read from the chunk image; same arithmetic per (episode, step), same bits.
// DC: the DoF count at compile time (0: c.D) -- the per-DoF contraction loop then is straight-line code: at run time it was 162
// instructions per pair of DoF, 85 of them scalar address arithmetic (round 5; 7 = BASELINE cfg2 / cfg4, the reference's Panda tasks)
template <int MP, int KQ, bool TL, bool FL = false, int DC = 0>
__global__ void __launch_bounds__(TL ? 1024 : 256) k_traj_phase(const PhaseArgs a) {
    static_assert(MP != MPK_MP_DMP, "dmp has its own kernel");
    static_assert(!TL || MP == MPK_MP_PRODMP, "only prodmp has a row table");
    static_assert(!FL || MP == MPK_MP_PRODMP, "flat rounds: prodmp (promp's difference crosses lanes)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    MPK_STAMP(0);                                       // (trace builds: kernel entry of the traced wave)
    const DevCfg& c = a.c;
    constexpr int KS = KQ * 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int D = DC > 0 ? DC : c.D, T = c.T, KT = c.KT;
    double* sCen = reinterpret_cast<double*>(smem);     // [c_pad / 2] RBF centres | bandwidths, shared by the workgroup
    float* sBT = smem + a.c_pad;                        // [t_pad] base times, shared by the workgroup
    float* sTab = sBT + a.t_pad;                        // TL: [n_pc][2*KS + 4] row table, shared by the workgroup
    float* sImg = sTab + a.tab_pad + (size_t)wave * a.wave_floats;  // [2][img_pad] inputs of this / the next chunk
    float* sO0 = sImg + 2 * a.img_pad;                  // [o_pad] pos staging: [sh + lane * D + d]
    float* sO1 = sO0 + a.o_pad;                         // [o_pad] vel staging
    float* sXf = sO1 + a.o_pad;                         // promp: [x_pad] this episode's columns (prodmp: in the input image)
    float* sWgs = smem;                                 // prodmp: [KS] column scales | the goal scale (in place of sCen)
    const int E = a.chunk, P = c.P;
    const int img_floats = a.img_pad;
    constexpr int NLP = 5;                              // E * P <= 320 parameter values per chunk
    const int nchunks = (a.B + E - 1) / E;
    const int cstride = (int)gridDim.x * wpb;
    int ch = (int)blockIdx.x * wpb + wave;
    float lp[NLP] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, lip = 0.0f, liv = 0.0f, lit = 0.0f;
    auto issue_chunk = [&](int cc) {
        const int b0 = cc * E, ne = min(E, a.B - b0);
        const float* prm = a.params + (size_t)b0 * P;
        int ep = E * P;
        asm volatile("" : "+s"(ep));
#pragma unroll
        for (int r = 0; r < NLP; ++r)
            if (r == 0 || 64 * r < ep) lp[r] = lane + 64 * r < ne * P ? prm[lane + 64 * r] : 0.0f;    // (uniform: one load for E * P <= 64)
        lip = lane < ne * D ? a.init_pos[(size_t)b0 * D + lane] : 0.0f;
        liv = lane < ne * D ? a.init_vel[(size_t)b0 * D + lane] : 0.0f;
        lit = a.init_time && lane < ne ? a.init_time[b0 + lane] : a.init_time_shared;
    };
    auto park_chunk = [&](float* img) {
        // (the chunk's extents through an opaque scalar: the lane masks below are then formed where they are used -- hoisted out of the
        // chunk loop each of them lived in a pair of spilled SGPRs: round 5)
        int ep = E * P, ed = E * D, ee = E;
        asm volatile("" : "+s"(ep), "+s"(ed), "+s"(ee));
#pragma unroll
        for (int r = 0; r < NLP; ++r)
            if ((r == 0 || 64 * r < ep) && lane + 64 * r < ep) img[lane + 64 * r] = lp[r];
        if (lane < ed) { img[ep + lane] = lip; img[ep + ed + lane] = liv; }
        if (lane < ee) img[ep + 2 * ed + lane] = lit;
    };
    // the first chunk's inputs are requested BEFORE the workgroup stages its tables: one memory round trip under the other (round 5:
    // at a few thousand episodes a wave has one chunk, and its first 40 % were these two waits in a row -- tools/dev/trace_phase.py)
    if (ch < nchunks) issue_chunk(ch);
    MPK_STAMP(30);
    if (MP != MPK_MP_PRODMP) {
        for (int t = threadIdx.x; t < T; t += blockDim.x) sBT[t] = c.base_times[t];
        for (int k = threadIdx.x; k < 2 * c.n_total + 3; k += blockDim.x) sCen[k] = c.tab[k];
    } else {
        // every load of the tables is in flight before the first LDS write waits for one: the base times, the scale vector and the
        // row table four float4 per thread at a time (round 5: three loops of load -> write were three memory round trips in a row,
        // 2 100 of the 16 500 cycles a wave lives at a few thousand episodes -- tools/dev/trace_phase.py)
        const int tid = (int)threadIdx.x, bd = (int)blockDim.x;
        const float bt0 = tid < T ? c.base_times[tid] : 0.0f;
        // sWgs[k], k < KS: the scale of column k -- 0 where the column has no parameter (disabled block, padding) --, sWgs[KS]: the
        // goal scale itself, also when the goal is disabled (relative goals)
        const double* S = c.tab + 4 * (size_t)c.n_pc + 2 * (size_t)c.n_pc * (c.nb + 1);
        const double sk = tid <= c.nb ? S[tid] : 0.0;
        const double sg = tid == KS ? S[c.nb] : 0.0;
        if (TL) {
            // (round 5: only the rows an episode of this launch can reach -- a.tab_pad / kRow of them, by the launcher's bound on the
            // scaled time; at a few thousand episodes the copy of the whole 72 KB table was 40 % of a wave's life)
            const float4* src = reinterpret_cast<const float4*>(c.rows32);
            float4* dst = reinterpret_cast<float4*>(__builtin_assume_aligned(sTab, 16));
            const int n4 = a.tab_pad >> 2;
            for (int i0 = tid; i0 < n4; i0 += 4 * bd) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = i0 + u * bd < n4 ? src[i0 + u * bd] : float4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * bd < n4) dst[i0 + u * bd] = v[u];
            }
        }
        if (tid < T) sBT[tid] = bt0;
        for (int t = tid + bd; t < T; t += bd) sBT[t] = c.base_times[t];
        if (tid < KS) {
            const bool off = tid < c.nb ? c.disable_weights != 0 : (tid == c.nb ? c.disable_goal != 0 : true);
            sWgs[tid] = off ? 0.0f : (float)sk;
        }
        if (tid == KS) sWgs[KS] = (float)sg;
    }
    MPK_STAMP(31);
    __syncthreads();
    MPK_STAMP(32);
    const float* const rows = TL ? sTab : c.rows32;
    constexpr int kRow = 2 * KS + 4;    // [pos half .. y1 (f64) | vel half .. y2 (f64) | dy1 dy2 (f64)]
    const int row_max = TL ? a.tab_pad / kRow - 1 : c.n_pc - 1;       // last row a gather may touch (TL: last staged row)
    const ExactDiv dsdt = make_exact_div(c.scaled_dt);

    // A wave owns chunks of E consecutive episodes.  A chunk's inputs -- E parameter rows, E boundary positions /
    // velocities, E init_times: each one contiguous run -- are fetched with coalesced loads one chunk ahead and
    // collected once per chunk (into the other half of the wave's input image), right after the rows of the chunk's
    // last episode are built: the memory queue is in order, so collecting a load also waits for every store issued
    // before it, and that wait is paid per chunk instead of per episode.
    if (ch < nchunks) park_chunk(sImg);
    MPK_STAMP(33);
    constexpr int kStep = MP == MPK_MP_PROMP ? 63 : 64;
    ExpRegs ec;
    if (MP == MPK_MP_PROMP) ec.load();
    int slot = 0;
    for (; ch < nchunks; ch += cstride, slot ^= 1) {
        const float* img = sImg + slot * img_floats;
        const int b0 = ch * E, ne = min(E, a.B - b0);
        const bool more = ch + cstride < nchunks;
        MPK_STAMP(1);                                   // trace builds (tools/dev/trace_phase.py): chunk start
        if (more) issue_chunk(ch + cstride);
        __builtin_amdgcn_wave_barrier();
        if (MP == MPK_MP_PRODMP) {
            // The columns of ALL episodes of the chunk at once, one lane per (episode, DoF): wg = scale * [w; g] in fp32 as
            // the reference forms it, and the two boundary residuals of
            //   pos = xi1 * (y_b - Psi_b.wg) + xi2 * (tau ydot_b - dPsi_b.wg) + Psi.wg
            // (the reference's xi1 y_b + xi2 v_b + (Psi - xi1 Psi_b - xi2 dPsi_b).wg, regrouped so that nothing large cancels
            // in fp32) in float64, ascending k.  Per episode -- D lanes working, the wave paying every instruction -- this
            // block was 27 % of the kernel's cycles (profiles/r02_per_episode_phase.md); per chunk it is a quarter of that.
            // The chunk's input image is rewritten IN PLACE: every lane reads what it needs first, then the image becomes
            // [E][D][KS] columns [wg_0 .. wg_{K-1}, 0.., r1, r2] | [E][tau, delay, init_time] (clipped) -- no LDS on top.
            float* const imw = sImg + slot * img_floats;
            const int le = (int)(((unsigned)lane * (65536u / (unsigned)D + 1u)) >> 16), ld = lane - le * D;    // lane / D
            const bool on = lane < ne * D;
            // (round 5: straight-line selects -- weights sit at local index k < nw, the goal behind them; the per-column "is there a
            // parameter" logic as nested conditions cost ~700 scalar instructions and 127 spilled SGPRs per wave)
            // (the asm keeps the column tests where they are used: hoisted out of the chunk loop, each became a 64-bit mask in a pair
            // of spilled SGPRs)
            int nw = c.disable_weights ? 0 : c.nb, nbk = c.nb;
            asm volatile("" : "+s"(nw), "+s"(nbk));
            float raw[KS - 2], rawg = 0.0f, taul = c.tau, delayl = c.delay, itl = 0.0f, yb = 0.0f, ydb = 0.0f;
#pragma unroll
            for (int k = 0; k < KS - 2; ++k) raw[k] = 0.0f;
            if (on) {
                const float* prl = img + le * P;
                if (c.learn_tau) taul = fminf(fmaxf(prl[0], c.tau_lo), c.tau_hi);
                if (c.learn_delay) delayl = fminf(fmaxf(prl[c.learn_tau ? 1 : 0], c.delay_lo), c.delay_hi);
                itl = img[E * P + 2 * E * D + le];
                yb = img[E * P + lane]; ydb = img[E * P + E * D + lane];
                const float* loc = prl + c.off + ld * c.Kloc;
                // (reads past an episode's local block stay inside the wave's image; the select drops them)
#pragma unroll
                for (int k = 0; k < KS - 2; ++k) {
                    const float v = loc[k];
                    raw[k] = k < nw ? v : 0.0f;
                }
                if (!c.disable_goal) rawg = loc[nw];
            }
            __builtin_amdgcn_wave_barrier();                 // every read of the image is issued before its first write
            MPK_STAMP(4);
            if (on) {
                const float sbl = fmaxf(div_exact(itl - delayl, make_exact_div(taul)), 0.0f);
                const float* rb = rows + (size_t)min((int)rintf(div_exact(sbl, dsdt)), row_max) * kRow;
                double pb = 0.0, vb = 0.0;
                MPK_STAMP(5);
                float* xf = imw + le * a.x_pad + ld * KS;
                // the goal column: scaled goal; relative goal: init_pos joins the scaled goal, or (MPK_RELGOAL_BEFORE_SCALE) the raw
                // parameter -- zero when the goal is disabled -- before the scale
                float wgg = c.disable_goal ? 0.0f : rawg * sWgs[KS];
                if (c.relative_goal) wgg = c.relgoal_before_scale ? (rawg + yb) * sWgs[KS] : wgg + yb;
                if (c.goal_off_on) wgg = wgg + c.goal_offset;
#pragma unroll
                for (int k = 0; k < KS - 2; ++k) {
                    // (columns behind the goal: wg = +0 and the table holds zeros -- the sums take +0 and stay what they are)
                    float wg = raw[k] * sWgs[k];
                    wg = k == nbk ? wgg : wg;
                    pb += (double)rb[2 * k] * (double)wg;
                    vb += (double)rb[2 * k + 1] * (double)wg;
                    xf[k] = wg;
                }
                MPK_STAMP(6);
                xf[KS - 2] = (float)((double)yb - pb);
                xf[KS - 1] = (float)((double)(taul * ydb) - vb);
                if (ld == 0 && !FL) {
                    float* sc3 = imw + E * a.x_pad + 3 * le;
                    sc3[0] = taul; sc3[1] = delayl; sc3[2] = itl;
                }
                if (ld == 0 && FL) {
                    // [E][tau, delay, init_time, 1 / tau] | [E][4] float64 boundary-condition factors (see the per-episode
                    // block of the other path: the same expressions)
                    float* sc4 = imw + E * a.x_pad + 4 * le;
                    sc4[0] = taul; sc4[1] = delayl; sc4[2] = itl; sc4[3] = 1.0f / taul;
                    const double* yb4 = reinterpret_cast<const double*>(rb + 2 * KS - 4);
                    const double y1b = yb4[0], y2b = yb4[1], dy1b = yb4[2], dy2b = yb4[3];
                    const double idet = div_pos(1.0, y1b * dy2b - y2b * dy1b);
                    double* bc4 = reinterpret_cast<double*>(imw + E * a.x_pad + 4 * E) + 4 * le;
                    bc4[0] = dy2b * idet; bc4[1] = dy1b * idet; bc4[2] = y1b * idet; bc4[3] = y2b * idet;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (FL) {
            const int n_items = ne * T;
            const float rT = 1.0f / (float)T;
            float* const out_pos = a.pos + (size_t)b0 * T * D;
            float* const out_vel = a.vel + (size_t)b0 * T * D;
            for (int i0 = 0; i0 < n_items; i0 += 64) {
                const int nout = min(64, n_items - i0);
                const int i = min(i0 + lane, n_items - 1);
                int e = (int)(((float)i + 0.5f) * rT);          // i / T (i < 8 T), then made exact
                if (e * T > i) --e;
                if ((e + 1) * T <= i) ++e;
                const int t = i - e * T;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(img + E * a.x_pad + 4 * e);
                const float delay = sc[1], it = sc[2], inv_tau = sc[3];
                const ExactDiv dtau{sc[0], inv_tau, (__float_as_uint(sc[0]) & 0x7fffffu) == 0x7fffffu};
                const double* bc4 = reinterpret_cast<const double*>(img + E * a.x_pad + 4 * E) + 4 * e;
                const double bca = bc4[0], bcb = bc4[1], bcc = bc4[2], bcd = bc4[3];
                float hq[2 * KS];
                const float time = sBT[t] + it;
                const float s = fmaxf(div_exact(time - delay, dtau), 0.0f);
                if (s > (float)c.len_factor) atomicOr(a.flag, 1);
                const int idx = min((int)rintf(div_exact(s, dsdt)), row_max);
                const float4* row = reinterpret_cast<const float4*>(rows + (size_t)idx * kRow);
#pragma unroll
                for (int j = 0; j < (2 * KS - 4) / 4; ++j) {
                    const float4 q4 = row[j];
                    hq[4 * j] = q4.x; hq[4 * j + 1] = q4.y; hq[4 * j + 2] = q4.z; hq[4 * j + 3] = q4.w;
                }
                const double* y4 = reinterpret_cast<const double*>(row + (2 * KS - 4) / 4);
                const double y1 = y4[0], y2 = y4[1], dy1 = y4[2], dy2 = y4[3];
                hq[2 * KS - 4] = (float)fma(bca, y1, -(bcb * y2));
                hq[2 * KS - 3] = (float)fma(bca, dy1, -(bcb * dy2));
                hq[2 * KS - 2] = (float)fma(bcc, y2, -(bcd * y1));
                hq[2 * KS - 1] = (float)fma(bcc, dy2, -(bcd * dy1));
                if (more && i0 + 64 >= n_items) park_chunk(sImg + (slot ^ 1) * img_floats);
                float* const gp = out_pos + (size_t)i0 * D;
                const int sh = (int)((reinterpret_cast<uintptr_t>(gp) >> 2) & 3);
                const float* const sXl = img + e * a.x_pad;     // the lane's episode: at most two distinct ones per round
                auto dof = [&](int d) {
                    float x[KS];
#pragma unroll
                    for (int j = 0; j < KQ; ++j) {
                        const float4 v = *reinterpret_cast<const float4*>(sXl + d * KS + 4 * j);
                        x[4 * j + 0] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
                    }
                    f32x2 pv = {0.0f, 0.0f};
#pragma unroll
                    for (int k = 0; k < KS; ++k)
                        pv = __builtin_elementwise_fma(f32x2{hq[2 * k], hq[2 * k + 1]}, f32x2{x[k], x[k]}, pv);
                    sO0[sh + lane * D + d] = pv[0];
                    sO1[sh + lane * D + d] = pv[1] * inv_tau;
                };
                if constexpr (DC > 0) {
                    dofs_unrolled<DC, KQ>(sXl, hq, inv_tau, sO0 + sh + lane * DC, sO1 + sh + lane * DC);
                } else {
                    constexpr int ND = KQ <= 2 ? 2 : 1;
                    int d = 0;
                    for (; d + ND <= D; d += ND) {
#pragma unroll
                        for (int q = 0; q < ND; ++q) dof(d + q);
                    }
                    for (; d < D; ++d) dof(d);
                }
                __builtin_amdgcn_wave_barrier();
                if (a.wt) flush_span2<true>(sO0, sO1, gp, out_vel + (size_t)i0 * D, nout * D, sh, lane);
                else flush_span2<false>(sO0, sO1, gp, out_vel + (size_t)i0 * D, nout * D, sh, lane);
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }
        for (int e = 0; e < ne; ++e) {
            const int b = b0 + e;
            MPK_STAMP(2 + 40 * e);
            const float* prm = img + e * P;
            const float* ipe = img + E * P + e * D;
            const float* ive = ipe + E * D;
            // np.clip(action, low, high): only tau / delay carry finite bounds (black_box_wrapper.py:104-105)
            float tau = c.tau, delay = c.delay, it;
            if (MP == MPK_MP_PRODMP) {                  // clipped per chunk above
                const float* sc3 = img + E * a.x_pad + 3 * e;
                tau = sc3[0]; delay = sc3[1]; it = sc3[2];
            } else {
                if (c.learn_tau) tau = fminf(fmaxf(prm[0], c.tau_lo), c.tau_hi);
                if (c.learn_delay) delay = fminf(fmaxf(prm[c.learn_tau ? 1 : 0], c.delay_lo), c.delay_hi);
                it = img[E * P + 2 * E * D + e];
            }
            float inv_tau = 0.0f;
            double bca = 0.0, bcb = 0.0, bcc = 0.0, bcd = 0.0;    // dy2_b, dy1_b, y1_b, y2_b over det (see prodmp_bc)
            // table index = round(max((t - delay) / tau, 0) / scaled_dt): both quotients correctly rounded (div_exact), the
            // reciprocals taken once per episode / kernel instead of two IEEE divisions per step
            const ExactDiv dtau = make_exact_div(tau);
            const PosDiv taud = make_pos_div((double)tau);       // promp: the float64 phase divides by tau at every step
            if (MP == MPK_MP_PRODMP) {
                // Boundary conditions (SURVEY A.5 / mp_pytorch ProDMP): the episode's columns were built per chunk above;
                // xi1..xi4 are per (episode, step): the step's lane forms them below in float64 from the table values and
                // the factors kept here.
                const float sb = fmaxf(div_exact(it - delay, dtau), 0.0f);
                const int idxb = min((int)rintf(div_exact(sb, dsdt)), row_max);
                inv_tau = dtau.r;
                const float* rb = rows + (size_t)idxb * kRow;
                {
                    // y1, y2, dy1, dy2 sit behind the (Psi_k, dPsi_k) pairs as float64 (see mpk_create)
                    const double* yb4 = reinterpret_cast<const double*>(rb + 2 * KS - 4);
                    const double y1b = yb4[0], y2b = yb4[1], dy1b = yb4[2], dy2b = yb4[3];
                    const double idet = div_pos(1.0, y1b * dy2b - y2b * dy1b);      // det = y1_b^2 > 0
                    bca = dy2b * idet; bcb = dy1b * idet; bcc = y1b * idet; bcd = y2b * idet;
                }
            } else {
                // raw parameter columns [w_0 .. w_{nb-1}, init_pos (zero-padded family), 0 ..] per DoF
                for (int i = lane; i < D * KS; i += 64) {
                    const int dd = i / KS, k = i - dd * KS;
                    float v = 0.0f;
                    if (k < c.nb) v = prm[c.off + dd * c.Kloc + k];
                    else if (k < KT) v = ipe[dd];
                    sXf[i] = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            float* const out_pos = a.pos + (size_t)b * T * D;
            float* const out_vel = a.vel + (size_t)b * T * D;
            const float* const sXe = MP == MPK_MP_PRODMP ? img + e * a.x_pad : sXf;
            MPK_STAMP(3 + 40 * e);                      // columns built
            for (int r0 = 0; r0 < T; r0 += kStep) {
                const bool final_round = T - r0 <= 64;
                const int nout = final_round ? T - r0 : kStep;
                const int t = r0 + lane < T ? r0 + lane : T - 1;
                // prodmp: hq = (Psi_k, dPsi_k) pairs, as the table row holds them -- the position and velocity chains then are
                // ONE packed fp32 FMA per k (v_pk_fma_f32) instead of two; promp: h = the lane's RBF row
                float h[KS], hq[MP == MPK_MP_PRODMP ? 2 * KS : 2], rdt = 0.0f;
                const float time = sBT[t] + it;
                if (MP == MPK_MP_PRODMP) {
                    const float s = fmaxf(div_exact(time - delay, dtau), 0.0f);
                    if (s > (float)c.len_factor) atomicOr(a.flag, 1);
                    const int idx = min((int)rintf(div_exact(s, dsdt)), row_max);
                    const float4* row = reinterpret_cast<const float4*>(rows + (size_t)idx * kRow);
#pragma unroll
                    for (int j = 0; j < (2 * KS - 4) / 4; ++j) {
                        const float4 q4 = row[j];
                        hq[4 * j] = q4.x; hq[4 * j + 1] = q4.y; hq[4 * j + 2] = q4.z; hq[4 * j + 3] = q4.w;
                    }
                    // y1, y2, dy1, dy2 as float64 behind the pairs: turn them into (xi1, xi3) and (xi2, xi4), the pairs of
                    // the two boundary-condition columns
                    const double* y4 = reinterpret_cast<const double*>(row + (2 * KS - 4) / 4);
                    const double y1 = y4[0], y2 = y4[1], dy1 = y4[2], dy2 = y4[3];
                    // (a product and a fused multiply-add each: float64 runs at half rate, and this is per step)
                    hq[2 * KS - 4] = (float)fma(bca, y1, -(bcb * y2));
                    hq[2 * KS - 3] = (float)fma(bca, dy1, -(bcb * dy2));
                    hq[2 * KS - 2] = (float)fma(bcc, y2, -(bcd * y1));
                    hq[2 * KS - 1] = (float)fma(bcc, dy2, -(bcd * dy1));
                } else {
#pragma unroll
                    for (int k = 0; k < KS; ++k) h[k] = 0.0f;
                    const double x = phase_f64(c, time, taud, delay, ec);
                    rbf_row<KS>(c, sCen, sCen + c.n_total, x, (double)c.ws, h, ec);
                    const int th = t < T - 1 ? t + 1 : T - 1, tl = t < T - 1 ? t : T - 2;
                    rdt = 1.0f / ((sBT[th] + it) - (sBT[tl] + it));
                }
                MPK_STAMP(10 + 40 * e + (r0 ? 10 : 0));   // rows gathered / evaluated
                if (more && e == ne - 1 && r0 == 0) park_chunk(sImg + (slot ^ 1) * img_floats);
                float* const gp = out_pos + (size_t)r0 * D;
                const int sh = (int)((reinterpret_cast<uintptr_t>(gp) >> 2) & 3);
                // one (step, DoF) contraction; `ND` DoF per loop iteration: the pair's loads, chains and staging writes
                // share their address arithmetic and loop control, and the two chains fill each other's issue gaps (trace:
                // the kernel is vector-issue-bound; 9 of the 17 instructions per DoF were not FMAs)
                auto dof = [&](int d) {
                    float x[KS];
#pragma unroll
                    for (int j = 0; j < KQ; ++j) {
                        const float4 v = *reinterpret_cast<const float4*>(sXe + d * KS + 4 * j);
                        x[4 * j + 0] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
                    }
                    float p = 0.0f, v = 0.0f;
                    if (MP == MPK_MP_PRODMP) {
                        f32x2 pv = {0.0f, 0.0f};        // (pos, vel) chains, ascending k: one v_pk_fma_f32 per k
#pragma unroll
                        for (int k = 0; k < KS; ++k)
                            pv = __builtin_elementwise_fma(f32x2{hq[2 * k], hq[2 * k + 1]}, f32x2{x[k], x[k]}, pv);
                        p = pv[0];
                        v = pv[1] * inv_tau;
                    } else {
#pragma unroll
                        for (int k = 0; k < KS; ++k) p = fmaf(h[k], x[k], p);
                        const float nx = lane_above(p);
                        v = (nx - p) * rdt;
                        const float pv = lane_below(v);         // last row repeats the difference before it
                        if (r0 + lane == T - 1) v = pv;
                    }
                    sO0[sh + lane * D + d] = p;        // every lane: the staging holds 64 rows, rows >= nout never leave
                    sO1[sh + lane * D + d] = v;
                };
                if constexpr (DC > 0 && MP == MPK_MP_PRODMP) {
                    dofs_unrolled<DC, KQ>(sXe, hq, inv_tau, sO0 + sh + lane * DC, sO1 + sh + lane * DC);
                } else if constexpr (DC > 0 && MP == MPK_MP_PROMP) {
                    dofs_unrolled_promp<DC, KQ>(sXe, h, rdt, r0 + lane == T - 1, sO0 + sh + lane * DC, sO1 + sh + lane * DC);
                } else {
                    constexpr int ND = KQ <= 2 ? 2 : 1;
                    int d = 0;
                    for (; d + ND <= D; d += ND) {
#pragma unroll
                        for (int i = 0; i < ND; ++i) dof(d + i);
                    }
                    for (; d < D; ++d) dof(d);
                }
                __builtin_amdgcn_wave_barrier();
                MPK_STAMP(12 + 40 * e + (r0 ? 10 : 0));   // contracted, staged
                if (a.wt) flush_span2<true>(sO0, sO1, gp, out_vel + (size_t)r0 * D, nout * D, sh, lane);
                else flush_span2<false>(sO0, sO1, gp, out_vel + (size_t)r0 * D, nout * D, sh, lane);
                __builtin_amdgcn_wave_barrier();
                MPK_STAMP(13 + 40 * e + (r0 ? 10 : 0));   // stored
                if (final_round) break;
            }
        }
    }
}

template <int KQ, int DC = 0>       // DC: the DoF count at compile time (0: c.D), as k_traj_phase_dmp_pipe
__global__ void __launch_bounds__(512) k_traj_phase_dmp(const PhaseArgs a) {      // (four or eight waves: the launcher, by the waves a CU's LDS then holds)
    // DMP with a per-episode phase.  The Euler recurrence is serial in t and needs one lane per (episode, DoF); run per
    // episode it keeps D of 64 lanes busy for T dependent steps -- 7 % of the HBM roofline for 7 DoF (round 1 / 2).  Here a
    // wave owns a CHUNK of E (four, see the launcher) consecutive episodes and walks the horizon in tiles of 16 steps:
    //   A  lane <-> (episode, step of the tile): phase, RBF row (float64, the builders' functions: same bits as every other
    //      DMP kernel), the D forcing values of the step as fmaf chains in ascending k, the step's ds;
    //   B  lane <-> (episode, DoF): 16 Euler steps, one rounding per operation, all E * D recurrences at once;
    //   C  the tile's [E][16 * D] (pos | vel) blocks leave as float4 stores.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevCfg& c = a.c;
    constexpr int KS = KQ * 4, TT = 16;
    constexpr int MP = MPK_MP_DMP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int D = DC > 0 ? DC : c.D, T = c.T, E = a.chunk, P = c.P;
    const int seg = TT * D;                             // floats of one episode's tile
    double* sCen = reinterpret_cast<double*>(smem);     // [c_pad / 2] RBF centres | bandwidths (| recurrence constants)
    float* sFast = smem + a.c_pad;                      // [h_pad] interpolation table of the forcing rows (fast_rows_build)
    float* sBT = sFast + a.h_pad;                       // [t_pad] base times, shared by the workgroup
    float* sX = sBT + a.t_pad + (size_t)wave * a.wave_floats;   // [E][D][KS] columns: weights .., goal, y0, ydot0
    float* sPh = sX + E * a.x_pad;                      // [E][8] tau, delay, init_time (clipped), -, 1 / tau refined (float64), -
    float* sDs = sPh + 8 * E;                           // [E][TT] ds of the tile's steps
    float* sP = sDs + E * TT;                           // [E][TT * D] forcing -> pos   (round 5: the rows stay in registers, no row buffer)
    float* sV = sP + a.o_pad;                           // [E][TT * D] vel
    // (first elements of the small tables requested before the row table's copy waits for its loads)
    const int tid_ = (int)threadIdx.x, bd_ = (int)blockDim.x;
    const float bt0 = tid_ < T ? c.base_times[tid_] : 0.0f;
    const double cen0 = tid_ < 2 * c.n_total + 3 ? c.tab[tid_] : 0.0;
    const bool fast = a.h_pad > 0;
    if (fast) fast_rows_stage(c.rows32, sFast, a.h_pad, tid_, bd_);
    if (tid_ < T) sBT[tid_] = bt0;
    for (int t = tid_ + bd_; t < T; t += bd_) sBT[t] = c.base_times[t];
    if (tid_ < 2 * c.n_total + 3) sCen[tid_] = cen0;
    for (int k = tid_ + bd_; k < 2 * c.n_total + 3; k += bd_) sCen[k] = c.tab[k];
    __syncthreads();
    const float inv_d = 1.0f / (float)D;
    const int le = (int)(((float)lane + 0.5f) * inv_d), ld = lane - le * D;        // lane <-> (episode, DoF)
    const float inv_seg4 = 4.0f / (float)seg, inv_seg = 1.0f / (float)seg;   // (idx + 0.5) * inv: exact floor for idx < 2^16
    const bool vec = a.vec_ok != 0;                     // float4 stores: 16-byte aligned outputs, T * D a multiple of 4
    const int nchunks = (a.B + E - 1) / E;
    const int cstride = (int)gridDim.x * wpb;
    for (int ch = (int)blockIdx.x * wpb + wave; ch < nchunks; ch += cstride) {
        const int b0 = ch * E, ne = min(E, a.B - b0);
        // ---- the chunk's inputs: columns of every (episode, DoF), phase values per episode
        for (int idx = lane; idx < ne * D * KS; idx += 64) {
            const int pi = idx / KS, k = idx - pi * KS;             // pi = e * D + dd
            const int e = (int)(((float)pi + 0.5f) * inv_d), dd = pi - e * D;
            const size_t bb = (size_t)(b0 + e);
            sX[idx] = phase_x_value<MP>(c, a.params + bb * P, a.init_pos + bb * D, a.init_vel + bb * D, dd, k, KS);
        }
        float tau = c.tau, delay = c.delay, it = a.init_time_shared;
        const bool on = lane < ne * D;
        if (on) {
            const float* prm = a.params + (size_t)(b0 + le) * P;
            // np.clip(action, low, high): only tau / delay carry finite bounds (black_box_wrapper.py:104-105)
            if (c.learn_tau) tau = fminf(fmaxf(prm[0], c.tau_lo), c.tau_hi);
            if (c.learn_delay) delay = fminf(fmaxf(prm[c.learn_tau ? 1 : 0], c.delay_lo), c.delay_hi);
            if (a.init_time) it = a.init_time[b0 + le];
            if (ld == 0) {
                sPh[8 * le] = tau; sPh[8 * le + 1] = delay; sPh[8 * le + 2] = it;
                *reinterpret_cast<double*>(sPh + 8 * le + 4) = make_pos_div((double)tau).y;
            }
        }
        __builtin_amdgcn_wave_barrier();
        float y = 0.0f, z = 0.0f, g = 0.0f;
        if (on) {
            const float* xc = sX + lane * KS;
            g = xc[KS - 3] * c.gs; y = xc[KS - 2]; z = xc[KS - 1] * tau;
        }
        const TauDiv td = make_tau_div(tau);
        for (int t0 = 0; t0 < T; t0 += TT) {
            const int rows = min(TT, T - t0);
            const int ti_ = t0 / TT;                   // (trace builds: tools/dev/trace_phase_dmp.py -- 10 + 5 tile: tile start, + 1 rows,
            if (ti_ < 8) MPK_STAMP(10 + 5 * ti_);      //  + 2 forcing, + 3 Euler steps, + 4 stored)
            // ---- A: rows and forcing of the tile
            for (int i0 = 0; i0 < ne * TT; i0 += 64) {
                const int idx = i0 + lane, e = idx >> 4, tl = idx & (TT - 1), t = t0 + tl;
                const bool live = idx < ne * TT && t < T;
                // (the item's row stays in registers: round 4 wrote it to LDS and read it back once per DoF)
                float h[KS];
#pragma unroll
                for (int k = 0; k < KS; ++k) h[k] = 0.0f;
                if (live) {
                    const float taue = sPh[8 * e], delaye = sPh[8 * e + 1], ite = sPh[8 * e + 2];
                    const float time = sBT[t] + ite;
                    const float s_item = scaled_time(time, delaye, taue);
                    if (KS == 8 && fast && fast_rows_arg(c, s_item) < kFastS) {
                        if constexpr (KS == 8) fast_rows_eval<KS>(sFast, fast_rows_arg(c, s_item), h);
                    } else {
                        const PosDiv taud{(double)taue, *reinterpret_cast<const double*>(sPh + 8 * e + 4)};
                        const double x = phase_f64(c, time, taud, delaye, ExpLiteral());
                        // every RBF once, in registers (rbf_cols evaluates them for the sum and again for the values; same bits)
                        rbf_row<KS>(c, sCen, sCen + c.n_total, x, x * (double)c.ws, h, ExpLiteral());
                    }
                    if (t < T - 1) sDs[idx] = scaled_time(sBT[t + 1] + ite, delaye, taue) - s_item;
                }
                if (ti_ < 8 && i0 == 0) MPK_STAMP(11 + 5 * ti_);
                if constexpr (DC > 0) {
                    if (live) {                             // the DC chains side by side, offsets as constants (same chain order per DoF: same bits)
                        float x[DC][KS];
#pragma unroll
                        for (int d = 0; d < DC; ++d)
#pragma unroll
                            for (int j = 0; j < KQ; ++j) {
                                const float4 v = *reinterpret_cast<const float4*>(sX + (e * DC + d) * KS + 4 * j);
                                x[d][4 * j + 0] = v.x; x[d][4 * j + 1] = v.y; x[d][4 * j + 2] = v.z; x[d][4 * j + 3] = v.w;
                            }
                        float acc[DC];
#pragma unroll
                        for (int d = 0; d < DC; ++d) acc[d] = 0.0f;
#pragma unroll
                        for (int k = 0; k < KS - 3; ++k)
#pragma unroll
                            for (int d = 0; d < DC; ++d) acc[d] = fmaf(h[k], x[d][k], acc[d]);
#pragma unroll
                        for (int d = 0; d < DC; ++d) sP[e * seg + tl * DC + d] = acc[d];
                    }
                } else
                if (live) {
                    for (int d = 0; d < D; ++d) {
                        float x[KS];
#pragma unroll
                        for (int j = 0; j < KQ; ++j) {
                            const float4 v = *reinterpret_cast<const float4*>(sX + (e * D + d) * KS + 4 * j);
                            x[4 * j + 0] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
                        }
                        // row_chain's fmaf chain in ascending k over the weight columns (the last three columns carry goal, y0, ydot0:
                        // not weights -- the chain multiplied them by 0, which leaves the accumulator's bits alone)
                        float acc = 0.0f;
#pragma unroll
                        for (int k = 0; k < KS - 3; ++k) acc = fmaf(h[k], x[k], acc);
                        sP[e * seg + tl * D + d] = acc;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (ti_ < 8) MPK_STAMP(12 + 5 * ti_);
            // ---- B: 16 Euler steps of every (episode, DoF) of the chunk (one rounding per operation)
            if (on) {
                float* pp = sP + le * seg + ld;
                float* pv = sV + le * seg + ld;
                const float* pds = sDs + le * TT;
                if (t0 + TT < T) {
                    // a tile every step of which advances the state (all but the horizon's last): the 16 forcing values and step sizes
                    // into registers first (4 + 16 LDS reads issued together), then the chain without a read, a compare or a branch in it
                    // (left as the loop below, every step read its forcing value behind the previous step's writes to the same image)
                    float fr[TT], dsr[TT];
#pragma unroll
                    for (int j = 0; j < TT / 4; ++j) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(pds + 4 * j);
                        dsr[4 * j] = v[0]; dsr[4 * j + 1] = v[1]; dsr[4 * j + 2] = v[2]; dsr[4 * j + 3] = v[3];
                    }
#pragma unroll
                    for (int tl = 0; tl < TT; ++tl) fr[tl] = pp[tl * D];
#pragma unroll
                    for (int tl = 0; tl < TT; ++tl) {
                        pp[tl * D] = y;
                        pv[tl * D] = div_tau(z, td);
                        dmp_phase_step(y, z, g, fr[tl], dsr[tl], c.dmp_alpha, c.dmp_beta);
                    }
                } else {
                for (int tl = 0; tl < rows; ++tl) {
                    const float f = pp[tl * D];
                    pp[tl * D] = y;
                    pv[tl * D] = div_tau(z, td);
                    if (t0 + tl < T - 1) {
                        const float ds = pds[tl];
                        dmp_phase_step(y, z, g, f, ds, c.dmp_alpha, c.dmp_beta);
                    }
                }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (ti_ < 8) MPK_STAMP(13 + 5 * ti_);
            // ---- C: the tile's blocks, one contiguous run of rows * D floats per episode and array
            const int n = rows * D;
            if (vec) {
                const int n4 = n >> 2, tail = n & 3;
                for (int i0 = 0; i0 < ne * (seg >> 2); i0 += 64) {
                    const int idx = i0 + lane;
                    const int e = (int)(((float)idx + 0.5f) * inv_seg4), q = idx - e * (seg >> 2);
                    if (e < ne && q < n4) {
                        const size_t go = ((size_t)(b0 + e) * T + t0) * D + 4 * q;
                        const f32x4 vp = *reinterpret_cast<const f32x4*>(sP + e * seg + 4 * q);
                        const f32x4 vv = *reinterpret_cast<const f32x4*>(sV + e * seg + 4 * q);
                        if (a.wt) { store16<true>(a.pos + go, vp); store16<true>(a.vel + go, vv); }
                        else { store16<false>(a.pos + go, vp); store16<false>(a.vel + go, vv); }
                    }
                }
                if (tail) {                             // the last tile of a horizon whose rows * D is no multiple of 4
                    for (int i0 = 0; i0 < ne * 4; i0 += 64) {
                        const int idx = i0 + lane, e = idx >> 2, r = idx & 3;
                        if (e < ne && r < tail) {
                            const size_t go = ((size_t)(b0 + e) * T + t0) * D + 4 * n4 + r;
                            if (a.wt) { store4<true>(a.pos + go, sP[e * seg + 4 * n4 + r]); store4<true>(a.vel + go, sV[e * seg + 4 * n4 + r]); }
                            else { store4<false>(a.pos + go, sP[e * seg + 4 * n4 + r]); store4<false>(a.vel + go, sV[e * seg + 4 * n4 + r]); }
                        }
                    }
                }
            } else {
                for (int i0 = 0; i0 < ne * seg; i0 += 64) {
                    const int idx = i0 + lane;
                    const int e = (int)(((float)idx + 0.5f) * inv_seg), w = idx - e * seg;
                    if (e < ne && w < n) {
                        const size_t go = ((size_t)(b0 + e) * T + t0) * D + w;
                        if (a.wt) { store4<true>(a.pos + go, sP[e * seg + w]); store4<true>(a.vel + go, sV[e * seg + w]); }
                        else { store4<false>(a.pos + go, sP[e * seg + w]); store4<false>(a.vel + go, sV[e * seg + w]); }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();            // the tile's LDS reads are issued before the next tile's writes
            if (ti_ < 8) MPK_STAMP(14 + 5 * ti_);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_traj_phase_dmp_wg (round 4, second session): the per-episode-phase DMP for launches that leave k_traj_phase_dmp one wave per
// SIMD or fewer (a few thousand episodes: 55 us whatever the batch, every wave walking 13 tiles of rows -> Euler -> stores on its
// own).  Here a WORKGROUP of four waves owns a chunk of E episodes and walks the horizon in blocks of four 16-step tiles:
//   A  wave w builds the rows and forcing values of tile w of the block (the 64 (episode, step) items of a tile are one round of a
//      wave, as in k_traj_phase_dmp: the same functions, the same bits) -- four tiles at once;
//   B  wave 0 runs the block's 64 Euler steps of every (episode, DoF) lane (the serial part: the only one that stays serial);
//   C  all four waves store the block's (pos, vel) runs of 64 x D floats per episode.
// ------------------------------------------------------------------------------------------------------------
#ifndef MPK_DMP_WG_UNROLL
#define MPK_DMP_WG_UNROLL 8        // the Euler loop unrolled: 1 / 4 / 8 -> 33.8 / 32.1 / 30.9 us at 4 096 episodes of cfg3 + learned tau (86 registers, no scratch)
#endif
template <int KQ, int NTB>
__global__ void __launch_bounds__(NTB == 5 ? 320 : 256, 4) k_traj_phase_dmp_wg(const PhaseArgs a) {   // (four workgroups per CU: 4 096 episodes of cfg3 in one round)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevCfg& c = a.c;
    // NTB = tiles per block: 4 with chunks of up to four episodes (wave w: tile w), 2 with chunks of up to eight (wave w: tile w % 2
    // of episodes 4 (w / 2) ..): the four waves always build four rounds of 64 (episode, step) items at once; 5 = FIVE waves and blocks
    // of 80 steps where that saves a block (a block costs its rows + two barriers whatever its length: T = 200 is 80 + 80 + 40 instead
    // of 64 + 64 + 64 + 8 -- 17.5 -> 16.1 us at 2 048 episodes of cfg3 + learned tau; up to two workgroups per CU: the launcher)
    constexpr int KS = KQ * 4, TT = 16, TB = TT * NTB;     // steps per block
    constexpr int MP = MPK_MP_DMP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int D = c.D, T = c.T, E = a.chunk, P = c.P;
    const int bseg = TB * D;                            // floats of one episode's block
    MPK_STAMP(0);                                       // (trace builds, tools/dev/trace_phase_dmp.py wg: kernel entry)
    double* sCen = reinterpret_cast<double*>(smem);     // [c_pad / 2] RBF centres | bandwidths
    float* sFast = smem + a.c_pad;                      // [h_pad] interpolation table of the forcing rows (fast_rows_build)
    float* sBT = sFast + a.h_pad;                       // [t_pad] base times
    float* sX = sBT + a.t_pad;                          // [E][D][KS] columns: weights .., goal, y0, ydot0
    float* sPh = sX + E * a.x_pad;                      // [E][8] tau, delay, init_time (clipped), -, 1 / tau refined (float64), -
    float* sDs = sPh + 8 * E;                           // [E][TB] ds of the block's steps
    // [E][TB * D] PAIRS (forcing -> pos, tau x vel): the Euler wave leaves a step's two results with ONE 8-byte LDS write (a write costs
    // the lone wave ~18 cycles whatever its width: 37 of a step's 98 cycles were the two 4-byte ones), the storing waves part them
    float* sPV = sDs + E * TB;
    const float inv_d = 1.0f / (float)D;
    const int le = (int)(((float)lane + 0.5f) * inv_d), ld = lane - le * D;        // lane <-> (episode, DoF)  (wave 0)
    const int nchunks = (a.B + E - 1) / E;
    // the FIRST chunk's inputs are requested before the tables are staged (one memory round trip under the other: behind the table copy
    // they were 1 700 of a workgroup's 40 000 cycles at 4 096 episodes, where a workgroup has one chunk)
    const int ch0 = (int)blockIdx.x;
    float xf0 = 0.0f, tau0 = c.tau, delay0 = c.delay, it0 = a.init_time_shared;
    if (ch0 < nchunks) {
        const int b0 = ch0 * E, ne = min(E, a.B - b0);
        if ((int)threadIdx.x < ne * D * KS) {
            const int idx = (int)threadIdx.x, pi = idx / KS, kk = idx - pi * KS;
            const int e = (int)(((float)pi + 0.5f) * inv_d), dd = pi - e * D;
            const size_t bb = (size_t)(b0 + e);
            xf0 = phase_x_value<MP>(c, a.params + bb * P, a.init_pos + bb * D, a.init_vel + bb * D, dd, kk, KS);
        }
        if (wave == 0 && lane < ne * D) {
            const float* prm = a.params + (size_t)(b0 + le) * P;
            if (c.learn_tau) tau0 = fminf(fmaxf(prm[0], c.tau_lo), c.tau_hi);
            if (c.learn_delay) delay0 = fminf(fmaxf(prm[c.learn_tau ? 1 : 0], c.delay_lo), c.delay_hi);
            if (a.init_time) it0 = a.init_time[b0 + le];
        }
    }
    // (first elements of the small tables requested before the row table's copy waits for its loads)
    const int tid_ = (int)threadIdx.x, bd_ = (int)blockDim.x;
    const float bt0 = tid_ < T ? c.base_times[tid_] : 0.0f;
    const double cen0 = tid_ < 2 * c.n_total + 3 ? c.tab[tid_] : 0.0;
    const bool fast = a.h_pad > 0;
    if (fast) fast_rows_stage(c.rows32, sFast, a.h_pad, tid_, bd_);
    if (tid_ < T) sBT[tid_] = bt0;
    for (int t = tid_ + bd_; t < T; t += bd_) sBT[t] = c.base_times[t];
    if (tid_ < 2 * c.n_total + 3) sCen[tid_] = cen0;
    for (int k = tid_ + bd_; k < 2 * c.n_total + 3; k += bd_) sCen[k] = c.tab[k];
    const bool vec = a.vec_ok != 0;
    for (int ch = ch0; ch < nchunks; ch += (int)gridDim.x) {
        const int b0 = ch * E, ne = min(E, a.B - b0);
        // (no barrier here: the previous chunk's last block ended with one, and the tables are only read behind the next one -- the
        // chunk's input loads travel together with the table copy: 1 800 of a workgroup's 42 000 cycles at 4 096 episodes)
        MPK_STAMP(1);
        int idx = (int)threadIdx.x;
        if (ch == ch0) {
            if (idx < ne * D * KS) sX[idx] = xf0;
            idx += (int)blockDim.x;
        }
        for (; idx < ne * D * KS; idx += blockDim.x) {
            const int pi = idx / KS, k = idx - pi * KS;             // pi = e * D + dd
            const int e = (int)(((float)pi + 0.5f) * inv_d), dd = pi - e * D;
            const size_t bb = (size_t)(b0 + e);
            sX[idx] = phase_x_value<MP>(c, a.params + bb * P, a.init_pos + bb * D, a.init_vel + bb * D, dd, k, KS);
        }
        float tau = c.tau, delay = c.delay, it = a.init_time_shared;
        // (the Euler wave is wave 0 of EVERY workgroup: the serial chains of a CU's resident workgroups then share one SIMD, where they
        // interleave at no cost to each other -- a chain issues one instruction per ~9 cycles -- and leave the other three SIMDs to the
        // row / store phases; rotating the Euler wave over the SIMDs put every chain behind three busy waves: 21.9 -> 23.5 us at 4 096)
        const bool on = wave == 0 && lane < ne * D;
        if (on) {
            if (ch == ch0) {
                tau = tau0; delay = delay0; it = it0;
            } else {
                const float* prm = a.params + (size_t)(b0 + le) * P;
                if (c.learn_tau) tau = fminf(fmaxf(prm[0], c.tau_lo), c.tau_hi);
                if (c.learn_delay) delay = fminf(fmaxf(prm[c.learn_tau ? 1 : 0], c.delay_lo), c.delay_hi);
                if (a.init_time) it = a.init_time[b0 + le];
            }
            if (ld == 0) {
                sPh[8 * le] = tau; sPh[8 * le + 1] = delay; sPh[8 * le + 2] = it;
                *reinterpret_cast<double*>(sPh + 8 * le + 4) = make_pos_div((double)tau).y;
            }
        }
        __syncthreads();
        MPK_STAMP(2);
        float y = 0.0f, z = 0.0f, g = 0.0f;
        if (on) {
            const float* xc = sX + lane * KS;
            g = xc[KS - 3] * c.gs; y = xc[KS - 2]; z = xc[KS - 1] * tau;
        }
        for (int t0 = 0; t0 < T; t0 += TB) {
            const int rows = min(TB, T - t0);
            [[maybe_unused]] const int bi_ = t0 / TB;       // (trace builds: stamps 10 + 4 block: rows + forcing built, + 1 Euler starts, + 2 done, + 3 stored)
            // ---- A: wave w: rows and forcing of tile w of the block
            {
                const int e = (wave / NTB) * 4 + (lane >> 4), tl = lane & (TT - 1), tb = (wave % NTB) * TT + tl, t = t0 + tb;
                const bool live = e < ne && t < T;
                float h[KS];                        // (the item's row stays in registers, as in k_traj_phase_dmp)
#pragma unroll
                for (int k = 0; k < KS; ++k) h[k] = 0.0f;
                if (live) {
                    const float taue = sPh[8 * e], delaye = sPh[8 * e + 1], ite = sPh[8 * e + 2];
                    const float time = sBT[t] + ite;
                    const float s_item = scaled_time(time, delaye, taue);
                    if (KS == 8 && fast && fast_rows_arg(c, s_item) < kFastS) {
                        if constexpr (KS == 8) fast_rows_eval<KS>(sFast, fast_rows_arg(c, s_item), h);
                    } else {
                        const PosDiv taud{(double)taue, *reinterpret_cast<const double*>(sPh + 8 * e + 4)};
                        const double x = phase_f64(c, time, taud, delaye, ExpLiteral());
                        rbf_row<KS>(c, sCen, sCen + c.n_total, x, x * (double)c.ws, h, ExpLiteral());
                    }
                    if (t < T - 1) sDs[e * TB + tb] = scaled_time(sBT[t + 1] + ite, delaye, taue) - s_item;
                }
                if (live) {
                    // four DoF side by side (one after the other: two LDS reads, KS - 3 dependent FMAs and an LDS write, D times in a row)
                    for (int d0 = 0; d0 < D; d0 += 4) {
                        float x[4][KS];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int d = min(d0 + i, D - 1);
#pragma unroll
                            for (int j = 0; j < KQ; ++j) {
                                const float4 v = *reinterpret_cast<const float4*>(sX + (e * D + d) * KS + 4 * j);
                                x[i][4 * j + 0] = v.x; x[i][4 * j + 1] = v.y; x[i][4 * j + 2] = v.z; x[i][4 * j + 3] = v.w;
                            }
                        }
                        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};    // (row_chain's chain over the weight columns: k_traj_phase_dmp)
#pragma unroll
                        for (int k = 0; k < KS - 3; ++k)
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i] = fmaf(h[k], x[i][k], acc[i]);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (d0 + i < D) sPV[2 * (e * bseg + tb * D + d0 + i)] = acc[i];
                    }
                }
            }
            if (bi_ < 8) MPK_STAMP(10 + 4 * bi_);
            __syncthreads();
            if (bi_ < 8) MPK_STAMP(11 + 4 * bi_);
            // ---- B: wave 0: the block's Euler steps of every (episode, DoF) of the chunk (one rounding per operation)
            if (on) {
                float* pq = sPV + 2 * (le * bseg + ld);          // the lane's (pos, tau x vel) pairs, 2 D floats apart per step
                const float* pds = sDs + le * TB;
                // round 5: 8-step pieces every step of which advances the state take their forcing values and step sizes into
                // registers first (the loop form reads each forcing value behind the previous step's write to the same image: an LDS
                // round trip inside every one of a chunk's 200 dependent steps -- most of this kernel's time at 4 096 episodes)
                constexpr int PC = 8;               // (16 at a time spill: four workgroups per CU leave 128 registers)
                int tl0 = 0;
                for (; tl0 + PC <= rows && t0 + tl0 + PC < T; tl0 += PC) {
                    float fr[PC], dsr[PC];
#pragma unroll
                    for (int j = 0; j < PC / 4; ++j) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(pds + tl0 + 4 * j);
                        dsr[4 * j] = v[0]; dsr[4 * j + 1] = v[1]; dsr[4 * j + 2] = v[2]; dsr[4 * j + 3] = v[3];
                    }
#pragma unroll
                    for (int i = 0; i < PC; ++i) fr[i] = pq[2 * (tl0 + i) * D];
#pragma unroll
                    // (what a step costs the lone Euler wave, round 5 trace at one workgroup per CU: 98 cycles -- 61 for the five dependent
                    // operations and the reads, 37 for the two LDS writes; a lone wave issues one instruction per 5.6 - 9 cycles)
                    for (int i = 0; i < PC; ++i) {
                        *reinterpret_cast<f32x2*>(pq + 2 * (tl0 + i) * D) = f32x2{y, z};      // (tau x velocity: the division by tau is the storing waves')
                        dmp_phase_step(y, z, g, fr[i], dsr[i], c.dmp_alpha, c.dmp_beta);
                    }
                }
#pragma unroll 1
                for (int tl = tl0; tl < rows; ++tl) {
                    const float f = pq[2 * tl * D];
                    *reinterpret_cast<f32x2*>(pq + 2 * tl * D) = f32x2{y, z};
                    if (t0 + tl < T - 1) {
                        const float ds = pds[tl];
                        dmp_phase_step(y, z, g, f, ds, c.dmp_alpha, c.dmp_beta);
                    }
                }
            }
            if (bi_ < 8) MPK_STAMP(12 + 4 * bi_);
            __syncthreads();
            // ---- C: the block's runs, rows * D contiguous floats per episode and array
            const int n = rows * D;
            if (vec) {
                const int n4 = n >> 2;
                for (int idx = threadIdx.x; idx < ne * n4; idx += blockDim.x) {
                    const int e = idx / n4, q4 = idx - e * n4;
                    const size_t go = ((size_t)(b0 + e) * T + t0) * D + 4 * q4;
                    const f32x4 pa = *reinterpret_cast<const f32x4*>(sPV + 2 * (e * bseg + 4 * q4));
                    const f32x4 pb = *reinterpret_cast<const f32x4*>(sPV + 2 * (e * bseg + 4 * q4) + 4);
                    const f32x4 vp = {pa[0], pa[2], pb[0], pb[2]};
                    f32x4 vv = {pa[1], pa[3], pb[1], pb[3]};
                    const TauDiv tde = make_tau_div(sPh[8 * e]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) vv[q] = div_tau(vv[q], tde);
                    if (a.wt) { store16<true>(a.pos + go, vp); store16<true>(a.vel + go, vv); }
                    else { store16<false>(a.pos + go, vp); store16<false>(a.vel + go, vv); }
                }
            } else {
                for (int idx = threadIdx.x; idx < ne * n; idx += blockDim.x) {
                    const int e = idx / n, w = idx - e * n;
                    const size_t go = ((size_t)(b0 + e) * T + t0) * D + w;
                    const f32x2 pr = *reinterpret_cast<const f32x2*>(sPV + 2 * (e * bseg + w));
                    const float vel = div_tau(pr[1], make_tau_div(sPh[8 * e]));
                    if (a.wt) { store4<true>(a.pos + go, pr[0]); store4<true>(a.vel + go, vel); }
                    else { store4<false>(a.pos + go, pr[0]); store4<false>(a.vel + go, vel); }
                }
            }
            if (bi_ < 8) MPK_STAMP(13 + 4 * bi_);
            __syncthreads();                            // the block's LDS reads are issued before the next block's writes
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_traj_phase_dmp_pipe (round 5): k_traj_phase_dmp_wg as a PIPELINE.  The workgroup kernel walks rows -> barrier -> Euler steps on
// wave 0 -> barrier -> stores, and its trace at 4 096 episodes says 38 000 cycles of which 17 000 are the 200 Euler steps on ONE lone
// wave (84 cycles a step: the floor of this decomposition) and the rest the rows and stores that wait for them and are waited for.
// Here wave 0 ONLY steps; waves 1 - 3 (other SIMDs: the Euler chain keeps SIMD 0 to itself and to the chains of the CU's other
// workgroups, which interleave for free) each own ONE 16-step tile of every 48-step block and, stage after stage, store their tile of
// block s - 2 and then build rows + forcing of their tile of block s into the SAME place of buffer s & 1, while wave 0 steps through
// block s - 1 in the other buffer: one barrier per stage, no hazard between waves (a helper only ever touches its own tile's slots).
//   stage s:   wave 0: Euler steps of block s - 1 (buffer (s - 1) & 1)   |   wave 1 + h: store tile h of block s - 2, build tile h of block s
// Same functions, same order of operations per item as the workgroup and the wave kernels: the same bits.  Chunks of up to four episodes
// (a tile is 4 episodes x 16 steps = the 64 items of one round of a helper wave), eight columns, T > 48.
// ------------------------------------------------------------------------------------------------------------
template <int KQ, int DC = 0>      // DC: the DoF count at compile time (0: c.D) -- the kernel is bound by its instruction count
__global__ void __launch_bounds__(256, 4) k_traj_phase_dmp_pipe(const PhaseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevCfg& c = a.c;
    constexpr int KS = KQ * 4, TT = 16, NH = 3, TB = TT * NH;     // 48 steps per block: a tile per helper wave
    constexpr int MP = MPK_MP_DMP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int D = DC > 0 ? DC : c.D, T = c.T, E = a.chunk, P = c.P;
    const int bseg = TB * D;                            // (pos, tau x vel) pairs of one episode's block
    MPK_STAMP(0);
    double* sCen = reinterpret_cast<double*>(smem);     // [c_pad / 2] RBF centres | bandwidths
    float* sFast = smem + a.c_pad;                      // [h_pad] interpolation table of the forcing rows
    float* sBT = sFast + a.h_pad;                       // [t_pad] base times
    float* sX = sBT + a.t_pad;                          // [E][D][KS] columns: weights .., goal, y0, ydot0
    float* sPh = sX + E * a.x_pad;                      // [E][8] tau, delay, init_time (clipped), -, 1 / tau refined (float64), -
    float* sDs = sPh + 8 * E;                           // [2][E][TB] ds of a block's steps
    float* sPV = sDs + 2 * E * TB;                      // [2][E][TB * D] pairs: forcing -> pos | tau x vel
    const float inv_d = 1.0f / (float)D;
    const int le = (int)(((float)lane + 0.5f) * inv_d), ld = lane - le * D;        // lane <-> (episode, DoF)  (wave 0)
    const int nchunks = (a.B + E - 1) / E;
    const int ch0 = (int)blockIdx.x;
    float xf0 = 0.0f, tau0 = c.tau, delay0 = c.delay, it0 = a.init_time_shared;
    if (ch0 < nchunks) {                                // the first chunk's inputs: requested before the tables are staged
        const int b0 = ch0 * E, ne = min(E, a.B - b0);
        if ((int)threadIdx.x < ne * D * KS) {
            const int idx = (int)threadIdx.x, pi = idx / KS, kk = idx - pi * KS;
            const int e = (int)(((float)pi + 0.5f) * inv_d), dd = pi - e * D;
            const size_t bb = (size_t)(b0 + e);
            xf0 = phase_x_value<MP>(c, a.params + bb * P, a.init_pos + bb * D, a.init_vel + bb * D, dd, kk, KS);
        }
        if (wave == 0 && lane < ne * D) {
            const float* prm = a.params + (size_t)(b0 + le) * P;
            if (c.learn_tau) tau0 = fminf(fmaxf(prm[0], c.tau_lo), c.tau_hi);
            if (c.learn_delay) delay0 = fminf(fmaxf(prm[c.learn_tau ? 1 : 0], c.delay_lo), c.delay_hi);
            if (a.init_time) it0 = a.init_time[b0 + le];
        }
    }
    const int tid_ = (int)threadIdx.x, bd_ = (int)blockDim.x;
    const float bt0 = tid_ < T ? c.base_times[tid_] : 0.0f;
    const double cen0 = tid_ < 2 * c.n_total + 3 ? c.tab[tid_] : 0.0;
    const bool fast = a.h_pad > 0;
    if (fast) fast_rows_stage(c.rows32, sFast, a.h_pad, tid_, bd_);
    if (tid_ < T) sBT[tid_] = bt0;
    for (int t = tid_ + bd_; t < T; t += bd_) sBT[t] = c.base_times[t];
    if (tid_ < 2 * c.n_total + 3) sCen[tid_] = cen0;
    for (int k = tid_ + bd_; k < 2 * c.n_total + 3; k += bd_) sCen[k] = c.tab[k];
    const bool vec = a.vec_ok != 0;
    const int NB = (T + TB - 1) / TB;
    for (int ch = ch0; ch < nchunks; ch += (int)gridDim.x) {
        const int b0 = ch * E, ne = min(E, a.B - b0);
        MPK_STAMP(1);
        int idx = (int)threadIdx.x;
        if (ch == ch0) {
            if (idx < ne * D * KS) sX[idx] = xf0;
            idx += (int)blockDim.x;
        }
        for (; idx < ne * D * KS; idx += blockDim.x) {
            const int pi = idx / KS, k = idx - pi * KS;             // pi = e * D + dd
            const int e = (int)(((float)pi + 0.5f) * inv_d), dd = pi - e * D;
            const size_t bb = (size_t)(b0 + e);
            sX[idx] = phase_x_value<MP>(c, a.params + bb * P, a.init_pos + bb * D, a.init_vel + bb * D, dd, k, KS);
        }
        float tau = c.tau, delay = c.delay, it = a.init_time_shared;
        const bool on = wave == 0 && lane < ne * D;
        if (on) {
            if (ch == ch0) {
                tau = tau0; delay = delay0; it = it0;
            } else {
                const float* prm = a.params + (size_t)(b0 + le) * P;
                if (c.learn_tau) tau = fminf(fmaxf(prm[0], c.tau_lo), c.tau_hi);
                if (c.learn_delay) delay = fminf(fmaxf(prm[c.learn_tau ? 1 : 0], c.delay_lo), c.delay_hi);
                if (a.init_time) it = a.init_time[b0 + le];
            }
            if (ld == 0) {
                sPh[8 * le] = tau; sPh[8 * le + 1] = delay; sPh[8 * le + 2] = it;
                *reinterpret_cast<double*>(sPh + 8 * le + 4) = make_pos_div((double)tau).y;
            }
        }
        __syncthreads();
        MPK_STAMP(2);
        float y = 0.0f, z = 0.0f, g = 0.0f;
        if (on) {
            const float* xc = sX + lane * KS;
            g = xc[KS - 3] * c.gs; y = xc[KS - 2]; z = xc[KS - 1] * tau;
        }
        for (int s = 0; s <= NB + 1; ++s) {
            if (wave == 0) {
                // ---- B: the Euler steps of block s - 1 (the workgroup kernel's, step for step)
                if (on && s >= 1 && s <= NB) {
                    const int t0 = (s - 1) * TB, rows = min(TB, T - t0), bi = (s - 1) & 1;
                    float* pq = sPV + (size_t)bi * 2 * E * bseg + 2 * (le * bseg + ld);
                    const float* pds = sDs + bi * E * TB + le * TB;
                    constexpr int PC = 8;
                    int tl0 = 0;
                    for (; tl0 + PC <= rows && t0 + tl0 + PC < T; tl0 += PC) {
                        float fr[PC], dsr[PC];
#pragma unroll
                        for (int j = 0; j < PC / 4; ++j) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(pds + tl0 + 4 * j);
                            dsr[4 * j] = v[0]; dsr[4 * j + 1] = v[1]; dsr[4 * j + 2] = v[2]; dsr[4 * j + 3] = v[3];
                        }
#pragma unroll
                        for (int i = 0; i < PC; ++i) fr[i] = pq[2 * (tl0 + i) * D];
#pragma unroll
                        for (int i = 0; i < PC; ++i) {
                            *reinterpret_cast<f32x2*>(pq + 2 * (tl0 + i) * D) = f32x2{y, z};
                            dmp_phase_step(y, z, g, fr[i], dsr[i], c.dmp_alpha, c.dmp_beta);
                        }
                    }
#pragma unroll 1
                    for (int tl = tl0; tl < rows; ++tl) {
                        const float f = pq[2 * tl * D];
                        *reinterpret_cast<f32x2*>(pq + 2 * tl * D) = f32x2{y, z};
                        if (t0 + tl < T - 1) {
                            const float ds = pds[tl];
                            dmp_phase_step(y, z, g, f, ds, c.dmp_alpha, c.dmp_beta);
                        }
                    }
                }
                if (s < 8) MPK_STAMP(10 + 2 * s);
            } else {
                const int hw = wave - 1, bi = s & 1;
                float* const pvb = sPV + (size_t)bi * 2 * E * bseg;
                // ---- C: tile hw of block s - 2 leaves: rows * D contiguous floats per episode and array
                if (s >= 2) {
                    const int t0 = (s - 2) * TB + hw * TT, rows = min(TT, T - t0), n = rows * D;
                    if (rows > 0) {
                        const float* src = pvb + 2 * hw * TT * D;
                        if (vec && (n & 3) == 0) {
                            const int n4 = n >> 2;
                            const float inv_n4 = 1.0f / (float)n4;      // (i + 0.5) / n4: the exact floor for these few hundred indices
                            for (int i = lane; i < ne * n4; i += 64) {
                                const int e = (int)(((float)i + 0.5f) * inv_n4), q4 = i - e * n4;
                                const size_t go = ((size_t)(b0 + e) * T + t0) * D + 4 * q4;
                                const f32x4 pa = *reinterpret_cast<const f32x4*>(src + 2 * (e * bseg + 4 * q4));
                                const f32x4 pb = *reinterpret_cast<const f32x4*>(src + 2 * (e * bseg + 4 * q4) + 4);
                                const f32x4 vp = {pa[0], pa[2], pb[0], pb[2]};
                                f32x4 vv = {pa[1], pa[3], pb[1], pb[3]};
                                const TauDiv tde = make_tau_div(sPh[8 * e]);
#pragma unroll
                                for (int q = 0; q < 4; ++q) vv[q] = div_tau(vv[q], tde);
                                if (a.wt) { store16<true>(a.pos + go, vp); store16<true>(a.vel + go, vv); }
                                else { store16<false>(a.pos + go, vp); store16<false>(a.vel + go, vv); }
                            }
                        } else {
                            for (int i = lane; i < ne * n; i += 64) {
                                const int e = i / n, w = i - e * n;
                                const size_t go = ((size_t)(b0 + e) * T + t0) * D + w;
                                const f32x2 pr = *reinterpret_cast<const f32x2*>(src + 2 * (e * bseg + w));
                                const float vel = div_tau(pr[1], make_tau_div(sPh[8 * e]));
                                if (a.wt) { store4<true>(a.pos + go, pr[0]); store4<true>(a.vel + go, vel); }
                                else { store4<false>(a.pos + go, pr[0]); store4<false>(a.vel + go, vel); }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();    // (the tile's LDS reads are issued before its slots are written again)
                }
                // ---- A: rows and forcing of tile hw of block s
                if (s < NB) {
                    const int e = lane >> 4, tl = lane & (TT - 1), tb = hw * TT + tl, t = s * TB + tb;
                    const bool live = e < ne && t < T;
                    float h[KS];
#pragma unroll
                    for (int k = 0; k < KS; ++k) h[k] = 0.0f;
                    if (live) {
                        const float taue = sPh[8 * e], delaye = sPh[8 * e + 1], ite = sPh[8 * e + 2];
                        const float time = sBT[t] + ite;
                        const float s_item = scaled_time(time, delaye, taue);
                        if (KS == 8 && fast && fast_rows_arg(c, s_item) < kFastS) {
                            if constexpr (KS == 8) fast_rows_eval<KS>(sFast, fast_rows_arg(c, s_item), h);
                        } else {
                            const PosDiv taud{(double)taue, *reinterpret_cast<const double*>(sPh + 8 * e + 4)};
                            const double x = phase_f64(c, time, taud, delaye, ExpLiteral());
                            rbf_row<KS>(c, sCen, sCen + c.n_total, x, x * (double)c.ws, h, ExpLiteral());
                        }
                        if (t < T - 1) sDs[bi * E * TB + e * TB + tb] = scaled_time(sBT[t + 1] + ite, delaye, taue) - s_item;
                        if constexpr (DC > 0) {
                            float x[DC][KS];
#pragma unroll
                            for (int d = 0; d < DC; ++d)
#pragma unroll
                                for (int j = 0; j < KQ; ++j) {
                                    const float4 v = *reinterpret_cast<const float4*>(sX + (e * DC + d) * KS + 4 * j);
                                    x[d][4 * j + 0] = v.x; x[d][4 * j + 1] = v.y; x[d][4 * j + 2] = v.z; x[d][4 * j + 3] = v.w;
                                }
                            float acc[DC];
#pragma unroll
                            for (int d = 0; d < DC; ++d) acc[d] = 0.0f;
#pragma unroll
                            for (int k = 0; k < KS - 3; ++k)
#pragma unroll
                                for (int d = 0; d < DC; ++d) acc[d] = fmaf(h[k], x[d][k], acc[d]);
#pragma unroll
                            for (int d = 0; d < DC; ++d) pvb[2 * (e * bseg + tb * DC + d)] = acc[d];
                        } else
                        for (int d0 = 0; d0 < D; d0 += 4) {
                            float x[4][KS];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int d = min(d0 + i, D - 1);
#pragma unroll
                                for (int j = 0; j < KQ; ++j) {
                                    const float4 v = *reinterpret_cast<const float4*>(sX + (e * D + d) * KS + 4 * j);
                                    x[i][4 * j + 0] = v.x; x[i][4 * j + 1] = v.y; x[i][4 * j + 2] = v.z; x[i][4 * j + 3] = v.w;
                                }
                            }
                            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                            for (int k = 0; k < KS - 3; ++k)
#pragma unroll
                                for (int i = 0; i < 4; ++i) acc[i] = fmaf(h[k], x[i][k], acc[i]);
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (d0 + i < D) pvb[2 * (e * bseg + tb * D + d0 + i)] = acc[i];
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

#ifndef MPK_DEVICE_ONLY
int fast_rows_stride(const DevCfg& c) {          // 0: no interpolation table for this shape; else the kernels' KS (floats per node)
    // dmp: up to five basis functions in eight columns (where the error bound was derived).  ProMP was tried (round 5) and dropped: its
    // velocity is the forward difference of the positions, which multiplies the table's 1e-6 by 2 / dt -- 2e-5 of the velocity scale on
    // cfg5' for 5 - 10 % of the kernel's time (its rows are two exponentials by the product recurrence already)
    return c.mp_type == MPK_MP_DMP && c.KT + 3 <= 8 ? 8 : 0;
}
int fast_rows_floats(const DevCfg& c) { return kFastRows * fast_rows_stride(c); }
int launch_fast_rows_table(const DevCfg& c, float* out, void* stream) {
    hipLaunchKernelGGL(k_fast_rows_table, dim3((kFastRows + 255) / 256), dim3(256), 0, (hipStream_t)stream, c, out);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}

static int launch_traj_phase(const DevCfg& c, const PhaseArgs& base, int num_cu, void* stream,
                             const char** kernel_name, const Tuning& tune) {
    PhaseArgs pa = base;
    const bool dmp = c.mp_type == MPK_MP_DMP;
    bool flat = false, modelled = false;   // prodmp: chunk size chosen by the cost model (no balance rule on top)
    // dmp: + goal, y0, ydot0 columns; prodmp: weights, goal, y1 | y2 (a goal offset is added to the goal itself here)
    const int need = c.mp_type == MPK_MP_PRODMP ? c.nb + 3 : c.KT + (dmp ? 3 : 0);
    if (need > 16 || c.D > 64) return MPK_ENOTIMPL;
    const int KQ = need <= 4 && c.mp_type == MPK_MP_PROMP ? 1 : (need <= 8 ? 2 : 4), KS = KQ * 4;
    if (c.D * KS > 256) return MPK_ENOTIMPL;
    if (c.mp_type == MPK_MP_PRODMP && (!c.rows32 || c.rows32_stride != 2 * KS + 4)) return MPK_ENOTIMPL;
    pa.t_pad = (c.T + 3) / 4 * 4;
    pa.x_pad = c.D * KS;
    pa.o_pad = (64 * c.D + 4 + 3) / 4 * 4;
    if (dmp) {
        // a wave owns chunks of E consecutive episodes, one lane per (episode, DoF) in the Euler recurrence
        // measured at 7 DoF, T = 200 (us at B = 4096 / 65536): E = 1 94 / 1220, 2 67 / 633, 3 70 / 479, **4 62 / 406**, 6 96 / 454,
        // 9 131 / 503 -- four episodes make the 64 (episode, step) items of a tile exactly one round of the wave, and the
        // per-wave LDS (6.8 KB) still lets 20 waves share a CU; "phase_chunk" overrides (up to 64 / D, at most 16)
        const int e_max = 64 / c.D > 16 ? 16 : 64 / c.D;
        int E = e_max < 4 ? e_max : 4;
        if (tune.phase_chunk >= 1 && tune.phase_chunk <= e_max) E = tune.phase_chunk;
        pa.chunk = E;
        pa.o_pad = E * 16 * c.D;                                  // one (pos or vel) tile of the chunk
        pa.wave_floats = E * pa.x_pad + 8 * E + E * 16 + 2 * pa.o_pad;
        pa.vec_ok = ((reinterpret_cast<uintptr_t>(pa.pos) | reinterpret_cast<uintptr_t>(pa.vel)) & 15u) == 0 && (c.T * c.D) % 4 == 0 ? 1 : 0;
        // forcing rows by interpolation (fast_rows_build / _eval) where the table's error bound was derived: up to five basis
        // functions in eight columns; "phase_table" 0: the exact rows
        pa.h_pad = KS == 8 && tune.phase_table != 0 && c.rows32 && c.rows32_stride == 8 ? kFastRows * KS : 0;
    } else {
        // chunks of up to 4 consecutive episodes whose parameter rows fit the loader's 5 x 64 values and whose boundary
        // states fit one 64-lane load
        int E = 320 / c.P;
        E = E > 4 ? 4 : E;
        E = E > 64 / c.D ? 64 / c.D : E;
        if (E < 1) return MPK_ENOTIMPL;
        // prodmp: per-episode rounds with one episode per chunk, or flat rounds (k_traj_phase<.., FL>) over chunks of up to 8
        // episodes -- whichever has the shorter critical path per wave: passes over the resident waves x (rounds of a chunk +
        // ~2.5 rounds of per-chunk work: inputs, columns, boundary factors); a flat round costs ~15 % more (per-lane episode
        // constants).  Measured at cfg2 + learned tau (T = 100): B = 4096 11.7 us per-episode vs 15 - 24 flat; 16 384 31.9 vs
        // 23.4 - 24.8 (5 - 7 episodes per chunk); 65 536 94 vs 89; 262 144 equal (HBM) -- profiles/r03_per_episode_phase.md.
        // "phase_flat" / "phase_chunk" override.
        if (c.mp_type == MPK_MP_PRODMP) {
            int e_max = 320 / c.P;
            e_max = e_max > 8 ? 8 : e_max;
            e_max = e_max > 64 / c.D ? 64 / c.D : e_max;
            const size_t shared0 = (size_t)(pa.t_pad + KS + 4) * sizeof(float);
            const size_t tab_bytes = (size_t)c.n_pc * (2 * KS + 4) * sizeof(float);
            auto resident = [&](int e, bool fl) -> long {           // waves of the whole chip for this layout (as below)
                const int img_in = e * (c.P + 2 * c.D + 1), img_cols = e * (pa.x_pad + (fl ? 12 : 3));
                const size_t wb = (size_t)(2 * (((img_in > img_cols ? img_in : img_cols) + 3) / 4 * 4) + 2 * pa.o_pad) * sizeof(float);
                const bool tab = tune.phase_table != 0 && tab_bytes + 8 * wb <= kLdsPerCu - shared0 && (long)pa.B >= (long)num_cu * 8;
                int w = tab ? (int)((kLdsPerCu - shared0 - tab_bytes) / wb) : (int)((kLdsDefault - shared0) / wb);
                w = tab ? (w > 16 ? 16 : w) : (w > 4 ? 4 : (w < 1 ? 1 : w));
                int pc = (int)(kLdsPerCu / (wb * w + shared0 + (tab ? tab_bytes : 0)));
                pc = pc > 32 / w ? 32 / w : (pc < 1 ? 1 : pc);
                return (long)num_cu * pc * w;
            };
            auto cost = [&](int e, bool fl) -> double {
                const long chunks = ((long)pa.B + e - 1) / e, W = resident(e, fl);
                const double passes = (double)((chunks + W - 1) / W);
                const double rounds = fl ? 1.15 * (double)((e * c.T + 63) / 64) : (double)(e * ((c.T + 63) / 64));
                return passes * (rounds + 2.5);
            };
            if (tune.phase_flat == 0) {
                flat = false;
            } else if (tune.phase_flat == 1) {
                flat = true;
                double best = 1e300;
                for (int e = 1; e <= e_max; ++e)
                    if (cost(e, true) < best - 1e-9) { best = cost(e, true); E = e; }
                modelled = true;
            } else {
                double best = cost(1, false);
                E = 1; flat = false;
                for (int e = 2; e <= e_max; ++e)
                    if (cost(e, true) < best * 0.97) { best = cost(e, true); E = e; flat = true; }
                modelled = true;
            }
            if (flat && tune.phase_chunk >= 1 && tune.phase_chunk <= e_max) E = tune.phase_chunk;
            if (tune.phase_chunk >= 1) modelled = flat;
        }
        pa.chunk = E;
        // prodmp: the image is rewritten in place into [E][x_pad] columns + [E][3] clipped phase values (flat rounds: [E][4]
        // + [E][4] float64 boundary-condition factors)
        const int img_in = E * (c.P + 2 * c.D + 1), img_cols = c.mp_type == MPK_MP_PRODMP ? E * (pa.x_pad + (flat ? 12 : 3)) : 0;
        pa.img_pad = ((img_in > img_cols ? img_in : img_cols) + 3) / 4 * 4;
        pa.wave_floats = 2 * pa.img_pad + 2 * pa.o_pad + (c.mp_type == MPK_MP_PRODMP ? 0 : pa.x_pad);
    }
    pa.c_pad = c.mp_type == MPK_MP_PRODMP ? KS + 4 : (4 * c.n_total + 6 + 3) / 4 * 4;
    if (!dmp) pa.h_pad = 0;
    const size_t wave_bytes = (size_t)pa.wave_floats * sizeof(float);
    size_t shared_bytes = (size_t)(pa.t_pad + pa.c_pad + pa.h_pad) * sizeof(float);
    if (wave_bytes + shared_bytes > kLdsPerCu) return MPK_ENOTIMPL;
    int wpb = (int)((kLdsDefault - shared_bytes) / wave_bytes);
    wpb = wpb > 4 ? 4 : (wpb < 1 ? 1 : wpb);
    if (dmp && pa.h_pad > 0 && wpb == 4) {
        // the waves of a workgroup share one copy of the row table (16.5 KB): eight waves per workgroup where that puts more waves
        // on a CU -- cfg3': 45 KB x 3 = 12 waves per CU with four, 73 KB x 2 = 16 with eight, i.e. 16 384 instead of 12 288 episodes in
        // ONE round of resident waves (round 5: 16 384 episodes 87.6 -> 69.0 us, 32 768: 150.8 -> 137.9, 65 536: 279 -> 265)
        auto resident = [&](int w) {
            const size_t l = wave_bytes * w + shared_bytes;
            if (l > kLdsPerCu) return 0;
            const int pc = (int)(kLdsPerCu / l);
            return w * (pc > 32 / w ? 32 / w : pc);
        };
        // (only where four waves per workgroup need a second round: a single round runs faster with fewer waves per SIMD --
        // 12 288 episodes 59 us with twelve waves per CU, 69 with sixteen)
        const long chunks_ = ((long)pa.B + pa.chunk - 1) / pa.chunk;
        // (a tie goes to eight: half as many workgroups build the table, and half as many copies of it sit in the CU's LDS --
        // cfg3' at 65 536 episodes 236 us with four waves per workgroup, 217 with eight, sixteen waves per CU either way)
        if (resident(8) >= resident(4) && chunks_ > (long)num_cu * resident(4)) wpb = 8;
    }
    // prodmp: stage the row table in LDS when it leaves room for at least 8 waves ("phase_table" 0: gather from L2)
    bool lds_table = false;
    if (c.mp_type == MPK_MP_PRODMP) {
        const size_t full_bytes = (size_t)c.n_pc * (2 * KS + 4) * sizeof(float);
        const size_t room = kLdsPerCu - shared_bytes;
        lds_table = full_bytes + 8 * wave_bytes <= room && (long)pa.B >= (long)num_cu * 8;
        if (tune.phase_table == 0) lds_table = false;
        if (lds_table) {
            // rows an episode can reach: scaled time <= (last grid time + init_time - smallest delay) / smallest tau (the kernels clip tau
            // and delay to their bounds); per-episode init_times are device data: the whole table then
            int rows_needed = c.n_pc;
            if (!pa.init_time) {
                const float tau_lo = c.learn_tau ? c.tau_lo : c.tau, delay_lo = c.learn_delay ? c.delay_lo : c.delay;
                const double t_last = (double)c.t_last;
                if (t_last > 0.0 && tau_lo > 0.f) {
                    const double s_max = (t_last + (double)pa.init_time_shared - (double)delay_lo) / (double)tau_lo;
                    const double r = s_max / (double)c.scaled_dt + 4.0;
                    if (r < (double)c.n_pc) rows_needed = r < 4.0 ? 4 : (int)r;
                }
            }
            pa.tab_pad = rows_needed * (2 * KS + 4);
            const size_t tab_bytes = (size_t)pa.tab_pad * sizeof(float);
            shared_bytes += tab_bytes;
            wpb = (int)((kLdsPerCu - shared_bytes) / wave_bytes);
            wpb = wpb > 16 ? 16 : wpb;
            // Fewer waves per CU for large launches of the flat rounds: every wave writes its own chunk's run of HBM, and sixteen streams
            // per CU write slower than eight (round 5, alternating rounds on three boxes; workgroups of four waves = two workgroups per
            // CU by the table's LDS: 65 536 episodes 87.5 / 102.6 us with sixteen waves per workgroup, 74.7 / 90.8 with four; 262 144:
            // 368 -> 350; against workgroups of eight: 24 576 28.7 -> 27.2, 32 768 41.0 -> 35.8, 131 072 167 -> 158, 262 144 288 -> 281;
            // workgroups of two lose again: 302.  At 16 384 and below the sixteen are faster: 21.5 against 23.0).
            // "tiles_wpb" 1 .. 8 sets the cap itself (A/B runs).
            if (tune.tiles_wpb > 0) wpb = wpb > tune.tiles_wpb ? tune.tiles_wpb : wpb;
            else if (flat && wpb > 4 && pa.B >= 24576) wpb = 4;
        }
    }
    size_t lds = wave_bytes * wpb + shared_bytes;
    int per_cu = (int)(kLdsPerCu / lds);
    per_cu = per_cu > 32 / wpb ? 32 / wpb : per_cu;
    if (tune.phase_waves > 0 && per_cu * wpb > tune.phase_waves) per_cu = tune.phase_waves / wpb > 1 ? tune.phase_waves / wpb : 1;
    if (!dmp && !modelled) {
        // chunks cost balance (a wave's work is quantised in E episodes): only when every resident wave still gets >= 4
        const long resident = (long)num_cu * per_cu * wpb;
        int E = pa.chunk;
        while (E > 1 && (long)pa.B / E < 4 * resident) E >>= 1;
        if (tune.phase_chunk >= 1 && tune.phase_chunk <= pa.chunk) E = tune.phase_chunk;
        pa.chunk = E;
    }
    const long units = ((long)pa.B + pa.chunk - 1) / pa.chunk;
    // fewer chunks than one workgroup per CU would take: smaller workgroups, so that every CU gets its share (round 5: chunks of two
    // at 4 096 episodes were 128 workgroups of 16 waves on 256 CUs)
    if (lds_table && units < (long)num_cu * wpb) {
        const int w = (int)((units + num_cu - 1) / num_cu);
        wpb = w < 1 ? 1 : w;
        lds = wave_bytes * wpb + shared_bytes;
    }
    long blocks = (units + wpb - 1) / wpb;
    if (blocks > (long)num_cu * per_cu) blocks = (long)num_cu * per_cu;
    auto go = [&](auto kern) -> int {
        if (lds > kLdsDefault) {
            hipError_t e = allow_full_lds(kern);
            if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * wpb), lds, (hipStream_t)stream, pa);
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    };
    switch (c.mp_type) {
        case MPK_MP_PRODMP:
            // (seven DoF, <= 8 columns: the instantiations with the DoF loop unrolled; "pd_generic" 1: the run-time loop, for A/B runs and tests)
            if (lds_table) {
                *kernel_name = flat ? "k_traj_phase<prodmp,lds,flat>" : "k_traj_phase<prodmp,lds>";
                if (KQ == 2 && c.D == 7 && tune.pd_generic != 1)
                    return flat ? go(k_traj_phase<MPK_MP_PRODMP, 2, true, true, 7>) : go(k_traj_phase<MPK_MP_PRODMP, 2, true, false, 7>);
                if (flat) return KQ == 2 ? go(k_traj_phase<MPK_MP_PRODMP, 2, true, true>) : go(k_traj_phase<MPK_MP_PRODMP, 4, true, true>);
                return KQ == 2 ? go(k_traj_phase<MPK_MP_PRODMP, 2, true>) : go(k_traj_phase<MPK_MP_PRODMP, 4, true>);
            }
            *kernel_name = flat ? "k_traj_phase<prodmp,flat>" : "k_traj_phase<prodmp>";
            if (KQ == 2 && c.D == 7 && tune.pd_generic != 1)
                return flat ? go(k_traj_phase<MPK_MP_PRODMP, 2, false, true, 7>) : go(k_traj_phase<MPK_MP_PRODMP, 2, false, false, 7>);
            if (flat) return KQ == 2 ? go(k_traj_phase<MPK_MP_PRODMP, 2, false, true>) : go(k_traj_phase<MPK_MP_PRODMP, 4, false, true>);
            return KQ == 2 ? go(k_traj_phase<MPK_MP_PRODMP, 2, false>) : go(k_traj_phase<MPK_MP_PRODMP, 4, false>);
        case MPK_MP_PROMP:
            *kernel_name = "k_traj_phase<promp>";
            if (c.D == 7 && KQ <= 2 && tune.pd_generic != 1)      // (seven DoF: the DoF chains unrolled side by side, as prodmp's)
                return KQ == 1 ? go(k_traj_phase<MPK_MP_PROMP, 1, false, false, 7>) : go(k_traj_phase<MPK_MP_PROMP, 2, false, false, 7>);
            if (KQ == 1) return go(k_traj_phase<MPK_MP_PROMP, 1, false>);
            return KQ == 2 ? go(k_traj_phase<MPK_MP_PROMP, 2, false>) : go(k_traj_phase<MPK_MP_PROMP, 4, false>);
        default: {
            // few chunks (the wave-per-chunk kernel would run at one wave per SIMD or fewer, latency bound): a workgroup per chunk,
            // four tiles of rows at once (k_traj_phase_dmp_wg).  "phase_flat" 1 forces it, 0 forbids it (A/B runs, tests)
            // two geometries: chunks of (up to) four episodes in blocks of four tiles, or -- when that needs more than one round of
            // resident workgroups -- chunks of eight in blocks of two tiles (twice the episodes per round)
            // (the interpolation table here too since it is a copy of the handle's: cfg3' at 4 096 episodes 26.8 us on the exact rows,
            // 21.9 with the table; when every workgroup built it for its one or two chunks it cost more than it saved, 35.3 vs 31)
            const int wg_h = pa.h_pad;
            auto wg_bytes = [&](int e, int ntb) {
                return ((size_t)pa.c_pad + wg_h + pa.t_pad + (size_t)e * pa.x_pad + 8 * e + (size_t)e * 16 * ntb +
                        2 * (size_t)e * 16 * ntb * c.D) * sizeof(float);
            };
            auto wg_resident = [&](size_t bytes) {
                int r = (int)(kLdsPerCu / bytes);                        // LDS, and five (four) by its 86 - 108 registers
                return r > (KQ == 2 ? 5 : 4) ? (KQ == 2 ? 5 : 4) : r;
            };
            const bool user_chunk = tune.phase_chunk >= 1;
            int wgE = pa.chunk, wgNTB = 4;
            bool wg_ok = pa.chunk <= 4 && pa.chunk * c.D <= 64 && wg_bytes(pa.chunk, 4) <= kLdsPerCu;
            long chunks = ((long)pa.B + wgE - 1) / wgE;
            if (wg_ok && !user_chunk && pa.chunk == 4 && 8 * c.D <= 64 && chunks > (long)num_cu * wg_resident(wg_bytes(4, 4)) &&
                wg_bytes(8, 2) <= kLdsPerCu) {
                wgE = 8; wgNTB = 2;
                chunks = ((long)pa.B + 7) / 8;
            }
            // five waves, blocks of 80 steps: where that is a block fewer and a CU holds at most two workgroups (with four, the 20 waves did
            // not fit the CU's SIMDs -- 4 096 episodes 29.7 us against 20.1; "tiles_wpb" 4: the four-wave geometry, for A/B runs)
            if (wg_ok && wgNTB == 4 && KQ == 2 && (c.T + 79) / 80 < (c.T + 63) / 64 && wg_bytes(wgE, 5) * 4 <= kLdsPerCu &&
                chunks <= (long)num_cu * 2 && tune.tiles_wpb != 4)
                wgNTB = 5;
            const size_t wg_lds = wg_bytes(wgE, wgNTB);
            const int wg_res = wg_resident(wg_lds);
            // the pipeline form (k_traj_phase_dmp_pipe: wave 0 only steps, three helper waves build and store around it in blocks of 48
            // steps, two buffers): chunks of up to four episodes, eight columns, more than one block, and the launch in ONE round of its
            // resident workgroups ("pipe" 0: the plain workgroup kernel, for A/B runs and tests)
            const size_t pipe_lds = ((size_t)pa.c_pad + wg_h + pa.t_pad + (size_t)wgE * pa.x_pad + 8 * wgE + 2 * (size_t)wgE * 48 +
                                     4 * (size_t)wgE * 48 * c.D) * sizeof(float);
            int pipe_res = (int)(kLdsPerCu / pipe_lds);
            pipe_res = pipe_res > 4 ? 4 : pipe_res;
            const bool pipe_form = wg_ok && wgE <= 4 && wgNTB != 2 && KQ == 2 && c.T > 48 && tune.pipe != 0 && pipe_res >= 1 &&
                                   chunks <= (long)num_cu * pipe_res;
            // (beyond ONE round of resident workgroups the wave-per-chunk kernel is as fast: cfg3' at 6 144 episodes in chunks of
            // four 58 vs 61 us)
            const bool wg = wg_ok && tune.phase_flat != 0 && (tune.phase_flat == 1 || chunks <= (long)num_cu * wg_res);
            if (wg && pipe_form) {
                pa.h_pad = wg_h;
                pa.chunk = wgE;
                if (pipe_lds > kLdsDefault) {
                    hipError_t e = c.D == 7 ? allow_full_lds(k_traj_phase_dmp_pipe<2, 7>) : allow_full_lds(k_traj_phase_dmp_pipe<2>);
                    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
                }
                *kernel_name = "k_traj_phase<dmp,wg,pipe>";
                if (c.D == 7 && tune.pd_generic != 1)
                    hipLaunchKernelGGL((k_traj_phase_dmp_pipe<2, 7>), dim3((unsigned)chunks), dim3(256), pipe_lds, (hipStream_t)stream, pa);
                else
                    hipLaunchKernelGGL(k_traj_phase_dmp_pipe<2>, dim3((unsigned)chunks), dim3(256), pipe_lds, (hipStream_t)stream, pa);
                MPK_LAUNCH_CHECK();
                return MPK_OK;
            }
            if (wg) {
                pa.h_pad = wg_h;
                pa.chunk = wgE;
                long nb = chunks < (long)num_cu * wg_res ? chunks : (long)num_cu * wg_res;
                auto gow = [&](auto kern) -> int {
                    if (wg_lds > kLdsDefault) {
                        hipError_t e = allow_full_lds(kern);
                        if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
                    }
                    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(wgNTB == 5 ? 320 : 256), wg_lds, (hipStream_t)stream, pa);
                    MPK_LAUNCH_CHECK();
                    return MPK_OK;
                };
                *kernel_name = "k_traj_phase<dmp,wg>";
                if (wgNTB == 5) return gow(k_traj_phase_dmp_wg<2, 5>);
                if (wgNTB == 4) return KQ == 2 ? gow(k_traj_phase_dmp_wg<2, 4>) : gow(k_traj_phase_dmp_wg<4, 4>);
                return KQ == 2 ? gow(k_traj_phase_dmp_wg<2, 2>) : gow(k_traj_phase_dmp_wg<4, 2>);
            }
            *kernel_name = "k_traj_phase<dmp>";
            if (KQ == 2 && c.D == 7 && tune.pd_generic != 1) return go(k_traj_phase_dmp<2, 7>);      // (seven DoF compiled in: the kernel is issue bound)
            return KQ == 2 ? go(k_traj_phase_dmp<2>) : go(k_traj_phase_dmp<4>);
        }
    }
}
#endif  // MPK_DEVICE_ONLY

// ------------------------------------------------------------------------------------------------------------
// k_dmp_prestep (MPK_DMP_FIRST_IS_STEP): the boundary state advanced by ONE Euler step from init_time to the first grid
// time, with the forcing and the scaled-time increment at init_time -- the state the trajectory kernels then start
// from.  One lane per (episode, DoF); the row arithmetic of rbf_cols (both of its branches), operation for operation.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dmp_prestep(const DevCfg c, const float* __restrict__ params,
                                                     const float* __restrict__ init_pos,
                                                     const float* __restrict__ init_vel,
                                                     const float* __restrict__ init_time, const float init_time_shared,
                                                     float* __restrict__ pos1, float* __restrict__ vel1, const int B) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)B * c.D) return;
    const int b = (int)(e / c.D), d = (int)(e - (long)b * c.D);
    const float* prm = params + (size_t)b * c.P;
    float tau = c.tau, delay = c.delay;
    int o = 0;
    if (c.learn_tau) { tau = fminf(fmaxf(prm[o], c.tau_lo), c.tau_hi); ++o; }
    if (c.learn_delay) delay = fminf(fmaxf(prm[o], c.delay_lo), c.delay_hi);
    const float it = init_time ? init_time[b] : init_time_shared;
    const float t1 = c.base_times[0] + it;
    const float ds0 = scaled_time(t1, delay, tau) - scaled_time(it, delay, tau);
    const double x = phase_f64(c, it, tau, delay, ExpLiteral());
    const double* cen = c.tab;
    const double* bw = c.tab + c.n_total;
    // the forcing row at init_time: the SAME arithmetic as rbf_cols / rbf_row (product recurrence where the host enabled
    // it), so this sample is bit-identical to what the trajectory kernels produce for the same phase value
    const double mul = x * (double)c.ws;
    const float* w = prm + c.off + d * c.Kloc;
    float f0 = 0.0f;
    if (c.rbf_uniform) {
        RbfRecur s1(cen, bw, c.n_total, x, ExpLiteral());
        double sum = 0.0;
        for (int k = 0; k < c.n_total; ++k) sum += s1.next();
        const double scale = div_pos(mul, sum);
        RbfRecur s2(cen, bw, c.n_total, x, ExpLiteral());
        for (int k = 0; k < c.zs + c.nb; ++k) {
            const double ek = s2.next();
            if (k >= c.zs) f0 = fmaf((float)(ek * scale), w[k - c.zs], f0);
        }
    } else {
        double sum = 0.0;
        for (int k = 0; k < c.n_total; ++k) {
            const double dx = x - cen[k];
            sum += exp_nonpos(-(dx * dx * bw[k]) * 0.5);
        }
        const double scale = c.n_total > 1 ? div_pos(mul, sum) : mul;
        for (int k = 0; k < c.nb; ++k) {
            const double dx = x - cen[c.zs + k];
            const float h = (float)(exp_nonpos(-(dx * dx * bw[c.zs + k]) * 0.5) * scale);
            f0 = fmaf(h, w[k], f0);
        }
    }
    float y = init_pos[e];
    float z = init_vel[e] * tau;
    const float g = w[c.nb] * c.gs;
    const float t1_ = g - y;
    const float t2 = c.dmp_beta * t1_;
    const float t3 = t2 - z;
    const float t4 = c.dmp_alpha * t3;
    const float acc = t4 + f0;
    z = z + ds0 * acc;
    y = y + ds0 * z;
    pos1[e] = y;
    vel1[e] = div_tau(z, make_tau_div(tau));
}

#ifndef MPK_DEVICE_ONLY
int launch_dmp_prestep(const DevCfg& c, const float* params, const float* init_pos, const float* init_vel,
                       const float* init_time, float init_time_shared, float* pos1, float* vel1, int B, void* stream) {
    hipLaunchKernelGGL(k_dmp_prestep, dim3((unsigned)(((long)B * c.D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c,
                       params, init_pos, init_vel, init_time, init_time_shared, pos1, vel1, B);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

#ifndef MPK_DEVICE_ONLY
int launch_traj_rows(const DevCfg& c, const float* params, const float* init_pos, const float* init_vel,
                     const float* init_time, float init_time_shared, float* pos, float* vel, int32_t* range_flag,
                     int B, int num_cu, void* stream, const char** kernel_name, const Tuning& tune) {
    if (c.mp_type == MPK_MP_PROMP && c.T < 2) {
        set_error("promp needs at least two time steps for the finite-difference velocity");
        return MPK_EINVAL;
    }
    // wave-per-episode kernel whenever the shape fits it ("phase" 0: the workgroup-per-episode kernel below)
    const bool wave_kernel = tune.phase != 0;
    if (wave_kernel) {
        PhaseArgs pa{c, params, init_pos, init_vel, init_time, init_time_shared, pos, vel, range_flag, B, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        pa.wt = (double)B * c.T * c.D * 8.0 <= kWtBytes ? 1 : 0;
        if (tune.write_through >= 0) pa.wt = tune.write_through != 0 ? 1 : 0;
        const int rc = launch_traj_phase(c, pa, num_cu, stream, kernel_name, tune);
        if (rc != MPK_ENOTIMPL) return rc;
    }
    const int nrow = c.mp_type == MPK_MP_PRODMP ? 2 : 1;
    const size_t floats = (size_t)c.D * c.KT + (size_t)nrow * c.T * c.KT + (size_t)c.T * c.D +
                          (c.mp_type == MPK_MP_DMP ? (size_t)c.T * c.D : 0) + c.T + 8;
    const size_t lds = floats * sizeof(float);
    if (lds > kLdsPerCu) { set_error("trajectory too large for the per-episode kernel's LDS budget"); return MPK_EINVAL; }
    RowArgs ra{c, params, init_pos, init_vel, init_time, init_time_shared, pos, vel, range_flag, B};
    int blocks = B < num_cu * 8 ? B : num_cu * 8;
    auto go = [&](auto kern) -> int {
        if (lds > kLdsDefault) {
            hipError_t e = allow_full_lds(kern);
            if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, (hipStream_t)stream, ra);
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    };
    switch (c.mp_type) {
        case MPK_MP_PRODMP: *kernel_name = "k_traj_rows<prodmp>"; return go(k_traj_rows<MPK_MP_PRODMP>);
        case MPK_MP_PROMP: *kernel_name = "k_traj_rows<promp>"; return go(k_traj_rows<MPK_MP_PROMP>);
        default: *kernel_name = "k_traj_rows<dmp>"; return go(k_traj_rows<MPK_MP_DMP>);
    }
}
#endif  // MPK_DEVICE_ONLY

}  // namespace mpk
