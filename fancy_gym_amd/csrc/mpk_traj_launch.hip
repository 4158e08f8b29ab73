// k_build_shared (phase / exponential-kernel evaluation for a phase all episodes share) and launch_traj_shared: the rule
// that picks a shared-phase trajectory kernel family, its work decomposition and its store policy for a launch.
#include <cstdio>

#include "mpk_tile.h"
#include "mpk_traj_quad.h"   // kQuadImg, kPipeGroups: the LDS budgets the rule checks
#include "mpk_traj_pipe.h"
#include "mpk_traj_stream.h" // kChunkGroups
#include "mpk_traj_ring.h"   // kRingThreads, kRingSyncInts

namespace mpk {

// a shape only k_traj_wide can take (more than kMaxKP columns, or more than kMaxD DoF): that kernel reads the position rows
// (prodmp: + velocity rows) of the k-major table and nothing else, so such a handle's slots hold just those -- no
// finite-difference operand rows, no step-major copy (a sixth of the full promp table: ~1.7 MB instead of 10 MB per slot at
// K = 1000, T = 200)
bool shared_tables_lean(const DevCfg& c) { return c.KP > kMaxKP || c.D > kMaxD; }

size_t shared_tables_floats(const DevCfg& c, int* TS, int* n_out) {
    const int TP = (c.T + 15) / 16 * 16;
    const int ts = ((TP + 15) / 32) * 32 + 16;  // TS % 32 == 16: the two k rows of a 32-lane LDS read hit disjoint banks
    const bool lean = shared_tables_lean(c);
    const int no = c.mp_type == MPK_MP_PRODMP ? 2 : (c.mp_type == MPK_MP_PROMP && !lean ? 3 : 1);
    *TS = ts;
    *n_out = no;
    // the k-major table A [n_out][KP][TS] (MFMA fragment loads) followed by its step-major copy At [TS][n_out * KP]
    // (one contiguous row per time step: the serial role of k_traj_split reads it with scalar loads)
    return (lean ? 1 : 2) * (size_t)no * c.KP * ts;
}

// ------------------------------------------------------------------------------------------------------------
// k_build_shared: one block; A[(j*KP + k)*TS + t], aux[t]
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_build_shared(const DevCfg c, const float init_time, float* __restrict__ A,
                                                      float* __restrict__ aux, const int TS, const int n_out, const int lean,
                                                      int32_t* __restrict__ idx_out, int32_t* __restrict__ flag) {
    const int tid = threadIdx.x, T = c.T, KP = c.KP;
    for (int i = tid; i < n_out * KP * TS; i += 256) A[i] = 0.0f;
    for (int i = tid; i < TS; i += 256) aux[i] = 0.0f;
    __syncthreads();
    if (c.dmp_resp) {
        // DMP with a phase all episodes share, as a CONTRACTION (round 5).  The explicit Euler recurrence of the reference's DMP
        // (the CPU restatement's dmp_trajectory: a = alpha (beta (g - y) - z) + f; z += ds a; y += ds z; vel = z / tau) is
        // linear in (w, g, y_b, v_b): pos[t] = sum_k R_pos[t, k] x_k with x = (w_1 .. w_nb, g, y_b, v_b), and R the response of THE
        // SAME Euler map -- same fp32 step sizes ds, same order of operations -- to the unit inputs, run here once per (init_time, T)
        // in float64 and rounded once to fp32.  Superposition of explicit Euler, not the ODE's analytic solution: first-order
        // convergence in dt stays what the reference's integrator gives (tests/test_gpu_ode.py).  Columns as ProDMP's: weights, goal,
        // y_b, v_b -- the two-output matrix-core kernels run unchanged (and with them fused actions and the closed loop).  The host
        // takes this route only where the Euler map is stable (alpha ds < 1: mpk_host.cpp dmp_response_ok), i.e. |R| = O(1).
        const int nb = c.nb;
        // (1) forcing rows phi_k(x) x weights_scale as the serial kernels contract them (fp32), aux = fp32 step sizes
        for (int t = tid; t < T; t += 256) {
            const float time = c.base_times[t] + init_time;
            const double x = phase_f64(c, time, c.tau, c.delay, ExpLiteral());
            rbf_cols(c, x, x * (double)c.ws, A + t, TS);
            if (t < T - 1) {
                const float s0 = scaled_time(time, c.delay, c.tau);
                const float s1 = scaled_time(c.base_times[t + 1] + init_time, c.delay, c.tau);
                aux[t] = s1 - s0;
            }
        }
        __syncthreads();
        // (2) one thread per unit input runs the T - 1 steps (a forcing value is read before its slot takes the response)
        if (tid < nb + 3) {
            const int k = tid;
            const double al = (double)c.dmp_alpha, be = (double)c.dmp_beta, tau = (double)c.tau, itau = div_pos(1.0, (double)c.tau);
            const double G = k == nb ? (double)c.gs : 0.0;
            double y = k == nb + 1 ? 1.0 : 0.0, z = k == nb + 2 ? tau : 0.0;      // z_0 = tau v_b
            // sixteen steps per trip: their forcing values and step sizes are read TOGETHER, in front of the trip's stores (the response
            // takes the forcing value's slot, so the compiler cannot hoist a load over the stores before it: step by step the loop was T
            // dependent round trips to the table -- hundreds of us at T = 200 whenever a plan brings a new init_time; review of round 5)
            constexpr int U = 16;
            float fn[U], dn[U];
            auto fetch = [&](const int t0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u < T ? t0 + u : T - 1;
                    fn[u] = k < nb ? A[(size_t)k * TS + t] : 0.0f;
                    dn[u] = aux[t];
                }
            };
            fetch(0);
            for (int t0 = 0; t0 < T; t0 += U) {
                float fv[U], dv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { fv[u] = fn[u]; dv[u] = dn[u]; }
                if (t0 + U < T) fetch(t0 + U);           // the next trip's values: in flight while this trip's steps run
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u;
                    if (t < T) {
                        A[(size_t)(0 * KP + k) * TS + t] = (float)y;
                        A[(size_t)(1 * KP + k) * TS + t] = (float)(z * itau);
                        if (t < T - 1) {
                            const double acc = al * (be * (G - y) - z) + (double)fv[u];
                            z = z + (double)dv[u] * acc;
                            y = y + (double)dv[u] * z;
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < TS; i += 256) aux[i] = 0.0f;
    } else if (c.mp_type == MPK_MP_PRODMP) {
        const float sb = scaled_time(init_time, c.delay, c.tau);
        const int idxb = min(prodmp_index(sb, c.scaled_dt), c.n_pc - 1);
        ProdmpBC bc;
        prodmp_bc(c, idxb, bc);
        if (idx_out && tid == 0) idx_out[T] = idxb;
        for (int t = tid; t < T; t += 256) {
            const float time = c.base_times[t] + init_time;
            const float s = scaled_time(time, c.delay, c.tau);
            if (s > (float)c.len_factor) atomicOr(flag, 1);
            const int idx = min(prodmp_index(s, c.scaled_dt), c.n_pc - 1);
            if (idx_out) idx_out[t] = idx;
            double xi[4];
            prodmp_xi(c, bc, idx, xi);
            for (int k = 0; k < c.KT; ++k) {
                float h, hv;
                prodmp_col(c, bc, idx, xi, k, (double)c.tau, div_pos(1.0, (double)c.tau), &h, &hv);
                A[(size_t)(0 * KP + k) * TS + t] = h;
                A[(size_t)(1 * KP + k) * TS + t] = hv;
            }
        }
    } else if (c.mp_type == MPK_MP_PROMP) {
        for (int t = tid; t < T; t += 256) {
            const float time = c.base_times[t] + init_time;
            const double x = phase_f64(c, time, c.tau, c.delay, ExpLiteral());
            rbf_cols(c, x, (double)c.ws, A + t, TS);
            if (c.KT > c.nb) A[(size_t)c.nb * TS + t] = 1.0f;  // zero-padded family: + init_pos
        }
        __syncthreads();
        // velocity = forward difference: rows (t+1, t), last row repeats (T-1, T-2)
        for (int t = tid; t < T; t += 256) {
            const int th = t < T - 1 ? t + 1 : T - 1, tl = t < T - 1 ? t : T - 2;
            for (int k = 0; k < c.KT && !lean; ++k) {
                A[(size_t)(1 * KP + k) * TS + t] = A[(size_t)k * TS + th];
                A[(size_t)(2 * KP + k) * TS + t] = A[(size_t)k * TS + tl];
            }
            // reciprocal of the fp32 time step (one IEEE divide per row here instead of one per output element later)
            aux[t] = 1.0f / ((c.base_times[th] + init_time) - (c.base_times[tl] + init_time));
        }
    } else {  // DMP: forcing rows phi*x, aux = diff of the fp32 scaled times
        for (int t = tid; t < T; t += 256) {
            const float time = c.base_times[t] + init_time;
            const double x = phase_f64(c, time, c.tau, c.delay, ExpLiteral());
            rbf_cols(c, x, x * (double)c.ws, A + t, TS);
            if (t < T - 1) {
                const float s0 = scaled_time(time, c.delay, c.tau);
                const float s1 = scaled_time(c.base_times[t + 1] + init_time, c.delay, c.tau);
                aux[t] = s1 - s0;
            }
        }
    }
    // step-major copy behind the k-major table: one contiguous row per step; with two outputs (prodmp) the row is
    // interleaved [pos_0 vel_0 pos_1 vel_1 ..] -- the operand pairs of the packed fp32 FMA the serial role contracts with
    __syncthreads();
    if (lean) return;
    const int RS = n_out * KP;
    float* At = A + (size_t)RS * TS;
    for (int i = tid; i < RS * TS; i += 256) {
        const int t = i / RS, e = i - t * RS;
        const int jk = n_out == 2 ? (e & 1) * KP + (e >> 1) : e;
        At[i] = A[(size_t)jk * TS + t];
    }
}

#ifndef MPK_DEVICE_ONLY
int launch_build_shared(const DevCfg& c, float init_time, const SharedTables& st, int32_t* idx_out,
                        int32_t* range_flag, void* stream) {
    if (c.mp_type == MPK_MP_PROMP && c.T < 2) {
        set_error("promp needs at least two time steps for the finite-difference velocity");
        return MPK_EINVAL;
    }
    hipLaunchKernelGGL(k_build_shared, dim3(1), dim3(256), 0, (hipStream_t)stream, c, init_time, st.A, st.aux, st.TS,
                       st.n_out, shared_tables_lean(c) ? 1 : 0, idx_out, range_flag);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

#ifndef MPK_DEVICE_ONLY
#ifndef MPK_AMALGAMATED
// defined in mpk_traj_family.hip (one translation unit per MP type)
template <int MP>
int launch_traj_ct(const TrajArgs& ta, const ActArgs& aa, int ct, bool stream_mode, bool write_through, bool bulk,
                   int quad, int blocks, size_t lds, void* stream, bool split, bool pipe);
extern template int launch_traj_ct<MPK_MP_PROMP>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
extern template int launch_traj_ct<MPK_MP_DMP>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
extern template int launch_traj_ct<MPK_MP_PRODMP>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
// defined in mpk_traj_ring.hip (one translation unit per MP type)
template <int MP>
int launch_traj_ring(const TrajArgs& ta, const ActArgs& aa, int ct, int blocks, size_t lds, void* stream);
extern template int launch_traj_ring<MPK_MP_PROMP>(const TrajArgs&, const ActArgs&, int, int, size_t, void*);
extern template int launch_traj_ring<MPK_MP_DMP>(const TrajArgs&, const ActArgs&, int, int, size_t, void*);
extern template int launch_traj_ring<MPK_MP_PRODMP>(const TrajArgs&, const ActArgs&, int, int, size_t, void*);
#endif
#endif

#ifndef MPK_DEVICE_ONLY
#ifndef MPK_AMALGAMATED
// defined in mpk_episode.hip (one translation unit per MP type)
template <int MP>
int launch_episode_kernel(const TrajArgs& ta, const ActArgs& aa, const EpArgs& ea, int ct, int nq, int rwd, int blocks, size_t lds,
                          void* stream);
extern template int launch_episode_kernel<MPK_MP_PROMP>(const TrajArgs&, const ActArgs&, const EpArgs&, int, int, int, int, size_t, void*);
extern template int launch_episode_kernel<MPK_MP_DMP>(const TrajArgs&, const ActArgs&, const EpArgs&, int, int, int, int, size_t, void*);
extern template int launch_episode_kernel<MPK_MP_PRODMP>(const TrajArgs&, const ActArgs&, const EpArgs&, int, int, int, int, size_t, void*);
#endif

// mpk_episode_return: plan + controller + plant + reward + aggregation of a `verbose < 2` step in one launch (k_episode_return).
// MPK_ENOTIMPL where the tables do not fit beside the images (long horizons): the caller's separate launches take those.
// fp32 thresholds of a float64 interval: an fp32 position lies in [low, high] exactly when it lies in [up(low), down(high)]
static float f32_at_least(double x) {
    float f = (float)x;
    if ((double)f < x) f = nextafterf(f, INFINITY);
    return f;
}
static float f32_at_most(double x) {
    float f = (float)x;
    if ((double)f > x) f = nextafterf(f, -INFINITY);
    return f;
}
static void fill_gate_args(const GateDev* gate, const float* params, int D, TrajArgs& ta, ActArgs& aa) {
    ta.gate_valid = nullptr; ta.gate_penalty = nullptr; ta.gate_raw = nullptr; ta.gate_check_td = 0;
    ta.gate_tb[0] = ta.gate_tb[1] = ta.gate_db[0] = ta.gate_db[1] = 0.0;
    if (!gate) return;
    ta.gate_valid = gate->valid; ta.gate_penalty = gate->penalty;
    ta.gate_raw = gate->raw_params ? gate->raw_params : params;
    ta.gate_check_td = gate->check_td;
    ta.gate_tb[0] = gate->tau_b[0]; ta.gate_tb[1] = gate->tau_b[1];
    ta.gate_db[0] = gate->delay_b[0]; ta.gate_db[1] = gate->delay_b[1];
    for (int d = 0; d < D; ++d) {
        aa.glo[d] = gate->lo[d]; aa.ghi[d] = gate->hi[d];
        aa.glo32[d] = f32_at_least(gate->lo[d]); aa.ghi32[d] = f32_at_most(gate->hi[d]);
    }
}

int launch_episode_return(const DevCfg& c, const SharedTables& st, const float* params, const float* init_pos, const float* init_vel,
                          const RolloutDev& rc, double* q_state, double* qd_state, const int32_t* n_steps, const ReplanDev* rp,
                          int reward_type, const double* goal, const int32_t* step0, int steps_before_reward, int agg, double* ret,
                          int32_t* seg_out, int B, int num_cu, void* stream, const char** kernel_name, const Tuning& tune,
                          const GateDev* gate) {
    TrajArgs ta{};
    ta.wpb = 4; ta.ring_parts = 1;
    if (rp) ta.rp = *rp;
    ta.q_state = q_state; ta.qd_state = qd_state; ta.n_steps = n_steps; ta.plant_dt = rc.dt;
    ta.c = c; ta.A = st.A; ta.aux = st.aux; ta.TS = st.TS;
    ta.params = params; ta.init_pos = init_pos; ta.init_vel = init_vel;
    ta.B = B;
    int sh = 0;
    while ((1 << sh) < c.D) ++sh;
    ta.sh = sh;
    const int NTW = 16 >> sh;
    ta.G = (B + NTW - 1) / NTW;
    const int SEG = 16 * c.D;
    ta.pitch = SEG; ta.cps = SEG / 4 > 0 ? SEG / 4 : 1; ta.inv_cps = 65536u / (unsigned)ta.cps + 1u; ta.vec_ok = 1;
    ActArgs aa{};
    for (int d = 0; d < c.D; ++d) { aa.pg[d] = rc.pg[d]; aa.dg[d] = rc.dg[d]; aa.lo[d] = rc.lo[d]; aa.hi[d] = rc.hi[d]; }
    fill_gate_args(gate, params, c.D, ta, aa);
    EpArgs ea{};
    ea.ret = ret; ea.goal = goal; ea.step0 = step0; ea.seg_out = seg_out; ea.steps_before_reward = steps_before_reward; ea.agg = agg;
    ea.km = c.KP / 4;
    const size_t table_bytes = ((size_t)st.n_out * c.KP * st.TS + st.TS) * sizeof(float);
    // groups per wave: four while that still gives every SIMD a wave (the chain is latency bound: more lanes per instruction),
    // else two, else one ("quad" 2 / 3 / 4 force four / two / one)
    const long simds = (long)num_cu * 4;
    int nq = ta.G >= 4 * simds ? 4 : (ta.G >= 2 * simds ? 2 : 1);
    if (tune.quad == 2) nq = 4; else if (tune.quad == 3) nq = 2; else if (tune.quad == 4) nq = 1;
    while (reward_type && nq > 1 && nq * NTW > 4 * kEpMaxPass) nq >>= 1;      // the reward pass keeps two accumulators per lane
    const size_t img = reward_type ? kEpImg : kEpImgPlain;
    auto lds_of = [&](int n, int w) { return table_bytes + ((size_t)w * n * img + (size_t)w * kEpSlots * kEpSlotInts) * sizeof(float); };
    while (nq > 1 && lds_of(nq, 4) > kLdsPerCu) nq >>= 1;
    if (lds_of(nq, 4) > kLdsPerCu || (reward_type && nq * NTW > 4 * kEpMaxPass)) { set_error("trajectory too long for the episode kernel's LDS budget"); return MPK_ENOTIMPL; }
    // eight waves per workgroup where that puts more waves on a CU than four-wave workgroups do (they share one table copy)
    const long units = ((long)ta.G + nq - 1) / nq;
    const long w4 = (long)(kLdsPerCu / lds_of(nq, 4)) * 4, w8 = lds_of(nq, 8) <= kLdsPerCu ? (long)(kLdsPerCu / lds_of(nq, 8)) * 8 : 0;
    int wpb = w8 > w4 && units >= 8L * num_cu ? 8 : 4;
    if (tune.tiles_wpb == 4 || tune.tiles_wpb == 8) wpb = tune.tiles_wpb == 8 && w8 > 0 ? 8 : 4;     // ("tiles_wpb" 4 / 8: A/B runs, tests)
    ea.wpb = wpb;
    const size_t lds = lds_of(nq, wpb);
    const long per_cu = (long)(kLdsPerCu / lds) < 1 ? 1 : (long)(kLdsPerCu / lds);
    long blocks = (units + wpb - 1) / wpb;
    const long cap = (long)num_cu * (per_cu > 8 ? 8 : per_cu);
    if (blocks > cap) blocks = cap;
    if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
    const int ct = rc.controller_type + 3;
    const bool pd = c.mp_type == MPK_MP_PRODMP;
    *kernel_name = reward_type ? (pd ? "k_episode_return<prodmp,reacher>" : "k_episode_return<promp,reacher>")
                               : (pd ? "k_episode_return<prodmp>" : "k_episode_return<promp>");
    switch (c.mp_type) {
        case MPK_MP_PRODMP: return launch_episode_kernel<MPK_MP_PRODMP>(ta, aa, ea, ct, nq, reward_type, (int)blocks, lds, stream);
        case MPK_MP_PROMP: return launch_episode_kernel<MPK_MP_PROMP>(ta, aa, ea, ct, nq, reward_type, (int)blocks, lds, stream);
        default: set_error("internal: the episode kernel takes promp / prodmp rows"); return MPK_EINVAL;
    }
}

int launch_traj_shared(const DevCfg& c, const SharedTables& st, const float* params, const float* init_pos,
                       const float* init_vel, float* pos, float* vel, float* actions, const RolloutDev* rc,
                       const double* c_pos, const double* c_vel, double* q_state, double* qd_state,
                       const int32_t* n_steps, int B, int num_cu, void* stream, const char** kernel_name,
                       const Tuning& tune, const ReplanDev* rp, unsigned* ticket, int* fault, const GateDev* gate) {
    TrajArgs ta;
    ta.fault = fault;
    ta.nrt_magic = 0; ta.gstride = 0; ta.wt = 0; ta.flat_img = 0;
    ta.ring_np = 0; ta.ring_ns = 0; ta.ring_m = 0; ta.ring_nbuf = 0; ta.ring_nc = 0; ta.ring_aw = 0; ta.ring_dbg = tune.ring_dbg > 0 ? tune.ring_dbg : 0;
    // the bits that leave outputs unwritten (1 no production, 2 no stores, 128 a batch never published; open loop: 8 no input loads)
    // count only after mpk_set_option(.., "ablations", 1) -- measurements and fault injection, never by accident
    if (tune.ablations != 1) ta.ring_dbg &= ~(1 | 2 | 128 | (q_state ? 0 : 8)); ta.burst = 0; ta.inorder = 0; ta.lean = 0; ta.wpb = 4; ta.ring_ctr = nullptr; ta.ring_tb = 0; ta.ring_parts = 1;
    if (rp) ta.rp = *rp;
    const bool closed = q_state != nullptr;
    const bool gated = gate != nullptr;       // validity gate: the lane-quarter closed-loop kernels (k_traj_quad / duo / mono: gate_pass) and k_traj_pipe
    if (gated && !(closed && actions)) { set_error("the validity gate belongs to the closed-loop step"); return MPK_EINVAL; }
    ta.q_state = q_state; ta.qd_state = qd_state; ta.n_steps = n_steps; ta.plant_dt = rc ? rc->dt : 0.0;
    ta.c = c; ta.A = st.A; ta.aux = st.aux; ta.TS = st.TS;
    ta.params = params; ta.init_pos = init_pos; ta.init_vel = init_vel;
    ta.pos = pos; ta.vel = vel; ta.actions = actions; ta.c_pos = c_pos; ta.c_vel = c_vel;
    ta.B = B;
    int sh = 0;
    while ((1 << sh) < c.D) ++sh;  // DP = next power of two >= D (<= 16)
    ta.sh = sh;
    const int NTW = 16 >> sh;
    ta.G = (B + NTW - 1) / NTW;
    const bool act = actions != nullptr;
    const int nst = 2 + (act ? 1 : 0);
    const int SEG = 16 * c.D, seg4 = SEG / 4, TD = c.T * c.D;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool ptr_ok = aligned16(pos) && aligned16(vel) && (!act || aligned16(actions));
    // T*D % 4 != 0: episodes start 0..3 floats past a 16-byte boundary -> shifted staging image (one spare chunk per
    // episode segment), if the segments of a group still fit the 64 lanes of a wave
    // (misaligned output pointers take the generic store path, whose staging image is never shifted)
    ta.shifted = (ptr_ok && TD % 4 != 0 && NTW * (seg4 + 1) <= 64 && NTW * (SEG + 4) <= kStageStride) ? 1 : 0;
    ta.td3 = TD & 3;
    ta.pitch = ta.shifted ? SEG + 4 : SEG;
    ta.cps = ta.shifted ? seg4 + 1 : seg4;
    ta.inv_cps = 65536u / (unsigned)ta.cps + 1u;
    ta.vec_ok = ptr_ok && (TD % 4 == 0 || ta.shifted);
    ActArgs aa{};
    int ct = -1;
    if (act) {
        ct = rc->controller_type + (closed ? 3 : 0);
        for (int d = 0; d < c.D; ++d) { aa.pg[d] = rc->pg[d]; aa.dg[d] = rc->dg[d]; aa.lo[d] = rc->lo[d]; aa.hi[d] = rc->hi[d]; }
    }
    fill_gate_args(gate, params, c.D, ta, aa);
    const int NRT = (c.T + 15) / 16;
    const long max_waves = (long)num_cu * 32;     // 8 waves per SIMD resident
    // work decomposition: episode-major once the outputs stop being cache resident (or when it is the only option)
    const size_t table_bytes = ((size_t)st.n_out * c.KP * st.TS + st.TS) * sizeof(float);
    const double out_bytes = (double)B * c.T * c.D * 4.0 * nst;
    const int ov = tune.mapping == 1 || tune.mapping == 2 ? tune.mapping : 0;   // mpk_set_option "mapping"
    // closed loop, promp / prodmp, outputs cache resident: tile-major with a serial role (k_traj_split).  "split" 0 / 1
    // switches it off / forces it; a forced episode-major variant ("mapping" 2, "quad" 0 / 2 / 3 / 4, "bulk" 2) wins
    const bool variant_forced = ov == 2 || tune.quad == 0 || tune.quad >= 2 || tune.bulk == 2;
    // (its serial role stores actions as aligned float4 chunks: trajectories and the last row tile must be whole chunks)
    const int last_rows = c.T - (c.T - 1) / 16 * 16;
    const bool split_shape = ptr_ok && TD % 4 == 0 && (last_rows * c.D) % 4 == 0;
    // episode-major producer / consumer pipeline (k_traj_pipe): the default closed-loop kernel whenever its tables and
    // images fit; "pipe" 0 / 1 switches it off / forces it; "split" 1 forces the tile-major kernel with a serial role
    // Automatic up to three 5-wave workgroups per CU (B = 6144 at 7 DoF): measured against the best one-wave kernel
    // (profiles/r02_closed_loop.md) full step 10.1 vs 11.9 us at B = 2048, 11.8 vs 14.1 at 4096, 22.1 vs 19.7 at 8192;
    // 25-of-100-step plan 8.1 vs 8.8, 9.5 vs 11.6, 17.2 vs 16.5 -- beyond that the launch is store-bound and the barrier
    // per row tile only makes the store stream burstier.
    const bool pipe_fits = table_bytes + 2 * kPipeGroups * kQuadImg * sizeof(float) <= kLdsDefault;
    const long pipe_units = ((long)ta.G + kPipeGroups - 1) / kPipeGroups;
    // (with the validity gate -- k_traj_pipe<.., GATE>, 153 registers: two workgroups per CU -- only while ONE workgroup per CU holds the
    // launch: cfg5, us gated / ungated: 1 024 episodes 28.8 / 24.9 (k_traj_mono<.., gate>: 37.5), 2 048: 29.2 / 25.2, 4 096: 55.6 / 28.7)
    const bool pipe = closed && c.mp_type != MPK_MP_DMP && pipe_fits && tune.split != 1 &&
                      (tune.pipe == 1 || (tune.pipe != 0 && !variant_forced && pipe_units <= (gated ? 1L : 3L) * num_cu));
    const bool split = !pipe && closed && !gated && c.mp_type != MPK_MP_DMP && split_shape && tune.split == 1;
    // (trajectory-only launches of the shapes k_traj_flat takes -- two workgroups of whole-trajectory images per CU -- go episode-major from
    // kFlatTrajBytes on: round 5, cfg2's shape, us tiles / flat: 8 192 episodes 10.7 / 11.0, 12 288: 14.5 / 14.1, 16 384: 19.2 / 18.2)
    const bool flat_takes_it = !act && !closed && c.mp_type != MPK_MP_DMP && ptr_ok && (c.T * c.D) % 4 == 0 && tune.flat != 0 &&
                               table_bytes + (size_t)4 * nst * (((size_t)NTW * c.T * c.D + 3) / 4 * 4) * sizeof(float) <= kLdsHalf;
    const double stream_from = flat_takes_it && tune.bulk < 0 ? kFlatTrajBytes : kCachedBytes;
    bool stream_mode = !split && (c.mp_type == MPK_MP_DMP || closed || out_bytes > stream_from);
    if (c.mp_type != MPK_MP_DMP && !closed && ov == 1) stream_mode = false;
    if (ov == 2 && !split) stream_mode = true;       // a forced k_traj_split stays tile-major (its tiles role needs that geometry)
    if ((tune.flat == 1 || tune.ring >= 1) && !closed && c.mp_type != MPK_MP_DMP && !split) stream_mode = true;   // forced k_traj_flat (where it applies)
    if (stream_mode && table_bytes + 4 * kStageFloats * sizeof(float) > kLdsDefault) {
        // the caller falls back: per-episode kernels for dmp, trajectory + rollout launches for the closed loop
        if (c.mp_type == MPK_MP_DMP || closed) { set_error("trajectory too long for the episode-major kernel's LDS budget"); return MPK_ENOTIMPL; }
        stream_mode = false;
    }
    // write-through stores for the cache-resident tile-major case (mpk_set_option "write_through" overrides, for A/B runs)
    bool write_through = !stream_mode;
    ta.wt = stream_mode && out_bytes <= kWtBytes ? 1 : 0;
    if (tune.write_through >= 0) {
        write_through = tune.write_through != 0 && !stream_mode;
        ta.wt = tune.write_through != 0 && stream_mode ? 1 : 0;
    }
    // write-through stores address an output array through one buffer resource with 32-bit byte offsets (wt_store16): arrays
    // of 2 GiB and more (never cache resident anyway; only a forced option gets here) take plain stores
    if ((double)B * c.T * c.D * 4.0 >= 2147483648.0) { write_through = false; ta.wt = 0; }
    int blocks;
    size_t lds = 0;
    bool bulk = false;
    // serial-recurrence variants (DMP, closed loop): four (or two) groups per wave, recurrences in parallel on the lane
    // quarters; needs its staging (52 / 26 KB) + the tables within 64 KB.  quad = groups per wave, 0 = k_traj_stream.
    // mpk_set_option "quad": 0 off, 2 force four, 3 force two, 4 force one (A/B runs, tests)
    int quad = 0;
    {
        // static staging (fp32 images) + the tables
        auto fits = [&](int nq) {
            return table_bytes + (4 * nq * kQuadImg) * sizeof(float) <= kLdsDefault;
        };
        const bool serial_variant = stream_mode && (c.mp_type == MPK_MP_DMP || closed);
        const int quad_mode = tune.quad < 0 ? 1 : tune.quad;
        const long units4 = (ta.G + 3) / 4, units2 = (ta.G + 1) / 2;
        // automatic (A/B-measured: profiles/r01_replan_end_to_end.md, profiles/r04_closed_loop.md):
        //   four per wave  while that gives two waves per SIMD AND the outputs still fit the memory-side cache (kWtBytes):
        //                  cfg3 DMP at B = 16384 33.8 us vs 38.7 with two; closed loop at 16384 30.8 vs 34.7, at 32768 56.7 vs
        //                  69.1 (round 4: 256 registers = two waves per SIMD; 271 = one before, and two groups won everywhere);
        //   two per wave   below that (one wave per SIMD exposes every LDS / MFMA latency: closed loop at B = 8192
        //                  20.4 -> 19.2 us) AND above it: at HBM-streaming sizes the launch is bound by its store pattern
        //                  (stores alone 153 of 166 us), and a four-group wave keeps 12 - 16 output streams open (DMP at
        //                  B = 32768 82.4 vs 78.5 us, at 262144 608 vs 595; closed loop at B = 65536 182 vs 166 us);
        //   one per wave   for the closed loop at a few thousand episodes (cfg4 episodes at B = 2048: 0.061 -> 0.052 ms)
        if (serial_variant && quad_mode != 0) {
            if (quad_mode == 2) quad = fits(4) ? 4 : 0;
            else if (quad_mode == 3) quad = fits(2) ? 2 : 0;
            else if (quad_mode == 4) quad = fits(1) ? 1 : 0;
            else if (c.mp_type == MPK_MP_DMP && fits(4)) {
                // DMP (round 4, second session: sweep in steps of 2 048 episodes, profiles/r04_serial_quantization.md): the launches
                // are ROUNDS of resident waves -- two per SIMD with four groups per wave, three with two (cfg3's shape) -- and a
                // launch that needs one wave more than a round takes most of a second one: cfg3 at 18 432 episodes 53 us with four
                // groups (2 304 units for 2 048 places) against 42 with two.  Four groups per wave exactly where they fit ONE round and
                // two groups per wave would not (12 289 - 16 384 episodes of cfg3: 30.6 - 32.6 us against 38.5 - 38.9)
                if (units4 <= (long)num_cu * 8 && units2 > (long)num_cu * 12) quad = 4;
                else if (fits(2) && units2 >= (long)num_cu * 6) quad = 2;      // (below: one group per wave, k_traj_stream -- cfg3 at 4 096: 17.0 vs 18.5 us)
            }
            // (closed loop, second session: ... and four per wave ALSO where two per wave would need a second round of resident waves
            // (two per SIMD) and four per wave fit one -- 8 193 - 16 383 episodes at 7 DoF: a launch with one wave too many for a
            // round takes most of another; 8 704: 26.7 -> 25.3 us, 12 288: 28.0 -> 26.6, 14 336: 33.5 -> 27.8)
            else if (fits(4) && out_bytes <= kWtBytes &&
                     (units4 >= (long)num_cu * 8 || (closed && units2 > (long)num_cu * 8 && units4 <= (long)num_cu * 8))) quad = 4;
            else if (fits(2) && units2 >= (long)num_cu * 4) quad = 2;
            else if (closed && fits(1)) quad = 1;
        }
        if (gated && quad == 0 && !pipe) {
            // (a forced "quad" 0, or tables beyond the lane-quarter kernels' LDS: the caller's separate launches)
            if (serial_variant && tune.quad != 0 && fits(1)) quad = 1;
            else return MPK_ENOTIMPL;
        }
    }

    // wave-specialised store engine (k_traj_ring): ONE persistent workgroup per CU whose LDS holds the tables + a ring of NBUF
    // batch buffers of M whole-trajectory group images; producer waves fill them (contraction + controller epilogue), store-engine
    // waves write each batch's arrays as contiguous runs; batches are handed out in order from one device counter.  Open loop,
    // promp / prodmp (the serial-recurrence variants need four groups per wave: mpk_traj_ring.hip).  Automatic once a launch writes more than kRingBytes (A/B
    // measurements: profiles/r04_ring.md); mpk_set_option "ring": 0 off, 1 force; "ring_np" / "ring_ns" / "ring_m" /
    // "ring_parts": producer waves, store-engine waves, groups per batch, waves per group.  A forced episode-major variant
    // ("quad", "bulk", "flat" 1, "pipe" 1, "split" 1) wins over the automatic choice.
    bool ring = false;
    {
        const bool forced_other = tune.flat == 1 || tune.bulk >= 0 || tune.quad >= 0 || tune.pipe == 1 || tune.split == 1 || ov == 1;
        // (trajectory-only launches that k_traj_flat takes -- two workgroups of whole-trajectory images per CU, see below -- stay there
        // up to kRingTrajBytes: round 5, cfg2's shape, us flat / ring: 65 536 episodes 68.8 / 80.4, 131 072: 134.7 / 149.3, 262 144:
        // 259.5 / 282.7, 524 288: 533.8 / 551.9, 1 048 576: 1 151 / 1 092.  With actions the ring stays ahead from kRingBytes on.)
        const bool want = tune.ring == 1 || (tune.ring < 0 && !forced_other && out_bytes > (flat_takes_it ? kRingTrajBytes : kRingBytes));
        const int TD_ = c.T * c.D;
        int NS = tune.ring_ns > 0 ? tune.ring_ns : 2;
        int NP = tune.ring_np > 0 ? tune.ring_np : 8;
        if (NS > 8) NS = 8;
        if (NP + NS > kRingThreads / 64) NP = kRingThreads / 64 - NS;
        const size_t fixed = table_bytes + kRingSyncInts * sizeof(int);
        const int gimg = NTW * TD_;                                   // floats per (array, group) image, exactly
        auto buf_of = [&](int m) { return (size_t)nst * m * gimg * sizeof(float); };
        // groups per batch: as asked ("ring_m"), else the most (<= 4) that leave two batch buffers in the CU's LDS; a batch
        // must be a whole number of float4 per array (its runs are written as aligned 16-byte chunks)
        auto fits = [&](int m) { return fixed + 2 * buf_of(m) <= kLdsPerCu && ((long)m * gimg) % 4 == 0; };
        int M = tune.ring_m > 0 ? tune.ring_m : 4;
        while (M > 1 && !fits(M)) --M;
        if (ptr_ok && want && !closed && c.mp_type != MPK_MP_DMP && !split && tune.ring != 2 && fits(M)) {
            const size_t buf_bytes = buf_of(M);
            long nbuf = (long)((kLdsPerCu - fixed) / buf_bytes);
            if (nbuf * M > 32) nbuf = 32 / M;                         // 32 slots of sync counters
            if (nbuf > 3) nbuf = 3;
            // waves per group: with few slots (long horizons: one group's image fills a buffer) the producers share a group's
            // row tiles, so that all of them have work
            int P = tune.ring_parts > 0 ? tune.ring_parts : (int)((NP + nbuf * M - 1) / (nbuf * M));
            if (P > NRT) P = NRT;
            if (P > 8) P = 8;
            if (P < 1) P = 1;
            ta.flat_img = gimg;
            ta.ring_np = NP; ta.ring_ns = NS; ta.ring_m = M; ta.ring_nbuf = (int)nbuf; ta.ring_parts = P;
            // in-order dynamic batch assignment: tickets of TB batches from one counter word (~88 tickets / us at most: a
            // ticket must be worth well over 100 KB of output); zero at handle creation, zeroed again by the launch's last workgroup
            ta.ring_ctr = ticket;
            ta.ring_tb = (int)((kRingTicketBytes + buf_bytes - 1) / buf_bytes);
            {
                // ... but never so coarse that a workgroup sees fewer than ~16 tickets: wave 0 takes its first 3 - 4 tickets in ONE
                // atomic, so with tickets of five batches the first workgroups to start walked off with 15 - 20 batches each -- at
                // 12 288 episodes (6 batches per workgroup on average) a few dozen workgroups did all the work while the rest found
                // the counter past the end: THE ring's "fixed" 33 - 38 us below ~30 000 episodes (round 5, tools/ring_floor_probe.py:
                // no production + no stores 34 us with tickets, 8 us with static batches), and a coarse tail at 65 536
                const long batches_ = ((long)ta.G + M - 1) / M;
                const long fine = batches_ / (16L * (batches_ < (long)num_cu ? batches_ : (long)num_cu));
                if (tune.ring_tb > 0) ta.ring_tb = tune.ring_tb;
                else if ((long)ta.ring_tb > fine) ta.ring_tb = (int)fine;
            }
            if (ta.ring_tb < 1) ta.ring_tb = 1;
            // (the kernel keeps 2 + ceil(2 NP / (TB M P)) tickets in flight in 8 slots: a ticket covers at least NP / 2 work units)
            while (2 * ta.ring_tb * M * P < NP) ++ta.ring_tb;
            ring = true;
            stream_mode = true; quad = 0; bulk = false;
            ta.wt = out_bytes <= kWtBytes ? 1 : 0;
            if (tune.write_through >= 0) ta.wt = tune.write_through != 0 ? 1 : 0;
            if ((double)B * c.T * c.D * 4.0 >= 2147483648.0) ta.wt = 0;
            lds = fixed + (size_t)nbuf * buf_bytes;
            const long batches = ((long)ta.G + M - 1) / M;
            blocks = (int)(batches < (long)num_cu ? batches : (long)num_cu);
        }
    }
    // closed loop on the ring (k_traj_ring<.., closed>): producers + store engine for pos / vel, consumer waves for the recurrences
    // (mpk_traj_ring.h).  LDS: tables + NBUF (pos | vel) batch buffers of four groups + one 4 KB action tile set per consumer.
    bool pipe_sel = pipe;
    if (closed && !gated && c.mp_type != MPK_MP_DMP && act && ptr_ok && TD % 4 == 0 && (c.D == 5 || c.D == 7) && c.KP <= 8 && !split &&
        tune.ring != 0 && tune.ring != 2) {
        const bool forced_other = tune.quad >= 0 || tune.pipe == 1 || tune.split == 1 || ov != 0 || tune.bulk >= 0;
        const bool want = tune.ring == 1 || (tune.ring < 0 && !forced_other && !pipe && out_bytes > kRingClosedBytes);
        // geometry by the sweep in profiles/r04_ring_closed.md: the fewer waves store, the better (one engine wave), production and
        // recurrences need eight and four waves to keep three batch buffers turning
        int NS = tune.ring_ns > 0 ? tune.ring_ns : 1;
        int NP = tune.ring_np > 0 ? tune.ring_np : 8;
        int NC = tune.ring_nc > 0 ? tune.ring_nc : 3;
        if (NS > 4) NS = 4;
        const int AW = (ta.ring_dbg & 8) ? 0 : 1;                            // "ring_dbg" 8: the consumers store their action tiles themselves
        if (NP + NS + NC * (1 + AW) > kRingThreadsClosed / 64) NP = kRingThreadsClosed / 64 - NS - NC * (1 + AW);
        int M = tune.ring_m > 0 && tune.ring_m < 4 ? tune.ring_m : 4;          // groups per batch = lane quarters of a consumer
        const int gimg = NTW * TD;
        const size_t fixed = table_bytes + kRingSyncInts * sizeof(int);
        const size_t stage = (size_t)NC * 4 * kStageStride * sizeof(float);
        // (longer horizons: fewer groups per batch, while that leaves at least two batch buffers)
        while (M > 1 && fixed + stage + 2 * (size_t)2 * M * gimg * sizeof(float) > kLdsPerCu) --M;
        const size_t buf_bytes = (size_t)2 * M * gimg * sizeof(float);
        long nbuf = fixed + stage < kLdsPerCu ? (long)((kLdsPerCu - fixed - stage) / buf_bytes) : 0;
        if (nbuf * M > 32) nbuf = 32 / M;
        if (nbuf > 4) nbuf = 4;
        // automatic only with four groups per batch (one per lane quarter of a consumer): a 200-step horizon leaves room for two, and
        // the ring then loses to the lane-quarter kernels (round 5, cfg3's shape closed loop, us ring / duo: 32 768 episodes 192 / 124,
        // 65 536: 368 / 280, 131 072: 722 / 522 -- tools/dmp_closed_choice.py)
        const bool ring_pays = tune.ring == 1 || M == 4;
        if (want && ring_pays && NP >= 1 && nbuf >= 2 && ((long)M * gimg) % 4 == 0 && 16 * c.D * NTW <= kStageStride) {
            ta.flat_img = gimg;
            ta.ring_np = NP; ta.ring_ns = NS; ta.ring_nc = NC; ta.ring_aw = AW; ta.ring_m = M; ta.ring_nbuf = (int)nbuf; ta.ring_parts = 1;
            ta.ring_ctr = ticket;
            ta.ring_tb = (int)((kRingTicketBytes + buf_bytes * 3 / 2 - 1) / (buf_bytes * 3 / 2));   // (a batch writes 1.5 x its buffer)
            if (ta.ring_tb < 2) ta.ring_tb = 2;
            if (tune.ring_tb > 0) ta.ring_tb = tune.ring_tb;
            while (2 * ta.ring_tb * M < NP) ++ta.ring_tb;
            ring = true; pipe_sel = false;
            stream_mode = true; quad = 0; bulk = false;
            ta.wt = out_bytes <= kWtBytes ? 1 : 0;
            if (tune.write_through >= 0) ta.wt = tune.write_through != 0 ? 1 : 0;
            if ((double)B * c.T * c.D * 4.0 >= 2147483648.0) ta.wt = 0;
            lds = fixed + (size_t)nbuf * buf_bytes + stage;
            const long batches = ((long)ta.G + M - 1) / M;
            blocks = (int)(batches < (long)num_cu ? batches : (long)num_cu);
        }
    }
    if (ring) {
        // (set up above)
    } else if (pipe_sel) {
        quad = 0; bulk = false;
        lds = table_bytes;
        const long units = (ta.G + kPipeGroups - 1) / kPipeGroups;
        const long cap = (long)num_cu * 6;                                // 5-wave workgroups: one resident round
        blocks = (int)(units < cap ? units : cap);
        if (blocks >= 8) blocks = blocks / 8 * 8;                         // XCD-contiguous remap needs a multiple of 8
        if (blocks < 1) blocks = 1;
        ta.lean = units > 2 * (long)num_cu ? 1 : 0;                        // (see k_traj_pipe: 81 instead of 100 registers)
    } else if (quad) {
        lds = table_bytes;
        const long units = (ta.G + quad - 1) / quad;
        const long waves = units < max_waves ? units : max_waves;
        blocks = (int)((waves + 3) / 4);
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
        // "serial_order" 1: short-lived workgroups in address order (one unit per wave); 2: persistent without the XCD remap
        if (tune.serial_order >= 1) {
            ta.inorder = 1;
            if (tune.serial_order == 1) blocks = (int)((units + 3) / 4);
        }
    } else if (stream_mode) {
        lds = table_bytes;
        // bulk input staging: chunk blocks must be float4-sized / aligned and fit the per-lane register image
        const int EPC = kChunkGroups * NTW;
        const size_t img_floats = (size_t)EPC * (c.P + 2 * c.D + 4 * c.D);
        const size_t lds_bulk = table_bytes + 4 * 2 * img_floats * sizeof(float);
        bulk = (EPC * c.P) % 4 == 0 && (EPC * c.D) % 4 == 0 && (EPC * c.P) / 4 <= 128 && (EPC * c.D) / 2 <= 64 &&
               aligned16(params) && aligned16(init_pos) && aligned16(init_vel) &&
               (!act || closed || (aligned16(c_pos) && aligned16(c_vel))) &&
               lds_bulk + 4 * kStageFloats * sizeof(float) <= kLdsDefault;
        // mpk_set_option "bulk": 0 disables, 2 forces it below the size threshold too (tests); default: HBM-streaming sizes only
        const int bulk_mode = tune.bulk < 0 ? 1 : tune.bulk;
        // automatic: only when the outputs stream to HBM AND the 4x coarser work units still fill the chip; the
        // latency-bound DMP recurrence prefers occupancy over input staging
        const long chunks = (ta.G + kChunkGroups - 1) / kChunkGroups;
        const bool auto_ok = out_bytes > kCachedBytes && chunks >= max_waves / 2 && c.mp_type != MPK_MP_DMP;
        bulk = bulk && bulk_mode != 0 && (bulk_mode == 2 || auto_ok);
        long units = ta.G;
        if (bulk) { lds = lds_bulk; units = (ta.G + kChunkGroups - 1) / kChunkGroups; }
        long waves = units < max_waves ? units : max_waves;
        if (tune.phase_waves > 0 && waves > (long)num_cu * tune.phase_waves) waves = (long)num_cu * tune.phase_waves;   // (A/B runs: waves per CU)
        // whole-trajectory images (k_traj_flat): open loop, promp / prodmp, aligned outputs, T * D a multiple of 4, and
        // two workgroups' images + tables within a CU's LDS.  Automatic once the outputs stream to HBM (A/B on the
        // streaming row, profiles/r03_streaming.md); mpk_set_option "flat": 0 off, 1 force
        const int flat_img = (NTW * TD + 3) / 4 * 4;
        const size_t lds_flat = table_bytes + (size_t)4 * nst * flat_img * sizeof(float);
        const bool flat_ok = !closed && c.mp_type != MPK_MP_DMP && ptr_ok && TD % 4 == 0 && lds_flat <= kLdsHalf;
        if (flat_ok && tune.flat != 0 && (tune.flat == 1 || (out_bytes > stream_from && tune.bulk < 0))) {
            ta.flat_img = flat_img;
            bulk = false;
            // (write-through while the outputs fit the memory-side cache: kWtBytes)
            lds = lds_flat + (tune.lds_pad > 0 ? (size_t)tune.lds_pad * 1024 : 0);   // "lds_pad": occupancy experiments
            // workgroups per CU: TWO, also where the LDS holds three (round 5: three -- twelve waves, twelve write streams per CU -- were
            // 7 - 20 % slower than two at every size from 12 288 to 1 M episodes of cfg2's trajectory-only shape: 32 768 episodes 33.5 ->
            // 31.2 us, 65 536: 73.0 -> 68.8, 262 144: 318 -> 260; with three arrays two were all that fitted, and one is slower again)
            // ("phase_waves" 4 / 8 / 12: one / two / three workgroups, for A/B runs)
            const long wg_cap = tune.phase_waves >= 4 ? tune.phase_waves / 4 : 2;
            const long wg = (long)(kLdsPerCu / lds) < wg_cap ? (long)(kLdsPerCu / lds) : wg_cap;
            const long resident = (long)num_cu * (wg < 1 ? 1 : wg) * 4;   // 4-wave workgroups, persistent
            waves = ta.G < resident ? ta.G : resident;
            // the DoF count compiled in for the shapes the reference registers MP environments with (k_traj_flat_d, mpk_traj_ring.h);
            // "ring_dbg" bit 64: the generic kernel (A/B runs, tests)
            if ((c.D == 5 || c.D == 7) && c.KP <= 8 && !(ta.ring_dbg & 64)) ta.burst = 2;
        }
        blocks = (int)((waves + 3) / 4);
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;                  // XCD-contiguous remap needs a multiple of 8
        const int img = flat_img;                                        // floats per (array, group) image
        if (flat_ok && tune.ring == 2) {
            // short-lived workgroups (k_traj_burst): one batch of M groups per workgroup, WPG waves per group
            int M = tune.ring_m > 0 ? tune.ring_m : 4;
            int WPG = tune.ring_np > 0 ? tune.ring_np : 1;
            if (M > 8) M = 8;
            if (M * WPG > 8) WPG = 8 / M < 1 ? 1 : 8 / M;
            const size_t bytes = (size_t)nst * M * img * sizeof(float);
            if (table_bytes + bytes <= kLdsPerCu) {
                ta.flat_img = img;
                ta.burst = 1; ta.ring_m = M; ta.ring_np = WPG; ta.ring_ns = 0; ta.ring_nbuf = 0;
                bulk = false;
                lds = table_bytes + bytes;
                blocks = (int)(((long)ta.G + M - 1) / M);
            }
        }
    } else {
        const long items = (long)ta.G * NRT;
        long ipw = (items + max_waves - 1) / max_waves;                  // items per wave, balanced
        if (tune.ipw > 0) ipw = tune.ipw;                                // mpk_set_option "ipw" (A/B runs)
        // the kernel divides wave ids by NRT with a 32-bit multiply-high: exact while #waves < 2^32 / NRT
        const long wave_cap = (long)((1ull << 32) / (unsigned long long)NRT) - 8 * NRT;
        if ((items + ipw - 1) / ipw > wave_cap) ipw = (items + wave_cap - 1) / wave_cap;
        const long waves = (items + ipw - 1) / ipw;
        const int wpb = (tune.tiles_wpb == 1 || tune.tiles_wpb == 2) && !split ? tune.tiles_wpb : 4;   // A/B: smaller workgroups
        ta.wpb = wpb;
        blocks = (int)((waves + wpb - 1) / wpb);
        {   // #waves % NRT == 0, and a multiple of 8 blocks for the XCD remap once there are that many
            int g8 = 8, r = NRT;
            while (r) { const int t = g8 % r; g8 = r; r = t; }           // gcd(8, NRT)
            const int unit = blocks >= 8 ? NRT / g8 * 8 : NRT;           // lcm(8, NRT) or NRT
            blocks = (blocks + unit - 1) / unit * unit;
        }
        ta.gstride = blocks * wpb / NRT;
        ta.nrt_magic = NRT > 1 ? (unsigned)((1ull << 32) / (unsigned long long)NRT) + 1u : 0u;
    }
    if (blocks < 1) blocks = 1;
    if (!stream_mode && !pipe_sel && ta.gstride <= 0) { set_error("internal: tile-major launch without its group stride"); return MPK_EINVAL; }
    if (!stream_mode && tune.lds_pad > 0) lds = (size_t)tune.lds_pad * 1024;     // A/B runs: caps the workgroups per CU
    if (stream_mode && !pipe_sel && !ta.flat_img && tune.lds_pad > 0) lds += (size_t)tune.lds_pad * 1024;   // episode-major: EXTRA dynamic LDS (occupancy experiments)
    ta.ser_blocks = 0;
    if (split) {
        // serial-role workgroups first (they are the long pole and must start first), capped at one resident round of the chip
        const int EPW = 64 >> sh;                              // episodes per serial-role wave: one lane per (episode, DoF)
        const long units = ((long)B + EPW - 1) / EPW;
        long sb = (units + 3) / 4;
        const long cap = (long)num_cu * 8;
        if (sb > cap) sb = cap;
        ta.ser_blocks = (unsigned)sb;
        blocks += (int)sb;
    }
    if (ring || ta.burst) {
        const bool pd = c.mp_type == MPK_MP_PRODMP;
        *kernel_name = ta.burst == 2 ? (pd ? (act ? "k_traj_flat<prodmp,act>" : "k_traj_flat<prodmp>") : (act ? "k_traj_flat<promp,act>" : "k_traj_flat<promp>"))
                     : ta.burst ? (pd ? (act ? "k_traj_burst<prodmp,act>" : "k_traj_burst<prodmp>") : (act ? "k_traj_burst<promp,act>" : "k_traj_burst<promp>"))
                     : closed ? (pd ? "k_traj_ring<prodmp,closed>" : "k_traj_ring<promp,closed>")
                                : (pd ? (act ? "k_traj_ring<prodmp,act>" : "k_traj_ring<prodmp>") : (act ? "k_traj_ring<promp,act>" : "k_traj_ring<promp>"));
        switch (c.mp_type) {
            case MPK_MP_PRODMP: return launch_traj_ring<MPK_MP_PRODMP>(ta, aa, ct, blocks, lds, stream);
            case MPK_MP_PROMP: return launch_traj_ring<MPK_MP_PROMP>(ta, aa, ct, blocks, lds, stream);
            default: return launch_traj_ring<MPK_MP_DMP>(ta, aa, -1, blocks, lds, stream);
        }
    }
    if (gated) {
        const bool pd = c.mp_type == MPK_MP_PRODMP;
        *kernel_name = pipe_sel ? (pd ? "k_traj_pipe<prodmp,closed,gate>" : "k_traj_pipe<promp,closed,gate>")
                     : quad == 4 ? (pd ? "k_traj_quad<prodmp,closed,gate>" : "k_traj_quad<promp,closed,gate>")
                     : quad == 2 ? (pd ? "k_traj_duo<prodmp,closed,gate>" : "k_traj_duo<promp,closed,gate>")
                                 : (pd ? "k_traj_mono<prodmp,closed,gate>" : "k_traj_mono<promp,closed,gate>");
        if (c.mp_type == MPK_MP_DMP) return MPK_ENOTIMPL;
        return pd ? launch_traj_ct<MPK_MP_PRODMP>(ta, aa, ct, stream_mode, write_through, bulk, quad, blocks, lds, stream, split, pipe_sel)
                  : launch_traj_ct<MPK_MP_PROMP>(ta, aa, ct, stream_mode, write_through, bulk, quad, blocks, lds, stream, split, pipe_sel);
    }
    switch (c.mp_type) {
        case MPK_MP_PRODMP:
            *kernel_name = pipe_sel ? "k_traj_pipe<prodmp,closed>" : split ? "k_traj_split<prodmp,closed>" : closed ? (quad == 4 ? "k_traj_quad<prodmp,closed>" : quad == 2 ? "k_traj_duo<prodmp,closed>" : quad == 1 ? "k_traj_mono<prodmp,closed>" : "k_traj_stream<prodmp,closed>") : ta.flat_img ? (act ? "k_traj_flat<prodmp,act>" : "k_traj_flat<prodmp>") : stream_mode ? (act ? "k_traj_stream<prodmp,act>" : "k_traj_stream<prodmp>")
                                       : (act ? "k_traj_tiles<prodmp,act>" : "k_traj_tiles<prodmp>");
            return launch_traj_ct<MPK_MP_PRODMP>(ta, aa, ct, stream_mode, write_through, bulk, quad, blocks, lds, stream, split, pipe_sel);
        case MPK_MP_PROMP:
            *kernel_name = pipe_sel ? "k_traj_pipe<promp,closed>" : split ? "k_traj_split<promp,closed>" : closed ? (quad == 4 ? "k_traj_quad<promp,closed>" : quad == 2 ? "k_traj_duo<promp,closed>" : quad == 1 ? "k_traj_mono<promp,closed>" : "k_traj_stream<promp,closed>") : ta.flat_img ? (act ? "k_traj_flat<promp,act>" : "k_traj_flat<promp>") : stream_mode ? (act ? "k_traj_stream<promp,act>" : "k_traj_stream<promp>")
                                       : (act ? "k_traj_tiles<promp,act>" : "k_traj_tiles<promp>");
            return launch_traj_ct<MPK_MP_PROMP>(ta, aa, ct, stream_mode, write_through, bulk, quad, blocks, lds, stream, split, pipe_sel);
        default:
            *kernel_name = quad == 4 ? "k_traj_quad<dmp>" : quad == 2 ? "k_traj_duo<dmp>" : quad == 1 ? "k_traj_mono<dmp>" : "k_traj_stream<dmp>";
            return launch_traj_ct<MPK_MP_DMP>(ta, aa, -1, true, false, bulk, quad, blocks, lds, stream, false, false);
    }
}
#endif  // MPK_DEVICE_ONLY

}  // namespace mpk
