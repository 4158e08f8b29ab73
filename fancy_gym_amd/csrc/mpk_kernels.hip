// gfx950 (CDNA4 / MI355X) kernels of libmpk.so.  Compiled with -ffp-contract=off: every fused multiply-add in
// this file is an explicit fmaf()/MFMA, every other a*b+c rounds twice exactly like the reference's separate
// torch / numpy ops.
//
//   k_build_shared   phase / exponential-kernel evaluation: per-time-step basis rows for a phase that all episodes
//                    share (tau, delay, init_time equal) -> A tables [n_out][KP][TS] (k-major, fp32) + aux[TS]
//   k_traj_tiles /   the [T x K] . [K x D] contraction on the matrix cores (v_mfma_f32_16x16x4_f32), 16 time steps x
//   k_traj_stream    16 (episode, DoF) columns per tile, fused epilogue (ProMP finite-difference velocity, DMP Euler
//                    integration, optional PD action), wave-private LDS transpose, float4 stores; tile-major for
//                    cache-resident batches, episode-major (LDS-staged basis tables) for HBM-streaming batches
//   k_traj_rows      per-episode phase (learned tau / delay, per-episode init_time): table gather / RBF evaluation
//                    per row, fp32 fmaf chains in the same k order as the MFMA
//   k_pd_rollout     tracking-controller + plant loop in float64 (black_box_wrapper.py:175-203)
//   k_replan_advance integer replanning bookkeeping; k_validity: joint-limit / bound check reduction
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "mpk_internal.h"

namespace mpk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MPK_LAUNCH_CHECK()                                                          \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            set_error(std::string("kernel launch: ") + hipGetErrorString(e_));      \
            return MPK_EHIP;                                                        \
        }                                                                           \
    } while (0)

size_t shared_tables_floats(const DevCfg& c, int* TS, int* n_out) {
    const int TP = (c.T + 15) / 16 * 16;
    const int ts = ((TP + 15) / 32) * 32 + 16;  // TS % 32 == 16: the two k rows of a 32-lane LDS read hit disjoint banks
    const int no = c.mp_type == MPK_MP_PRODMP ? 2 : (c.mp_type == MPK_MP_PROMP ? 3 : 1);
    *TS = ts;
    *n_out = no;
    return (size_t)no * c.KP * ts;
}

// ------------------------------------------------------------------------------------------------------------
// device helpers shared by the shared-phase builder and the per-episode kernel
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float scaled_time(float t, float delay, float tau) {
    // left-bounded linear phase in fp32: max((t - delay) / tau, 0); IEEE division
    return fmaxf((t - delay) / tau, 0.0f);
}

__device__ __forceinline__ int prodmp_index(float s, float scaled_dt) {
    // times_to_indices: round-half-even of the fp32 quotient -- the bit-exact integer part of the path
    return (int)rintf(s / scaled_dt);
}

struct ProdmpBC {
    int idxb;
    double a, b, c, d;  // dy2_b/det, dy1_b/det, y1_b/det, y2_b/det
};

__device__ __forceinline__ void prodmp_bc(const DevCfg& c, int idxb, ProdmpBC& bc) {
    const int N = c.n_pc;
    const double y1b = c.tab[idxb], y2b = c.tab[N + idxb], dy1b = c.tab[2 * N + idxb], dy2b = c.tab[3 * N + idxb];
    const double det = y1b * dy2b - y2b * dy1b;
    bc.idxb = idxb;
    bc.a = dy2b / det; bc.b = dy1b / det; bc.c = y1b / det; bc.d = y2b / det;
}

__device__ __forceinline__ void prodmp_xi(const DevCfg& c, const ProdmpBC& bc, int idx, double xi[4]) {
    const int N = c.n_pc;
    const double y1 = c.tab[idx], y2 = c.tab[N + idx], dy1 = c.tab[2 * N + idx], dy2 = c.tab[3 * N + idx];
    xi[0] = bc.a * y1 - bc.b * y2;
    xi[1] = bc.c * y2 - bc.d * y1;
    xi[2] = bc.a * dy1 - bc.b * dy2;
    xi[3] = bc.c * dy2 - bc.d * dy1;
}

// Column k (< nb+3) of the ProDMP position / velocity rows at table index idx, as consumed by the contraction with the
// RAW parameter column x = [w_0..w_{nb-1}, g, y_b, ydot_b]:
//   k <  nb   : H_k  * weights_goal_scale[k]            (0 if the weights are disabled)
//   k == nb   : H_g  * weights_goal_scale[nb]           (0 if the goal is disabled)
//   k == nb+1 : xi1  (+ H_g for a relative goal: goal = scale*g + y_b)
//   k == nb+2 : xi2 * tau                               (v_b = tau * ydot_b)
// and the velocity row additionally carries the 1/tau of  vel = (...)/tau.  Everything is folded in float64 and
// rounded ONCE to fp32.
__device__ __forceinline__ void prodmp_col(const DevCfg& c, const ProdmpBC& bc, int idx, const double xi[4], int k,
                                           double tau, float* h, float* hv) {
    const int N = c.n_pc, K = c.nb + 1;
    const double* PB = c.tab + 4 * (size_t)N;
    const double* VB = PB + (size_t)N * K;
    auto hcol = [&](int kk, double* hp, double* hvp) {
        const double pb = PB[(size_t)bc.idxb * K + kk], vb = VB[(size_t)bc.idxb * K + kk];
        *hp = PB[(size_t)idx * K + kk] - (xi[0] * pb + xi[1] * vb);
        *hvp = VB[(size_t)idx * K + kk] - (xi[2] * pb + xi[3] * vb);
    };
    double p = 0.0, v = 0.0;
    if (k < K) {
        const bool off = k < c.nb ? c.disable_weights != 0 : c.disable_goal != 0;
        if (!off) {
            hcol(k, &p, &v);
            const double sc = (VB + (size_t)N * K)[k];   // weights_goal_scale[k], appended to the device tables
            p *= sc; v *= sc;
        }
    } else if (k == K) {
        p = xi[0]; v = xi[2];
        if (c.relative_goal) {
            double gp, gv;
            hcol(c.nb, &gp, &gv);
            p += gp; v += gv;
        }
    } else {
        p = xi[1] * tau; v = xi[3] * tau;
    }
    *h = (float)p;
    *hv = (float)(v / tau);
}

// bounded phase in float64 from an fp32 time value and fp32-held tau/delay (promp / dmp rows)
__device__ __forceinline__ double phase_f64(const DevCfg& c, float time, float tau, float delay, double* s_out) {
    const double s = ((double)time - (double)delay) / (double)tau;
    if (s_out) *s_out = s;
    if (c.phase_type == MPK_PHASE_LINEAR) return fmin(fmax(s, 0.0), 1.0);
    return exp(-(double)c.alpha_phase * fmax(s, 0.0));
}

// normalised RBF row: writes nb learnable columns scaled by `mul` (column zs.. of the zero-padded family)
__device__ __forceinline__ void rbf_cols(const DevCfg& c, double x, double mul, float* out, int stride) {
    const double* cen = c.tab;
    const double* bw = c.tab + c.n_total;
    double sum = 0.0;
    for (int k = 0; k < c.n_total; ++k) {
        const double dx = x - cen[k];
        sum += exp(-(dx * dx * bw[k]) / 2.0);
    }
    for (int k = 0; k < c.nb; ++k) {
        const double dx = x - cen[c.zs + k];
        double v = exp(-(dx * dx * bw[c.zs + k]) / 2.0);
        if (c.n_total > 1) v = v / sum;
        out[(size_t)k * stride] = (float)(v * mul);
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_build_shared: one block; A[(j*KP + k)*TS + t], aux[t]
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_build_shared(const DevCfg c, const float init_time, float* __restrict__ A,
                                                      float* __restrict__ aux, const int TS, const int n_out,
                                                      int32_t* __restrict__ idx_out, int32_t* __restrict__ flag) {
    const int tid = threadIdx.x, T = c.T, KP = c.KP;
    for (int i = tid; i < n_out * KP * TS; i += 256) A[i] = 0.0f;
    for (int i = tid; i < TS; i += 256) aux[i] = 0.0f;
    __syncthreads();
    if (c.mp_type == MPK_MP_PRODMP) {
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
                prodmp_col(c, bc, idx, xi, k, (double)c.tau, &h, &hv);
                A[(size_t)(0 * KP + k) * TS + t] = h;
                A[(size_t)(1 * KP + k) * TS + t] = hv;
            }
        }
    } else if (c.mp_type == MPK_MP_PROMP) {
        for (int t = tid; t < T; t += 256) {
            const float time = c.base_times[t] + init_time;
            const double x = phase_f64(c, time, c.tau, c.delay, nullptr);
            rbf_cols(c, x, (double)c.ws, A + t, TS);
            if (c.KT > c.nb) A[(size_t)c.nb * TS + t] = 1.0f;  // zero-padded family: + init_pos
        }
        __syncthreads();
        // velocity = forward difference: rows (t+1, t), last row repeats (T-1, T-2)
        for (int t = tid; t < T; t += 256) {
            const int th = t < T - 1 ? t + 1 : T - 1, tl = t < T - 1 ? t : T - 2;
            for (int k = 0; k < c.KT; ++k) {
                A[(size_t)(1 * KP + k) * TS + t] = A[(size_t)k * TS + th];
                A[(size_t)(2 * KP + k) * TS + t] = A[(size_t)k * TS + tl];
            }
            // reciprocal of the fp32 time step (one IEEE divide per row here instead of one per output element later)
            aux[t] = 1.0f / ((c.base_times[th] + init_time) - (c.base_times[tl] + init_time));
        }
    } else {  // DMP: forcing rows phi*x, aux = diff of the fp32 scaled times
        for (int t = tid; t < T; t += 256) {
            const float time = c.base_times[t] + init_time;
            const double x = phase_f64(c, time, c.tau, c.delay, nullptr);
            rbf_cols(c, x, x * (double)c.ws, A + t, TS);
            if (t < T - 1) {
                const float s0 = scaled_time(time, c.delay, c.tau);
                const float s1 = scaled_time(c.base_times[t + 1] + init_time, c.delay, c.tau);
                aux[t] = s1 - s0;
            }
        }
    }
}

int launch_build_shared(const DevCfg& c, float init_time, const SharedTables& st, int32_t* idx_out,
                        int32_t* range_flag, void* stream) {
    if (c.mp_type == MPK_MP_PROMP && c.T < 2) {
        set_error("promp needs at least two time steps for the finite-difference velocity");
        return MPK_EINVAL;
    }
    hipLaunchKernelGGL(k_build_shared, dim3(1), dim3(256), 0, (hipStream_t)stream, c, init_time, st.A, st.aux, st.TS,
                       st.n_out, idx_out, range_flag);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// The [T x K] . [K x D] contraction on the matrix cores (v_mfma_f32_16x16x4_f32) + fused epilogues.
//
// Tile = 16 time steps x 16 (episode, DoF) columns, K = 4*KM <= 16.  A fragments = basis rows (with weights_scale /
// goal_scale / tau / relative goal folded in at build time); B fragments = RAW parameters / boundary conditions
// gathered from HBM/L2 in fragment layout (wave-uniform base pointers + lane-constant 32-bit offsets, straight-line
// code, prefetched one episode group ahead).  The C tile is transposed through a wave-private LDS buffer so that
// every output array of a tile leaves as ONE coalesced float4 store instruction.  Wave-level indices live in SGPRs.
//
// Two work decompositions of the same tile code (tools/store_probe.hip, profiles/r01_store_patterns.md):
//   k_traj_tiles   tile-major: a wave owns ONE row tile (A fragments stay in registers) and walks episode groups.
//                  Maximum parallelism for small batches whose outputs stay cache resident.
//   k_traj_stream  episode-major: a wave owns an episode group and walks its row tiles in order, A fragments come
//                  from a per-workgroup LDS copy of the basis tables.  Every wave writes long contiguous runs,
//                  which is what the HBM write path needs at large batch (4.9 vs 3.0 TB/s for the same bytes).
//                  DMP always runs here (the Euler recurrence is serial in t).
// CT: fused controller: -1 none; MPK_CTRL_* (0..2) = open loop against a frozen state (c_pos, c_vel);
//     3 + MPK_CTRL_* = CLOSED loop with the double-integrator plant integrated in the kernel (episode-major only).
// ------------------------------------------------------------------------------------------------------------
struct TrajArgs {
    DevCfg c;
    const float* A;
    const float* aux;
    int TS;
    const float* params;
    const float* init_pos;
    const float* init_vel;
    float* pos;
    float* vel;
    float* actions;
    const double* c_pos;
    const double* c_vel;
    int B, sh, G, vec_ok;
    // LDS staging geometry: `pitch` floats per episode (16*D, +4 when the image is shifted), `cps` float4 chunks per
    // episode segment, inv_cps = 65536 / cps + 1.  shifted: T*D is not a multiple of 4, so episode b starts
    // ((b & 3) * (T*D & 3)) & 3 floats past a 16-byte boundary; its tile image is staged with the same offset so
    // that 16-byte-aligned LDS chunks map onto 16-byte-aligned HBM chunks (partial chunks at both ends go scalar)
    int pitch, cps, shifted, td3;
    unsigned inv_cps;
    // closed-loop rollout fused into the episode-major kernel (CT >= 3)
    double* q_state;       // [B, D] plant position, in/out
    double* qd_state;      // [B, D] plant velocity, in/out
    const int32_t* n_steps;  // [B] executed steps of this plan (NULL = T)
    double plant_dt;
};

struct ActArgs {
    double pg[kMaxD], dg[kMaxD], lo[kMaxD], hi[kMaxD];
};

constexpr int kStageStride = 256;   // floats between output arrays in the wave's LDS staging area (>= NTW*16*D)
constexpr int kStageFloats = 4 * kStageStride;   // pos | vel | actions or DMP forcing | controller constants

enum : int { XK_ZERO = 0, XK_PARAM = 1, XK_IPOS = 2, XK_IVEL = 3 };

// which raw input feeds element k of a DoF's extended parameter column, and its offset inside the DoF's local block
template <int MP>
__device__ __forceinline__ int x_kind(const DevCfg& c, int k, int* loc) {
    *loc = 0;
    if (MP == MPK_MP_PRODMP) {
        const int nb = c.nb;
        if (k < nb) { *loc = k; return c.disable_weights ? XK_ZERO : XK_PARAM; }
        if (k == nb) { *loc = c.disable_weights ? 0 : nb; return c.disable_goal ? XK_ZERO : XK_PARAM; }
        if (k == nb + 1) return XK_IPOS;
        if (k == nb + 2) return XK_IVEL;
        return XK_ZERO;
    } else if (MP == MPK_MP_PROMP) {
        if (k < c.nb) { *loc = k; return XK_PARAM; }
        if (k == c.nb && c.KT > c.nb) return XK_IPOS;
        return XK_ZERO;
    } else {
        if (k < c.nb) { *loc = k; return XK_PARAM; }
        return XK_ZERO;
    }
}

// lane-constant description of a lane's role in the 16x16 tile machinery
template <int KM>
struct LaneMap {
    int col, q, bl, d, dsafe, NTW;
    bool dvalid;
    bool isp[KM], isip[KM], isiv[KM];
    unsigned poff[KM];   // element offset of B-fragment element m inside the group's params block
    unsigned ioff;       // element offset inside the group's init_pos / init_vel / c_pos / c_vel block
    unsigned wofs;       // LDS transpose: write offset of (row 4q, this column)
    int sseg, w4;        // episode-in-group and float offset of the float4 this lane stores
    unsigned rofs, gofs; // LDS read offset / global offset (relative to the tile base) of that float4
};

template <int MP, int KM>
__device__ __forceinline__ LaneMap<KM> make_lane_map(const TrajArgs& a, int lane) {
    const DevCfg& c = a.c;
    LaneMap<KM> L;
    const int D = c.D, DP = 1 << a.sh;
    L.NTW = 16 >> a.sh;
    L.col = lane & 15; L.q = lane >> 4;
    L.bl = L.col >> a.sh; L.d = L.col & (DP - 1);
    L.dvalid = L.d < D;
    L.dsafe = L.dvalid ? L.d : D - 1;
#pragma unroll
    for (int m = 0; m < KM; ++m) {
        int loc;
        const int kind = x_kind<MP>(c, 4 * m + L.q, &loc);
        L.isp[m] = L.dvalid && kind == XK_PARAM;
        L.isip[m] = L.dvalid && kind == XK_IPOS;
        L.isiv[m] = L.dvalid && kind == XK_IVEL;
        L.poff[m] = (unsigned)(L.bl * c.P + c.off + L.dsafe * c.Kloc + loc);
    }
    L.ioff = (unsigned)(L.bl * D + L.dsafe);
    L.wofs = (unsigned)(L.bl * a.pitch + 4 * L.q * D + L.d);
    L.sseg = (int)(((unsigned)lane * a.inv_cps) >> 16);
    L.w4 = (lane - L.sseg * a.cps) * 4;
    L.rofs = (unsigned)(L.sseg * a.pitch + L.w4);
    L.gofs = (unsigned)(L.sseg * c.T * D + L.w4);
    return L;
}

// floats by which episode b's trajectories start past a 16-byte boundary (0 unless the image is shifted)
__device__ __forceinline__ unsigned ep_shift(const TrajArgs& a, int b) {
    return a.shifted ? (((unsigned)b & 3u) * (unsigned)a.td3) & 3u : 0u;
}

// raw inputs of one episode group for this lane (plain loads, no control flow)
template <int KM>
struct GroupIn {
    float raw[KM];
    float ip, iv;
    double cp, cv;
};

template <int MP, bool ACT, int KM>
__device__ __forceinline__ GroupIn<KM> load_group(const TrajArgs& a, const LaneMap<KM>& L, int g) {
    const DevCfg& c = a.c;
    GroupIn<KM> in;
    // the last group may be ragged: clamp its missing episodes onto the group's first one (computed, never stored)
    const int b0 = g * L.NTW;
    const bool bv = b0 + L.bl < a.B;
    const float* pb = a.params + (size_t)b0 * c.P;
    const unsigned io = bv ? L.ioff : (unsigned)L.dsafe;
#pragma unroll
    for (int m = 0; m < KM; ++m) in.raw[m] = pb[bv ? L.poff[m] : L.poff[m] - L.bl * c.P];
    in.ip = MP != MPK_MP_DMP ? (a.init_pos + (size_t)b0 * c.D)[io] : 0.0f;
    in.iv = MP == MPK_MP_PRODMP ? (a.init_vel + (size_t)b0 * c.D)[io] : 0.0f;
    in.cp = 0.0; in.cv = 0.0;
    if (ACT) { in.cp = (a.c_pos + (size_t)b0 * c.D)[io]; in.cv = (a.c_vel + (size_t)b0 * c.D)[io]; }
    return in;
}

template <int KM>
__device__ __forceinline__ void finish_group(const LaneMap<KM>& L, const GroupIn<KM>& in, float (&xb)[KM]) {
#pragma unroll
    for (int m = 0; m < KM; ++m) xb[m] = L.isp[m] ? in.raw[m] : (L.isip[m] ? in.ip : (L.isiv[m] ? in.iv : 0.0f));
}

// park the controller constants of every DoF in the wave's 4th staging slot (static kernarg indices: no spill)
__device__ __forceinline__ void park_gains(const ActArgs& act, int lane, int d, float* sSt) {
    double pgd = 0.0, dgd = 0.0, lod = 0.0, hid = 0.0;
#pragma unroll
    for (int dd = 0; dd < kMaxD; ++dd)
        if (dd == d) { pgd = act.pg[dd]; dgd = act.dg[dd]; lod = act.lo[dd]; hid = act.hi[dd]; }
    if (lane < 16) {
        double* sg = reinterpret_cast<double*>(sSt + 3 * kStageStride);
        sg[lane] = pgd; sg[16 + lane] = dgd; sg[32 + lane] = lod; sg[48 + lane] = hid;
    }
    __builtin_amdgcn_wave_barrier();
}

// epilogue of one C tile into the wave-private LDS transpose buffer (rows beyond T land in rows never stored)
template <int MP, int CT>
__device__ __forceinline__ void tile_epilogue(const f32x4& acc0, const f32x4& acc1, const f32x4& acc2,
                                              const float (&dtd)[4], double cp, double cv, const double* sg,
                                              float* sSt, unsigned wofs, int D) {
    double pgd = 0.0, dgd = 0.0, lod = 0.0, hid = 0.0;
    if (CT >= 0 && CT < 3) { pgd = sg[0]; dgd = sg[16]; lod = sg[32]; hid = sg[48]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float p = acc0[r];
        float v;
        if (MP == MPK_MP_PRODMP) v = acc1[r];            // 1/tau is folded into the velocity rows
        else v = (acc1[r] - acc2[r]) * dtd[r];           // forward difference of fp32 positions x (1 / dt)
        float* w = sSt + wofs + r * D;
        w[0] = p;
        w[kStageStride] = v;
        if (CT >= 0 && CT < 3) {
            // float64 without FMA: numpy's promotion in pd_controller.py:21-29 (fp32 desired (+) fp64 state)
            double u;
            if (CT == MPK_CTRL_MOTOR) u = pgd * ((double)p - cp) + dgd * ((double)v - cv);
            else if (CT == MPK_CTRL_POSITION) u = (double)p;
            else u = (double)v;
            u = fmin(fmax(u, lod), hid);
            w[2 * kStageStride] = (float)u;
        }
    }
}

// generic (slow) tile store: partial last row tile whose length is not a multiple of 4, or unaligned outputs.
// Takes plain values (a reference to the kernarg struct would force the whole struct into scratch).
__device__ __noinline__ void store_tile_generic(float* pos, float* vel, float* actions, int nst, int B, int T, int D,
                                                int NTW, const float* sSt, int lane, int b0, int rt, int rows) {
    const int SEG = 16 * D, len = rows * D;      // generic path: never shifted, pitch == SEG
    for (int j = 0; j < nst; ++j) {
        float* outp = j == 0 ? pos : (j == 1 ? vel : actions);
        for (int sb = 0; sb < NTW; ++sb) {
            const int bb = b0 + sb;
            if (bb >= B) continue;
            float* gp = outp + ((size_t)bb * T + rt * 16) * D;
            for (int e = lane; e < len; e += 64) gp[e] = sSt[j * kStageStride + sb * SEG + e];
        }
    }
}

// One coalesced float4 store per output array (every (episode, output) segment of a row tile is contiguous in HBM).
// WT = write-through (sc1) stores: for cache-resident batches the dirty lines then leave the L2 while the kernel is
// still computing instead of in one write-back burst at the kernel boundary (rocprof: 11.5 -> 9.8 us at B = 4096);
// for HBM-streaming batches plain stores are faster (3.5 vs 2.8 TB/s at B = 1M), so k_traj_stream keeps WT = false.
template <bool WT>
__device__ __forceinline__ void store16(float* p, const f32x4& v) {
    if (WT) {
        // hipcc does not count this store: nothing in the kernels waits on stores, the end of the kernel drains them
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else {
        *reinterpret_cast<f32x4*>(p) = v;
    }
}

template <int NST, int KM, bool WT>
__device__ __forceinline__ void tile_store(const TrajArgs& a, const LaneMap<KM>& L, const float* sSt, int lane,
                                           int b0, int rt, int rows) {
    const int D = a.c.D, T = a.c.T, len = rows * D;
    if (a.vec_ok) {
        const int bb = b0 + L.sseg;
        const int lo = (int)ep_shift(a, bb), hi = lo + len, c0 = L.w4;   // valid elements of the padded segment
        if (L.sseg < L.NTW && bb < a.B && c0 < hi && c0 + 4 > lo) {
            const size_t go = ((size_t)bb * T + rt * 16) * D - lo + c0;    // 16-byte aligned by construction
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(sSt + L.rofs);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(sSt + kStageStride + L.rofs);
            f32x4 d2 = d0;
            if (NST > 2) d2 = *reinterpret_cast<const f32x4*>(sSt + 2 * kStageStride + L.rofs);
            if (c0 >= lo && c0 + 4 <= hi) {
                store16<WT>(a.pos + go, d0);
                store16<WT>(a.vel + go, d1);
                if (NST > 2) store16<WT>(a.actions + go, d2);
            } else {                                   // the (at most two) partial chunks of a segment
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c0 + e >= lo && c0 + e < hi) {
                        a.pos[go + e] = d0[e];
                        a.vel[go + e] = d1[e];
                        if (NST > 2) a.actions[go + e] = d2[e];
                    }
                }
            }
        }
    } else {
        store_tile_generic(a.pos, a.vel, a.actions, NST, a.B, T, D, L.NTW, sSt, lane, b0, rt, rows);
    }
}

// ---- tile-major ------------------------------------------------------------------------------------------------
template <int MP, int CT, int KM, bool WT>
__global__ void __launch_bounds__(256) k_traj_tiles(const TrajArgs a, const ActArgs act) {
    __shared__ __attribute__((aligned(16))) float smem[4 * kStageFloats];
    static_assert(MP != MPK_MP_DMP, "dmp runs in k_traj_stream");
    static_assert(CT < 3, "closed-loop rollouts run in k_traj_stream");
    constexpr bool ACT = CT >= 0;
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    constexpr int NST = 2 + (ACT ? 1 : 0);
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPR
    const int KP = 4 * KM, TS = a.TS, D = c.D, T = c.T;
    float* sSt = smem + wave * kStageFloats;
    const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
    const int NRT = (T + 15) >> 4;
    const int wid = blockIdx.x * 4 + wave, Wn = gridDim.x * 4;
    const int rt = wid % NRT;             // Wn % NRT == 0: this wave owns row tile rt for every item
    const int gstride = Wn / NRT;
    int g = wid / NRT;
    if (g >= a.G) return;
    // basis rows of this row tile, MFMA A-fragment layout: lane (t = col, k = 4m + q)
    float af[NOUT][KM];
#pragma unroll
    for (int j = 0; j < NOUT; ++j)
#pragma unroll
        for (int m = 0; m < KM; ++m) af[j][m] = a.A[(size_t)(j * KP + 4 * m + L.q) * TS + rt * 16 + L.col];
    float dtd[4] = {1.f, 1.f, 1.f, 1.f};
    if (MP == MPK_MP_PROMP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dtd[r] = a.aux[rt * 16 + 4 * L.q + r];
    }
    const int rows = min(16, T - rt * 16);
    if (ACT) park_gains(act, lane, L.d, sSt);
    const double* sg = reinterpret_cast<const double*>(sSt + 3 * kStageStride) + (L.dvalid ? L.d : 0);

    float xb[KM];
    GroupIn<KM> cur = load_group<MP, ACT, KM>(a, L, g);
    finish_group<KM>(L, cur, xb);
    double cp = cur.cp, cv = cur.cv;
    while (g < a.G) {
        // 1. issue the NEXT group's loads (consumed at the bottom of this iteration)
        const int gn = g + gstride;
        const GroupIn<KM> nxt = load_group<MP, ACT, KM>(a, L, gn < a.G ? gn : g);
        // 2. matrix cores: C[t, col] = sum_k A[t, k] * X[k, col]
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < KM; ++m) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][m], xb[m], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][m], xb[m], acc1, 0, 0, 0);
            if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[NOUT > 2 ? 2 : 0][m], xb[m], acc2, 0, 0, 0);
        }
        // 3. epilogue -> LDS transpose; 4. coalesced stores
        if (L.dvalid)
            tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, cp, cv, sg, sSt, L.wofs + ep_shift(a, g * L.NTW + L.bl), D);
        __builtin_amdgcn_wave_barrier();
        tile_store<NST, KM, WT>(a, L, sSt, lane, g * L.NTW, rt, rows);
        __builtin_amdgcn_wave_barrier();
        // 5. finish the prefetched fragments for the next iteration
        finish_group<KM>(L, nxt, xb);
        cp = nxt.cp; cv = nxt.cv;
        g = gn;
    }
}

// ---- episode-major ---------------------------------------------------------------------------------------------
// all row tiles of one episode group, in order (shared by the two input-staging variants of k_traj_stream)
template <int MP, int CT, int KM>
__device__ __forceinline__ void stream_group(const TrajArgs& a, const LaneMap<KM>& L, const float* ap,
                                             const float* sAux, const double* sg, float* sSt, int lane, int b0,
                                             const float (&xb)[KM], double cp, double cv, float ey, float ez,
                                             float eg, bool eul, double& qs, double& qds, int nst, bool serial) {
    constexpr bool ACT = CT >= 0;
    constexpr bool CLOSED = CT >= 3;
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : (MP == MPK_MP_PROMP ? 3 : 1);
    constexpr int NST = 2 + (ACT ? 1 : 0);
    const DevCfg& c = a.c;
    const int KP = 4 * KM, TS = a.TS, D = c.D, T = c.T;
    const int NRT = (T + 15) >> 4;
    const unsigned shw = ep_shift(a, b0 + L.bl);          // this column's episode image offset (same for every tile)
    const unsigned wofs = L.wofs + shw;
    const int o0 = L.bl * a.pitch + L.d + (int)shw;       // (row 0, this column) for the serial recurrences
    for (int rt = 0; rt < NRT; ++rt) {
        const int rows = min(16, T - rt * 16);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < KM; ++m) {
            const float* am = ap + (4 * m) * TS + rt * 16;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[0], xb[m], acc0, 0, 0, 0);
            if (NOUT > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[KP * TS], xb[m], acc1, 0, 0, 0);
            if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[2 * KP * TS], xb[m], acc2, 0, 0, 0);
        }
        if (MP != MPK_MP_DMP) {
            float dtd[4] = {1.f, 1.f, 1.f, 1.f};
            if (MP == MPK_MP_PROMP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dtd[r] = sAux[rt * 16 + 4 * L.q + r];
            }
            if (L.dvalid) tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, cp, cv, sg, sSt, wofs, D);
            if (CLOSED) {
                // the step loop of black_box_wrapper.py:175-203 on the reference's torque double integrator
                // (base_reacher_torque.py:25-26), serial in t on the lanes (q == 0); float64, no FMA
                __builtin_amdgcn_wave_barrier();
                if (serial) {
                    const double pgd = sg[0], dgd = sg[16], lod = sg[32], hid = sg[48], dtp = a.plant_dt;
                    float pr[16], vr[16];
#pragma unroll
                    for (int tl = 0; tl < 16; ++tl) { pr[tl] = sSt[o0 + tl * D]; vr[tl] = sSt[kStageStride + o0 + tl * D]; }
#pragma unroll
                    for (int tl = 0; tl < 16; ++tl) {
                        if (tl < rows) {
                            const int t = rt * 16 + tl;
                            double u = 0.0;
                            if (t < nst) {
                                const double dp = (double)pr[tl], dv = (double)vr[tl];
                                if (CT - 3 == MPK_CTRL_MOTOR) u = pgd * (dp - qs) + dgd * (dv - qds);
                                else if (CT - 3 == MPK_CTRL_POSITION) u = dp;
                                else u = dv;
                                u = fmin(fmax(u, lod), hid);
                                qds = qds + dtp * u;
                                qs = qs + dtp * qds;
                            }
                            sSt[2 * kStageStride + o0 + tl * D] = (float)u;
                        }
                    }
                }
            }
        } else {
            // DMP: forcing tile -> LDS, then explicit Euler in scaled time on lanes (q == 0), serial in t;
            // one rounding per op (no FMA), first sample = initial condition
            float* sF = sSt + 2 * kStageStride;
            if (L.dvalid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sF[wofs + r * D] = acc0[r];
            }
            __builtin_amdgcn_wave_barrier();
            if (eul) {
                // the tile's 16 forcing values and scaled-time steps are fetched up front, so the recurrence itself is a
                // pure register chain; the lanes only park z here -- vel = z / tau is applied by ALL lanes below
                float fr[16], dsr[16];
#pragma unroll
                for (int tl = 0; tl < 16; ++tl) { fr[tl] = sF[o0 + tl * D]; dsr[tl] = sAux[rt * 16 + tl]; }
#pragma unroll
                for (int tl = 0; tl < 16; ++tl) {
                    if (tl < rows) {
                        const int t = rt * 16 + tl;
                        sSt[o0 + tl * D] = ey;
                        sSt[kStageStride + o0 + tl * D] = ez;
                        if (t < T - 1) {
                            const float t1 = eg - ey;
                            const float t2 = c.dmp_beta * t1;
                            const float t3 = t2 - ez;
                            const float t4 = c.dmp_alpha * t3;
                            const float acc = t4 + fr[tl];
                            ez = ez + dsr[tl] * acc;
                            ey = ey + dsr[tl] * ez;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // vel = z / tau (IEEE divide, as the reference's tensor op) on every lane that holds staged data
            if (L.dvalid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* w = sSt + kStageStride + wofs + r * D;
                    *w = *w / c.tau;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        tile_store<NST, KM, false>(a, L, sSt, lane, b0, rt, rows);
        __builtin_amdgcn_wave_barrier();
    }
}

constexpr int kChunkGroups = 4;   // episode groups whose inputs one bulk read brings in (BULK variant)

// BULK = false: the raw inputs of the next episode group are gathered per lane straight from HBM (as tile-major).
// BULK = true : a wave owns CHUNKS of kChunkGroups consecutive groups; the chunk's params / init_pos / init_vel
//               (/ c_pos / c_vel) blocks are contiguous in HBM and are read with a handful of coalesced float4 loads
//               one chunk ahead, parked in registers, and committed to a double-buffered wave-private LDS image from
//               which the B fragments are gathered.  Rationale: at HBM-streaming batch sizes the
//               scattered 336-byte parameter reads interleaved with the write stream cost ~35 % of the bandwidth.
template <int MP, int CT, int KM, bool BULK>
__global__ void __launch_bounds__(256) k_traj_stream(const TrajArgs a, const ActArgs act) {
    __shared__ __attribute__((aligned(16))) float smem[4 * kStageFloats];
    extern __shared__ __attribute__((aligned(16))) float sTab[];   // [NOUT][KP][TS] rows + [TS] aux (+ chunk images)
    constexpr bool ACT = CT >= 0 && CT < 3;   // open loop: frozen state (c_pos, c_vel) is an input
    constexpr bool CLOSED = CT >= 3;          // closed loop: plant state (q, qd) is read, integrated and written back
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : (MP == MPK_MP_PROMP ? 3 : 1);
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = 4 * KM, TS = a.TS, D = c.D, B = a.B, P = c.P;
    float* sSt = smem + wave * kStageFloats;
    float* sA = sTab;
    float* sAux = sTab + NOUT * KP * TS;
    {   // stage the shared basis tables through LDS once per workgroup
        const float4* src = reinterpret_cast<const float4*>(a.A);
        float4* dst = reinterpret_cast<float4*>(sA);
        const int n4 = (NOUT * KP * TS) >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) dst[i] = src[i];
        const float4* s2 = reinterpret_cast<const float4*>(a.aux);
        float4* d2 = reinterpret_cast<float4*>(sAux);
        for (int i = threadIdx.x; i < (TS >> 2); i += 256) d2[i] = s2[i];
    }
    __syncthreads();
    const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
    // XCD-contiguous virtual block id (workgroup b runs on XCD b % 8): neighbouring episode groups share an L2
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int wstride = gridDim.x * 4;
    const int w0 = vb * 4 + wave;
    const float* ap = sA + L.q * TS + L.col;
    const double* sg = reinterpret_cast<const double*>(sSt + 3 * kStageStride) + (L.dvalid ? L.d : 0);

    if (!BULK) {
        int g = w0;
        if (g >= a.G) return;
        if (CT >= 0) park_gains(act, lane, L.d, sSt);
        float xb[KM];
        GroupIn<KM> cur = load_group<MP, ACT, KM>(a, L, g);
        finish_group<KM>(L, cur, xb);
        double cp = cur.cp, cv = cur.cv;
        while (g < a.G) {
            const int b0 = g * L.NTW;
            const int gn = g + wstride;
            const GroupIn<KM> nxt = load_group<MP, ACT, KM>(a, L, gn < a.G ? gn : g);
            float ey = 0.f, ez = 0.f, eg = 0.f;
            const bool eul = MP == MPK_MP_DMP && L.dvalid && L.q == 0 && b0 + L.bl < B;
            if (MP == MPK_MP_DMP) {
                if (eul) {
                    const int b = b0 + L.bl;
                    ey = a.init_pos[(size_t)b * D + L.d];
                    ez = a.init_vel[(size_t)b * D + L.d] * c.tau;
                    eg = a.params[(size_t)b * P + c.off + L.d * c.Kloc + c.nb] * c.gs;
                }
            }
            const bool serial = CLOSED && L.dvalid && L.q == 0 && b0 + L.bl < B;
            double qs = 0.0, qds = 0.0;
            int nst = c.T;
            if (CLOSED) {
                if (serial) {
                    const size_t si = (size_t)(b0 + L.bl) * D + L.d;
                    qs = a.q_state[si]; qds = a.qd_state[si];
                    if (a.n_steps) nst = a.n_steps[b0 + L.bl];
                }
            }
            stream_group<MP, CT, KM>(a, L, ap, sAux, sg, sSt, lane, b0, xb, cp, cv, ey, ez, eg, eul, qs, qds, nst, serial);
            if (CLOSED) {
                if (serial) {
                    const size_t si = (size_t)(b0 + L.bl) * D + L.d;
                    a.q_state[si] = qs; a.qd_state[si] = qds;
                }
            }
            finish_group<KM>(L, nxt, xb);
            cp = nxt.cp; cv = nxt.cv;
            g = gn;
        }
    } else {
        constexpr int CH = kChunkGroups;
        const int NTW = L.NTW, EPC = CH * NTW;                 // episodes per chunk
        const int NCH = (B + EPC - 1) / EPC;
        int ch = w0;
        if (ch >= NCH) return;
        if (CT >= 0) park_gains(act, lane, L.d, sSt);
        // chunk image (floats): [params EPC*P | init_pos EPC*D | init_vel EPC*D | c_pos 2*EPC*D | c_vel 2*EPC*D]
        const int offIP = EPC * P, offIV = offIP + EPC * D, offCP = offIV + EPC * D, offCV = offCP + 2 * EPC * D;
        const int img = offCV + 2 * EPC * D;
        float* sImg = sAux + TS + wave * (2 * img);
        const int nP4 = (EPC * P) >> 2, nI4 = (EPC * D) >> 2, nC4 = (EPC * D) >> 1;    // float4 per block
        f32x4 rp0 = {0, 0, 0, 0}, rp1 = rp0, rip = rp0, riv = rp0, rcp = rp0, rcv = rp0;
        auto issue = [&](int chn) {       // coalesced float4 reads of a FULL chunk (ragged chunks are read below)
            const size_t e0 = (size_t)chn * EPC;
            const f32x4* p4 = reinterpret_cast<const f32x4*>(a.params + e0 * P);
            const f32x4* i4 = reinterpret_cast<const f32x4*>(a.init_pos + e0 * D);
            const f32x4* v4 = reinterpret_cast<const f32x4*>(a.init_vel + e0 * D);
            if (lane < nP4) rp0 = p4[lane];
            if (lane + 64 < nP4) rp1 = p4[lane + 64];
            if (lane < nI4) { rip = i4[lane]; riv = v4[lane]; }
            if (ACT) {
                if (lane < nC4) {
                    rcp = reinterpret_cast<const f32x4*>(a.c_pos + e0 * D)[lane];
                    rcv = reinterpret_cast<const f32x4*>(a.c_vel + e0 * D)[lane];
                }
            }
        };
        auto commit = [&](float* buf) {
            f32x4* b4 = reinterpret_cast<f32x4*>(buf);
            if (lane < nP4) b4[lane] = rp0;
            if (lane + 64 < nP4) b4[lane + 64] = rp1;
            if (lane < nI4) { b4[(offIP >> 2) + lane] = rip; b4[(offIV >> 2) + lane] = riv; }
            if (ACT) {
                if (lane < nC4) { b4[(offCP >> 2) + lane] = rcp; b4[(offCV >> 2) + lane] = rcv; }
            }
        };
        auto read_ragged = [&](int chn, float* buf) {   // last, incomplete chunk: element-wise, bounds-checked
            const size_t e0 = (size_t)chn * EPC;
            const int ne = B - (int)e0;
            for (int e = lane; e < ne * P; e += 64) buf[e] = a.params[e0 * P + e];
            for (int e = lane; e < ne * D; e += 64) {
                buf[offIP + e] = a.init_pos[e0 * D + e];
                buf[offIV + e] = a.init_vel[e0 * D + e];
                if (ACT) {
                    reinterpret_cast<double*>(buf + offCP)[e] = a.c_pos[e0 * D + e];
                    reinterpret_cast<double*>(buf + offCV)[e] = a.c_vel[e0 * D + e];
                }
            }
        };
        auto full = [&](int chn) { return (chn + 1) * EPC <= B; };
        int cur = 0;
        if (full(ch)) { issue(ch); commit(sImg); } else read_ragged(ch, sImg);
        __builtin_amdgcn_wave_barrier();
        while (ch < NCH) {
            const int chn = ch + wstride;
            const bool have_next = chn < NCH, next_full = have_next && full(chn);
            if (next_full) issue(chn);                      // in flight under this chunk's CH groups
            const float* buf = sImg + cur * img;
            for (int j = 0; j < CH; ++j) {
                const int g = ch * CH + j;
                if (g >= a.G) break;
                const int b0 = g * NTW;
                const float* pj = buf + j * NTW * P;
                const unsigned io = (unsigned)(j * NTW * D) + L.ioff;
                float xb[KM];
                const float ip = buf[offIP + io], iv = buf[offIV + io];
#pragma unroll
                for (int m = 0; m < KM; ++m) {
                    const float raw = pj[L.poff[m]];
                    xb[m] = L.isp[m] ? raw : (L.isip[m] ? ip : (L.isiv[m] ? iv : 0.0f));
                }
                double cp = 0.0, cv = 0.0;
                if (ACT) {
                    cp = reinterpret_cast<const double*>(buf + offCP)[io];
                    cv = reinterpret_cast<const double*>(buf + offCV)[io];
                }
                float ey = 0.f, ez = 0.f, eg = 0.f;
                const bool eul = MP == MPK_MP_DMP && L.dvalid && L.q == 0 && b0 + L.bl < B;
                if (MP == MPK_MP_DMP) {
                    if (eul) {
                        ey = ip;
                        ez = iv * c.tau;
                        eg = pj[L.bl * P + c.off + L.d * c.Kloc + c.nb] * c.gs;
                    }
                }
                const bool serial = CLOSED && L.dvalid && L.q == 0 && b0 + L.bl < B;
                double qs = 0.0, qds = 0.0;
                int nst = c.T;
                if (CLOSED) {
                    if (serial) {
                        const size_t si = (size_t)(b0 + L.bl) * D + L.d;
                        qs = a.q_state[si]; qds = a.qd_state[si];
                        if (a.n_steps) nst = a.n_steps[b0 + L.bl];
                    }
                }
                stream_group<MP, CT, KM>(a, L, ap, sAux, sg, sSt, lane, b0, xb, cp, cv, ey, ez, eg, eul, qs, qds, nst,
                                         serial);
                if (CLOSED) {
                    if (serial) {
                        const size_t si = (size_t)(b0 + L.bl) * D + L.d;
                        a.q_state[si] = qs; a.qd_state[si] = qds;
                    }
                }
            }
            if (have_next) {
                float* nb = sImg + (cur ^ 1) * img;
                if (next_full) commit(nb); else read_ragged(chn, nb);
                __builtin_amdgcn_wave_barrier();
            }
            cur ^= 1;
            ch = chn;
        }
    }
}

// ---- episode-major, four groups per wave: the serial-recurrence variants ---------------------------------------------
// DMP (Euler recurrence) and the closed-loop rollout (controller + plant recurrence) are serial in t and run on the
// 16 lanes that hold row 0 of a column.  Here a wave owns FOUR consecutive episode groups at once: per row tile it
// produces the four C tiles back to back on the matrix cores, then lane quarter q runs group q's recurrence, so the four
// recurrences advance in parallel (4x fewer serial instructions per episode), then the four tiles leave as coalesced
// float4 stores.  Same arithmetic and bits as k_traj_stream.
constexpr int kQuad = 4;

template <int MP, int CT, int KM>
__global__ void __launch_bounds__(256) k_traj_quad(const TrajArgs a, const ActArgs act) {
    __shared__ __attribute__((aligned(16))) float smem[4 * kQuad * 3 * kStageStride];   // per wave: 4 x (pos|vel|act or force)
    __shared__ double sgain[4][64];
    extern __shared__ __attribute__((aligned(16))) float sTab[];   // [NOUT][KP][TS] rows + [TS] aux
    constexpr bool CLOSED = CT >= 3;
    static_assert(MP == MPK_MP_DMP || CLOSED, "k_traj_quad is for the serial-recurrence variants");
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : (MP == MPK_MP_PROMP ? 3 : 1);
    constexpr int NST = CLOSED ? 3 : 2;
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = 4 * KM, TS = a.TS, D = c.D, B = a.B, P = c.P, T = c.T;
    float* sW = smem + wave * (kQuad * 3 * kStageStride);
    float* sA = sTab;
    float* sAux = sTab + NOUT * KP * TS;
    {
        const float4* src = reinterpret_cast<const float4*>(a.A);
        float4* dst = reinterpret_cast<float4*>(sA);
        const int n4 = (NOUT * KP * TS) >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) dst[i] = src[i];
        const float4* s2 = reinterpret_cast<const float4*>(a.aux);
        float4* d2 = reinterpret_cast<float4*>(sAux);
        for (int i = threadIdx.x; i < (TS >> 2); i += 256) d2[i] = s2[i];
    }
    __syncthreads();
    const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
    const int NTW = L.NTW, NRT = (T + 15) >> 4;
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int ustride = gridDim.x * 4;
    const int NU = (a.G + kQuad - 1) / kQuad;
    int u = vb * 4 + wave;
    if (u >= NU) return;
    const float* ap = sA + L.q * TS + L.col;
    double pgd = 0.0, dgd = 0.0, lod = 0.0, hid = 0.0;
    if (CLOSED) {
#pragma unroll
        for (int dd = 0; dd < kMaxD; ++dd)
            if (dd == L.d) { pgd = act.pg[dd]; dgd = act.dg[dd]; lod = act.lo[dd]; hid = act.hi[dd]; }
    }
    (void)sgain;

    float xb[kQuad][KM];
    GroupIn<KM> nx[kQuad];
#pragma unroll
    for (int j = 0; j < kQuad; ++j) {
        const int g = u * kQuad + j;
        nx[j] = load_group<MP, false, KM>(a, L, g < a.G ? g : a.G - 1);
    }
    while (u < NU) {
        const int g0 = u * kQuad;
#pragma unroll
        for (int j = 0; j < kQuad; ++j) finish_group<KM>(L, nx[j], xb[j]);
        const int un = u + ustride;
        if (un < NU) {
#pragma unroll
            for (int j = 0; j < kQuad; ++j) {
                const int g = un * kQuad + j;
                nx[j] = load_group<MP, false, KM>(a, L, g < a.G ? g : a.G - 1);
            }
        }
        // this lane's recurrence: group g0 + q, column (bl, d)
        const int gq = g0 + L.q, bq = gq * NTW + L.bl;
        const bool serial = L.dvalid && gq < a.G && bq < B;
        const int oq = L.bl * a.pitch + L.d + (int)ep_shift(a, bq);      // (row 0, this column) in group q's image
        float* sQ = sW + L.q * (3 * kStageStride);
        double qs = 0.0, qds = 0.0;
        int nst = T;
        float ey = 0.f, ez = 0.f, eg = 0.f;
        if (serial) {
            const size_t si = (size_t)bq * D + L.d;
            if (CLOSED) {
                qs = a.q_state[si]; qds = a.qd_state[si];
                if (a.n_steps) nst = a.n_steps[bq];
            } else {
                ey = a.init_pos[si];
                ez = a.init_vel[si] * c.tau;
                eg = a.params[(size_t)bq * P + c.off + L.d * c.Kloc + c.nb] * c.gs;
            }
        }
        for (int rt = 0; rt < NRT; ++rt) {
            const int rows = min(16, T - rt * 16);
            // 1. four C tiles on the matrix cores -> four staging images
#pragma unroll
            for (int j = 0; j < kQuad; ++j) {
                if (g0 + j < a.G) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < KM; ++m) {
                        const float* am = ap + (4 * m) * TS + rt * 16;
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[0], xb[j][m], acc0, 0, 0, 0);
                        if (NOUT > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[KP * TS], xb[j][m], acc1, 0, 0, 0);
                        if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[2 * KP * TS], xb[j][m], acc2, 0, 0, 0);
                    }
                    float* sJ = sW + j * (3 * kStageStride);
                    const unsigned wofs = L.wofs + ep_shift(a, (g0 + j) * NTW + L.bl);
                    if (L.dvalid) {
                        if (MP == MPK_MP_DMP) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) sJ[2 * kStageStride + wofs + r * D] = acc0[r];
                        } else {
                            float dtd[4] = {1.f, 1.f, 1.f, 1.f};
                            if (MP == MPK_MP_PROMP) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) dtd[r] = sAux[rt * 16 + 4 * L.q + r];
                            }
                            tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, 0.0, 0.0, nullptr, sJ, wofs, D);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // 2. four recurrences in parallel, one per lane quarter (float64 / fp32 without FMA, as k_traj_stream)
            if (serial) {
                if (CLOSED) {
                    float pr[16], vr[16];
#pragma unroll
                    for (int tl = 0; tl < 16; ++tl) { pr[tl] = sQ[oq + tl * D]; vr[tl] = sQ[kStageStride + oq + tl * D]; }
#pragma unroll
                    for (int tl = 0; tl < 16; ++tl) {
                        if (tl < rows) {
                            const int t = rt * 16 + tl;
                            double uu = 0.0;
                            if (t < nst) {
                                const double dp = (double)pr[tl], dv = (double)vr[tl];
                                if (CT - 3 == MPK_CTRL_MOTOR) uu = pgd * (dp - qs) + dgd * (dv - qds);
                                else if (CT - 3 == MPK_CTRL_POSITION) uu = dp;
                                else uu = dv;
                                uu = fmin(fmax(uu, lod), hid);
                                qds = qds + a.plant_dt * uu;
                                qs = qs + a.plant_dt * qds;
                            }
                            sQ[2 * kStageStride + oq + tl * D] = (float)uu;
                        }
                    }
                } else {
                    float fr[16], dsr[16];
#pragma unroll
                    for (int tl = 0; tl < 16; ++tl) { fr[tl] = sQ[2 * kStageStride + oq + tl * D]; dsr[tl] = sAux[rt * 16 + tl]; }
#pragma unroll
                    for (int tl = 0; tl < 16; ++tl) {
                        if (tl < rows) {
                            const int t = rt * 16 + tl;
                            sQ[oq + tl * D] = ey;
                            sQ[kStageStride + oq + tl * D] = ez;
                            if (t < T - 1) {
                                const float t1 = eg - ey;
                                const float t2 = c.dmp_beta * t1;
                                const float t3 = t2 - ez;
                                const float t4 = c.dmp_alpha * t3;
                                const float acc = t4 + fr[tl];
                                ez = ez + dsr[tl] * acc;
                                ey = ey + dsr[tl] * ez;
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (MP == MPK_MP_DMP) {
                // vel = z / tau (IEEE divide) on every lane, for the four images
                if (L.dvalid) {
#pragma unroll
                    for (int j = 0; j < kQuad; ++j) {
                        float* sJ = sW + j * (3 * kStageStride) + kStageStride + L.wofs +
                                    ep_shift(a, (g0 + j) * NTW + L.bl);
#pragma unroll
                        for (int r = 0; r < 4; ++r) sJ[r * D] = sJ[r * D] / c.tau;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            // 3. coalesced stores of the four tiles
#pragma unroll
            for (int j = 0; j < kQuad; ++j)
                if (g0 + j < a.G)
                    tile_store<NST, KM, false>(a, L, sW + j * (3 * kStageStride), lane, (g0 + j) * NTW, rt, rows);
            __builtin_amdgcn_wave_barrier();
        }
        if (CLOSED) {
            if (serial) {
                const size_t si = (size_t)bq * D + L.d;
                a.q_state[si] = qs; a.qd_state[si] = qds;
            }
        }
        u = un;
    }
}

template <int MP, int CT>
static int launch_traj_t(const TrajArgs& ta, const ActArgs& aa, bool stream_mode, bool write_through, bool bulk,
                         bool quad, int blocks, size_t lds, void* stream) {
    const dim3 g(blocks), b(256);
    hipStream_t s = (hipStream_t)stream;
    const int km = ta.c.KP / 4;
    if (stream_mode && quad) {
        if constexpr (MP == MPK_MP_DMP || CT >= 3) {
            switch (km) {
                case 1: hipLaunchKernelGGL((k_traj_quad<MP, CT, 1>), g, b, lds, s, ta, aa); break;
                case 2: hipLaunchKernelGGL((k_traj_quad<MP, CT, 2>), g, b, lds, s, ta, aa); break;
                case 3: hipLaunchKernelGGL((k_traj_quad<MP, CT, 3>), g, b, lds, s, ta, aa); break;
                default: hipLaunchKernelGGL((k_traj_quad<MP, CT, 4>), g, b, lds, s, ta, aa); break;
            }
        }
    } else if (stream_mode) {
        if (bulk) {
            switch (km) {
                case 1: hipLaunchKernelGGL((k_traj_stream<MP, CT, 1, true>), g, b, lds, s, ta, aa); break;
                case 2: hipLaunchKernelGGL((k_traj_stream<MP, CT, 2, true>), g, b, lds, s, ta, aa); break;
                case 3: hipLaunchKernelGGL((k_traj_stream<MP, CT, 3, true>), g, b, lds, s, ta, aa); break;
                default: hipLaunchKernelGGL((k_traj_stream<MP, CT, 4, true>), g, b, lds, s, ta, aa); break;
            }
        } else {
            switch (km) {
                case 1: hipLaunchKernelGGL((k_traj_stream<MP, CT, 1, false>), g, b, lds, s, ta, aa); break;
                case 2: hipLaunchKernelGGL((k_traj_stream<MP, CT, 2, false>), g, b, lds, s, ta, aa); break;
                case 3: hipLaunchKernelGGL((k_traj_stream<MP, CT, 3, false>), g, b, lds, s, ta, aa); break;
                default: hipLaunchKernelGGL((k_traj_stream<MP, CT, 4, false>), g, b, lds, s, ta, aa); break;
            }
        }
    } else {
        if constexpr (MP != MPK_MP_DMP && CT < 3) {
            if (write_through) {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 1, true>), g, b, 0, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 2, true>), g, b, 0, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 3, true>), g, b, 0, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 4, true>), g, b, 0, s, ta, aa); break;
                }
            } else {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 1, false>), g, b, 0, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 2, false>), g, b, 0, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 3, false>), g, b, 0, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 4, false>), g, b, 0, s, ta, aa); break;
                }
            }
        }
    }
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}

template <int MP>
static int launch_traj_ct(const TrajArgs& ta, const ActArgs& aa, int ct, bool stream_mode, bool write_through,
                          bool bulk, bool quad, int blocks, size_t lds, void* stream) {
    if constexpr (MP != MPK_MP_DMP) {
        switch (ct) {
            case MPK_CTRL_MOTOR: return launch_traj_t<MP, MPK_CTRL_MOTOR>(ta, aa, stream_mode, write_through, bulk, quad, blocks, lds, stream);
            case MPK_CTRL_VELOCITY: return launch_traj_t<MP, MPK_CTRL_VELOCITY>(ta, aa, stream_mode, write_through, bulk, quad, blocks, lds, stream);
            case MPK_CTRL_POSITION: return launch_traj_t<MP, MPK_CTRL_POSITION>(ta, aa, stream_mode, write_through, bulk, quad, blocks, lds, stream);
            case 3 + MPK_CTRL_MOTOR: return launch_traj_t<MP, 3 + MPK_CTRL_MOTOR>(ta, aa, true, false, bulk, quad, blocks, lds, stream);
            case 3 + MPK_CTRL_VELOCITY: return launch_traj_t<MP, 3 + MPK_CTRL_VELOCITY>(ta, aa, true, false, bulk, quad, blocks, lds, stream);
            case 3 + MPK_CTRL_POSITION: return launch_traj_t<MP, 3 + MPK_CTRL_POSITION>(ta, aa, true, false, bulk, quad, blocks, lds, stream);
            default: break;
        }
    }
    return launch_traj_t<MP, -1>(ta, aa, stream_mode, write_through, bulk, quad, blocks, lds, stream);
}

// 0 = automatic, 1 = force tile-major, 2 = force episode-major (MPK_MAPPING environment variable, for A/B runs)
static int mapping_override() {
    const char* e = getenv("MPK_MAPPING");   // read per launch: tests flip it at run time
    const int v = e ? atoi(e) : 0;
    return (v < 0 || v > 2) ? 0 : v;
}

int launch_traj_shared(const DevCfg& c, const SharedTables& st, const float* params, const float* init_pos,
                       const float* init_vel, float* pos, float* vel, float* actions, const RolloutDev* rc,
                       const double* c_pos, const double* c_vel, double* q_state, double* qd_state,
                       const int32_t* n_steps, int B, int num_cu, void* stream, const char** kernel_name) {
    TrajArgs ta;
    const bool closed = q_state != nullptr;
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
    ta.shifted = (TD % 4 != 0 && NTW * (seg4 + 1) <= 64 && NTW * (SEG + 4) <= kStageStride) ? 1 : 0;
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
    const int NRT = (c.T + 15) / 16;
    const long max_waves = (long)num_cu * 32;     // 8 waves per SIMD resident
    // work decomposition: episode-major once the outputs stop being cache resident (or when it is the only option)
    const size_t table_bytes = ((size_t)st.n_out * c.KP * st.TS + st.TS) * sizeof(float);
    const double out_bytes = (double)B * c.T * c.D * 4.0 * nst;
    bool stream_mode = c.mp_type == MPK_MP_DMP || closed || out_bytes > 96.0 * 1024 * 1024;
    const int ov = mapping_override();
    if (c.mp_type != MPK_MP_DMP && !closed && ov == 1) stream_mode = false;
    if (ov == 2) stream_mode = true;
    if (stream_mode && table_bytes + 4 * kStageFloats * sizeof(float) > 64 * 1024) {
        if (c.mp_type == MPK_MP_DMP || closed) { set_error("trajectory too long for the episode-major kernel's LDS budget"); return MPK_EINVAL; }
        stream_mode = false;
    }
    // write-through stores for the cache-resident tile-major case (MPK_WRITE_THROUGH=0/1 overrides, for A/B runs)
    bool write_through = !stream_mode;
    if (const char* e = getenv("MPK_WRITE_THROUGH")) write_through = atoi(e) != 0 && !stream_mode;
    int blocks;
    size_t lds = 0;
    bool bulk = false;
    // serial-recurrence variants (DMP, closed loop): four groups per wave, recurrences in parallel on the lane quarters
    // (MPK_QUAD=0 falls back to k_traj_stream, for A/B runs); needs its 52 KB of staging + the tables within 64 KB
    bool quad = stream_mode && (c.mp_type == MPK_MP_DMP || closed) &&
                table_bytes + (4 * kQuad * 3 * kStageStride) * sizeof(float) + 4 * 64 * sizeof(double) <= 64 * 1024;
    {   // automatic: only when the 4x coarser work units still give every CU a few waves (MPK_QUAD: 0 off, 2 force)
        int quad_mode = 1;
        if (const char* e = getenv("MPK_QUAD")) quad_mode = atoi(e);
        const long units = (ta.G + kQuad - 1) / kQuad;
        quad = quad && quad_mode != 0 && (quad_mode == 2 || units >= (long)num_cu * 4);
    }
    if (quad) {
        lds = table_bytes;
        const long units = (ta.G + kQuad - 1) / kQuad;
        const long waves = units < max_waves ? units : max_waves;
        blocks = (int)((waves + 3) / 4);
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
    } else if (stream_mode) {
        lds = table_bytes;
        // bulk input staging: chunk blocks must be float4-sized / aligned and fit the per-lane register image
        const int EPC = kChunkGroups * NTW;
        const size_t img_floats = (size_t)EPC * (c.P + 2 * c.D + 4 * c.D);
        const size_t lds_bulk = table_bytes + 4 * 2 * img_floats * sizeof(float);
        bulk = (EPC * c.P) % 4 == 0 && (EPC * c.D) % 4 == 0 && (EPC * c.P) / 4 <= 128 && (EPC * c.D) / 2 <= 64 &&
               aligned16(params) && aligned16(init_pos) && aligned16(init_vel) &&
               (!act || closed || (aligned16(c_pos) && aligned16(c_vel))) &&
               lds_bulk + 4 * kStageFloats * sizeof(float) <= 64 * 1024;
        // MPK_BULK=0 disables, =2 forces it below the size threshold too (tests); default: HBM-streaming sizes only
        int bulk_mode = 1;
        if (const char* e = getenv("MPK_BULK")) bulk_mode = atoi(e);
        // automatic: only when the outputs stream to HBM AND the 4x coarser work units still fill the chip; the
        // latency-bound DMP recurrence prefers occupancy over input staging
        const long chunks = (ta.G + kChunkGroups - 1) / kChunkGroups;
        const bool auto_ok = out_bytes > 96.0 * 1024 * 1024 && chunks >= max_waves / 2 && c.mp_type != MPK_MP_DMP;
        bulk = bulk && bulk_mode != 0 && (bulk_mode == 2 || auto_ok);
        long units = ta.G;
        if (bulk) { lds = lds_bulk; units = (ta.G + kChunkGroups - 1) / kChunkGroups; }
        const long waves = units < max_waves ? units : max_waves;
        blocks = (int)((waves + 3) / 4);
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;                  // XCD-contiguous remap needs a multiple of 8
    } else {
        const long items = (long)ta.G * NRT;
        long ipw = (items + max_waves - 1) / max_waves;                  // items per wave, balanced
        if (const char* e = getenv("MPK_IPW")) { const long v = atol(e); if (v > 0) ipw = v; }   // A/B runs
        const long waves = (items + ipw - 1) / ipw;
        blocks = (int)((waves + 3) / 4);
        blocks = (blocks + NRT - 1) / NRT * NRT;                         // #waves % NRT == 0
    }
    if (blocks < 1) blocks = 1;
    switch (c.mp_type) {
        case MPK_MP_PRODMP:
            *kernel_name = closed ? (quad ? "k_traj_quad<prodmp,closed>" : "k_traj_stream<prodmp,closed>") : stream_mode ? (act ? "k_traj_stream<prodmp,act>" : "k_traj_stream<prodmp>")
                                       : (act ? "k_traj_tiles<prodmp,act>" : "k_traj_tiles<prodmp>");
            return launch_traj_ct<MPK_MP_PRODMP>(ta, aa, ct, stream_mode, write_through, bulk, quad, blocks, lds, stream);
        case MPK_MP_PROMP:
            *kernel_name = closed ? (quad ? "k_traj_quad<promp,closed>" : "k_traj_stream<promp,closed>") : stream_mode ? (act ? "k_traj_stream<promp,act>" : "k_traj_stream<promp>")
                                       : (act ? "k_traj_tiles<promp,act>" : "k_traj_tiles<promp>");
            return launch_traj_ct<MPK_MP_PROMP>(ta, aa, ct, stream_mode, write_through, bulk, quad, blocks, lds, stream);
        default:
            *kernel_name = quad ? "k_traj_quad<dmp>" : "k_traj_stream<dmp>";
            return launch_traj_ct<MPK_MP_DMP>(ta, aa, -1, true, false, bulk, quad, blocks, lds, stream);
    }
}

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
                } else {
                    v = a.init_vel[(size_t)b * D + dd];
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
                    prodmp_col(c, bc, idx, xi, k, (double)tau, &h, &hv);
                    sH[t * KT + k] = h;
                    sH[(T + t) * KT + k] = hv;
                }
            }
        } else {
            for (int t = tid; t < T; t += nt) {
                const float time = c.base_times[t] + it;
                const double x = phase_f64(c, time, tau, delay, nullptr);
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
                for (int t = 0; t < T; ++t) {
                    const float f = sP[t * D + dd];
                    sP[t * D + dd] = y;
                    sV[t * D + dd] = z / tau;
                    if (t < T - 1) {
                        const float ds = sT[t];
                        const float t1 = g - y;
                        const float t2 = c.dmp_beta * t1;
                        const float t3 = t2 - z;
                        const float t4 = c.dmp_alpha * t3;
                        const float acc = t4 + f;
                        z = z + ds * acc;
                        y = y + ds * z;
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

int launch_traj_rows(const DevCfg& c, const float* params, const float* init_pos, const float* init_vel,
                     const float* init_time, float init_time_shared, float* pos, float* vel, int32_t* range_flag,
                     int B, int num_cu, void* stream, const char** kernel_name) {
    if (c.mp_type == MPK_MP_PROMP && c.T < 2) {
        set_error("promp needs at least two time steps for the finite-difference velocity");
        return MPK_EINVAL;
    }
    const int nrow = c.mp_type == MPK_MP_PRODMP ? 2 : 1;
    const size_t floats = (size_t)c.D * c.KT + (size_t)nrow * c.T * c.KT + (size_t)c.T * c.D +
                          (c.mp_type == MPK_MP_DMP ? (size_t)c.T * c.D : 0) + c.T + 8;
    const size_t lds = floats * sizeof(float);
    if (lds > 160 * 1024) { set_error("trajectory too large for the per-episode kernel's LDS budget"); return MPK_EINVAL; }
    RowArgs ra{c, params, init_pos, init_vel, init_time, init_time_shared, pos, vel, range_flag, B};
    int blocks = B < num_cu * 8 ? B : num_cu * 8;
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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

// ------------------------------------------------------------------------------------------------------------
// k_pd_rollout: controller + plant loop, one lane per (episode, DoF), float64, no FMA contraction
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pd_rollout(const RolloutDev rc, const int D, const float* __restrict__ des_pos,
                                                    const float* __restrict__ des_vel, double* __restrict__ Q,
                                                    double* __restrict__ QD, const int32_t* __restrict__ n_steps,
                                                    float* __restrict__ actions, const int B, const int T) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)B * D) return;
    const int b = (int)(e / D), d = (int)(e - (long)b * D);
    double q = Q[e], qd = QD[e];
    int n = n_steps ? n_steps[b] : T;
    n = n < T ? n : T;
    const double pg = rc.pg[d], dg = rc.dg[d], lo = rc.lo[d], hi = rc.hi[d], dt = rc.dt;
    const size_t base = (size_t)b * T * D + d;
    for (int t = 0; t < T; ++t) {
        double u = 0.0;
        if (t < n) {
            const double dp = (double)des_pos[base + (size_t)t * D], dv = (double)des_vel[base + (size_t)t * D];
            if (rc.controller_type == MPK_CTRL_MOTOR) u = pg * (dp - q) + dg * (dv - qd);
            else if (rc.controller_type == MPK_CTRL_POSITION) u = dp;
            else u = dv;
            u = fmin(fmax(u, lo), hi);
            if (rc.plant_type == MPK_PLANT_DOUBLE_INTEGRATOR) {
                qd = qd + dt * u;
                q = q + dt * qd;
            }
        }
        if (actions) actions[base + (size_t)t * D] = (float)u;
    }
    Q[e] = q;
    QD[e] = qd;
}

// Tile-streaming variant (D <= 16, float4-aligned trajectories): a wave owns a group of 16/DP episodes and walks their
// 16-step row tiles in order -- coalesced float4 loads of the desired (pos, vel) pieces one tile ahead, wave-private
// LDS image, the serial controller + plant recurrence on the lanes (q == 0) as a register chain (float64, no FMA),
// coalesced float4 store of the actions.  Same arithmetic, same bits as k_pd_rollout.
struct PdArgs {
    RolloutDev rc;
    const float* des_pos;
    const float* des_vel;
    double* Q;
    double* QD;
    const int32_t* n_steps;
    float* actions;
    int D, sh, B, T, G;
    unsigned inv_seg4;
};

__global__ void __launch_bounds__(256) k_pd_rollout_tiles(const PdArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 3 * kStageStride];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sSt = smem + wave * (3 * kStageStride);      // desired pos | desired vel | actions
    const int D = a.D, T = a.T, B = a.B, SEG = 16 * D, DP = 1 << a.sh, NTW = 16 >> a.sh;
    const int col = lane & 15, bl = col >> a.sh, d = col & (DP - 1);
    const bool lane_serial = lane < 16 && d < D;
    const int seg4 = SEG >> 2;
    const int sseg = (int)(((unsigned)lane * a.inv_seg4) >> 16);
    const int w4 = (lane - sseg * seg4) * 4;
    const unsigned rofs = (unsigned)(sseg * SEG + w4);
    const size_t gofs = (size_t)sseg * T * D + w4;
    const int NRT = (T + 15) >> 4;
    double pgd = 0.0, dgd = 0.0, lod = 0.0, hid = 0.0;
#pragma unroll
    for (int dd = 0; dd < kMaxD; ++dd)
        if (dd == d) { pgd = a.rc.pg[dd]; dgd = a.rc.dg[dd]; lod = a.rc.lo[dd]; hid = a.rc.hi[dd]; }
    const double dtp = a.rc.dt;
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    for (int g = vb * 4 + wave; g < a.G; g += gridDim.x * 4) {
        const int b0 = g * NTW;
        const bool serial = lane_serial && b0 + bl < B;
        const bool mover = sseg < NTW && b0 + sseg < B;
        double qs = 0.0, qds = 0.0;
        int nst = T;
        if (serial) {
            const size_t si = (size_t)(b0 + bl) * D + d;
            qs = a.Q[si]; qds = a.QD[si];
            if (a.n_steps) nst = min(a.n_steps[b0 + bl], T);
        }
        const float* gp = a.des_pos + (size_t)b0 * T * D + gofs;
        const float* gv = a.des_vel + (size_t)b0 * T * D + gofs;
        f32x4 lp = {0, 0, 0, 0}, lv = {0, 0, 0, 0};
        if (mover && w4 < min(16, T) * D) { lp = *reinterpret_cast<const f32x4*>(gp); lv = *reinterpret_cast<const f32x4*>(gv); }
        for (int rt = 0; rt < NRT; ++rt) {
            const int rows = min(16, T - rt * 16);
            const bool mine = mover && w4 < rows * D;
            if (mine) {
                *reinterpret_cast<f32x4*>(sSt + rofs) = lp;
                *reinterpret_cast<f32x4*>(sSt + kStageStride + rofs) = lv;
            }
            if (rt + 1 < NRT) {   // next tile's pieces travel under this tile's recurrence
                const int rows_n = min(16, T - (rt + 1) * 16);
                if (mover && w4 < rows_n * D) {
                    lp = *reinterpret_cast<const f32x4*>(gp + (size_t)(rt + 1) * SEG);
                    lv = *reinterpret_cast<const f32x4*>(gv + (size_t)(rt + 1) * SEG);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (serial) {
                const int o0 = bl * SEG + d;
                float pr[16], vr[16];
#pragma unroll
                for (int tl = 0; tl < 16; ++tl) { pr[tl] = sSt[o0 + tl * D]; vr[tl] = sSt[kStageStride + o0 + tl * D]; }
#pragma unroll
                for (int tl = 0; tl < 16; ++tl) {
                    if (tl < rows) {
                        const int t = rt * 16 + tl;
                        double u = 0.0;
                        if (t < nst) {
                            const double dp = (double)pr[tl], dv = (double)vr[tl];
                            if (a.rc.controller_type == MPK_CTRL_MOTOR) u = pgd * (dp - qs) + dgd * (dv - qds);
                            else if (a.rc.controller_type == MPK_CTRL_POSITION) u = dp;
                            else u = dv;
                            u = fmin(fmax(u, lod), hid);
                            if (a.rc.plant_type == MPK_PLANT_DOUBLE_INTEGRATOR) {
                                qds = qds + dtp * u;
                                qs = qs + dtp * qds;
                            }
                        }
                        sSt[2 * kStageStride + o0 + tl * D] = (float)u;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (a.actions && mine)
                *reinterpret_cast<f32x4*>(a.actions + (size_t)b0 * T * D + gofs + (size_t)rt * SEG) =
                    *reinterpret_cast<const f32x4*>(sSt + 2 * kStageStride + rofs);
            __builtin_amdgcn_wave_barrier();
        }
        if (serial) {
            const size_t si = (size_t)(b0 + bl) * D + d;
            a.Q[si] = qs; a.QD[si] = qds;
        }
    }
}

int launch_pd_rollout(const RolloutDev& rc, int D, const float* des_pos, const float* des_vel, double* q, double* qd,
                      const int32_t* n_steps, float* actions, int B, int T, void* stream) {
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int last_rows = T - (T - 1) / 16 * 16;
    const bool tiles_ok = D >= 1 && D <= kMaxD && (T * D) % 4 == 0 && (last_rows * D) % 4 == 0 && aligned16(des_pos) &&
                          aligned16(des_vel) && (!actions || aligned16(actions)) && !getenv("MPK_PD_SIMPLE");
    if (tiles_ok) {
        PdArgs pa;
        pa.rc = rc; pa.des_pos = des_pos; pa.des_vel = des_vel; pa.Q = q; pa.QD = qd; pa.n_steps = n_steps;
        pa.actions = actions; pa.D = D; pa.B = B; pa.T = T;
        int sh = 0;
        while ((1 << sh) < D) ++sh;
        pa.sh = sh;
        const int NTW = 16 >> sh;
        pa.G = (B + NTW - 1) / NTW;
        pa.inv_seg4 = 65536u / (unsigned)(4 * D) + 1u;
        int blocks = (pa.G + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
        hipLaunchKernelGGL(k_pd_rollout_tiles, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pa);
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    }
    const long n = (long)B * D;
    const int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(k_pd_rollout, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rc, D, des_pos, des_vel, q, qd,
                       n_steps, actions, B, T);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// integer replanning state
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_replan_advance(int32_t* __restrict__ traj_steps, int32_t* __restrict__ plan_steps,
                                                        int32_t* __restrict__ seg_len, uint8_t* __restrict__ done,
                                                        const int every, const int max_planning_times,
                                                        const int horizon, const int T, const int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    if (done[b]) { seg_len[b] = 0; return; }
    const int cur = traj_steps[b];
    const int plan = plan_steps[b] + 1;
    // first global step g = cur + t + 1 (t >= 0) at which the loop breaks
    int g_break = horizon;
    if (plan < max_planning_times) {
        const int gm = (cur / every + 1) * every;  // next multiple of `every` strictly above cur
        g_break = gm < horizon ? gm : horizon;
    }
    int seg = g_break - cur;
    if (seg > T) seg = T;
    if (seg < 1) seg = 1;
    plan_steps[b] = plan;
    seg_len[b] = seg;
    traj_steps[b] = cur + seg;
    done[b] = (cur + seg) >= horizon ? 1 : 0;
}

int launch_replan_advance(int32_t* traj_steps, int32_t* plan_steps, int32_t* seg_len, uint8_t* done, int every,
                          int max_planning_times, int horizon, int T, int B, void* stream) {
    hipLaunchKernelGGL(k_replan_advance, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, traj_steps,
                       plan_steps, seg_len, done, every, max_planning_times, horizon, T, B);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// validity reduction: one wave per episode
// ------------------------------------------------------------------------------------------------------------
struct ValidArgs {
    double lo[kMaxDofArgs], hi[kMaxDofArgs];
    double tb[2], db[2];
    int check_td, P, D, B, T;
};

__global__ void __launch_bounds__(256) k_validity(const ValidArgs v, const float* __restrict__ pos,
                                                  const float* __restrict__ params, uint8_t* __restrict__ valid) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= v.B) return;
    const int n = v.T * v.D;
    const float* p = pos + (size_t)b * n;
    bool ok = true;
    for (int e = lane; e < n; e += 64) {
        const int d = e % v.D;
        const double x = (double)p[e];
        ok = ok && (x >= v.lo[d]) && (x <= v.hi[d]);
    }
    if (v.check_td && lane == 0) {
        const double tau = (double)params[(size_t)b * v.P], delay = (double)params[(size_t)b * v.P + 1];
        ok = ok && tau >= v.tb[0] && tau <= v.tb[1] && delay >= v.db[0] && delay <= v.db[1];
    }
    const bool all_ok = __all(ok);
    if (lane == 0) valid[b] = all_ok ? 1 : 0;
}

int launch_validity(const float* pos, const float* params, int P, int D, const double* lo, const double* hi,
                    int check_td, const double* tb, const double* db, uint8_t* valid, int B, int T, void* stream) {
    ValidArgs v{};
    for (int d = 0; d < D; ++d) { v.lo[d] = lo[d]; v.hi[d] = hi[d]; }
    if (check_td) { v.tb[0] = tb[0]; v.tb[1] = tb[1]; v.db[0] = db[0]; v.db[1] = db[1]; }
    v.check_td = check_td; v.P = P; v.D = D; v.B = B; v.T = T;
    hipLaunchKernelGGL(k_validity, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, v, pos, params, valid);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}

}  // namespace mpk
