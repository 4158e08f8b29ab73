// gfx950 (CDNA4 / MI355X) kernels of libmpk.so.  Compiled with -ffp-contract=off: every fused multiply-add in
// this file is an explicit fmaf()/MFMA, every other a*b+c rounds twice exactly like the reference's separate
// torch / numpy ops.
//
//   k_build_shared   phase / exponential-kernel evaluation: per-time-step basis rows for a phase that all episodes
//                    share (tau, delay, init_time equal) -> A tables [n_out][KP][TS] (k-major, fp32) + aux[TS]
//   k_traj_shared    the [T x K] . [K x D] contraction on the matrix cores (v_mfma_f32_16x16x4_f32), 16 time steps x
//                    16 (episode, DoF) columns per tile, fused epilogue (ProDMP vel scaling, ProMP finite-difference
//                    velocity, DMP Euler integration, optional PD action), wave-private LDS transpose, float4 stores
//   k_traj_rows      per-episode phase (learned tau / delay, per-episode init_time): table gather / RBF evaluation
//                    per row, fp32 fmaf chains in the same k order as the MFMA
//   k_pd_rollout     tracking-controller + plant loop in float64 (black_box_wrapper.py:175-203)
//   k_replan_advance integer replanning bookkeeping; k_validity: joint-limit / bound check reduction
#include <hip/hip_runtime.h>

#include <cstdint>

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

// column k (< nb+3) of the ProDMP position / velocity rows at table index idx
__device__ __forceinline__ void prodmp_col(const DevCfg& c, const ProdmpBC& bc, int idx, const double xi[4], int k,
                                           float* h, float* hv) {
    const int N = c.n_pc, K = c.nb + 1;
    if (k < K) {
        const double* PB = c.tab + 4 * (size_t)N;
        const double* VB = PB + (size_t)N * K;
        const double pb = PB[(size_t)bc.idxb * K + k], vb = VB[(size_t)bc.idxb * K + k];
        *h = (float)(PB[(size_t)idx * K + k] - (xi[0] * pb + xi[1] * vb));
        *hv = (float)(VB[(size_t)idx * K + k] - (xi[2] * pb + xi[3] * vb));
    } else if (k == K) {
        *h = (float)xi[0]; *hv = (float)xi[2];
    } else {
        *h = (float)xi[1]; *hv = (float)xi[3];
    }
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
                prodmp_col(c, bc, idx, xi, k, &h, &hv);
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
            aux[t] = (c.base_times[th] + init_time) - (c.base_times[tl] + init_time);
        }
    } else {  // DMP: forcing rows phi*x, aux = diff of the fp32 scaled times
        for (int t = tid; t < T; t += 256) {
            const float time = c.base_times[t] + init_time;
            const double x = phase_f64(c, time, c.tau, c.delay, nullptr);
            rbf_cols(c, x, x, A + t, TS);
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
// k_traj_shared: MFMA contraction + fused epilogues
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
    int wave_floats, xtile_floats;
};

struct ActArgs {
    int controller_type;
    double pg[kMaxD], dg[kMaxD], lo[kMaxD], hi[kMaxD];
};

enum : int { XK_ZERO = 0, XK_PARAM = 1, XK_GOAL = 2, XK_IPOS = 3, XK_IVEL = 4 };

template <int MP, bool ACT>
__global__ void __launch_bounds__(256) k_traj_shared(const TrajArgs a, const ActArgs act) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : (MP == MPK_MP_PROMP ? 3 : 1);
    constexpr int NST = 2 + (ACT ? 1 : 0);
    const DevCfg& c = a.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KP = c.KP, KM = KP >> 2, TS = a.TS, D = c.D, T = c.T, P = c.P, B = a.B;
    const int SEG = 16 * D;
    const int sh = a.sh, DP = 1 << sh, NTW = 16 >> sh;

    float* sA = smem;                               // [NOUT][KP][TS]
    float* sAux = sA + NOUT * KP * TS;              // [TS]
    float* sW = sAux + TS + wave * a.wave_floats;   // wave-private
    float* sX = sW;                                 // [KP][17]
    float* sSt = sW + a.xtile_floats;               // [NTW][NST][SEG]
    float* sF = sSt + NTW * NST * SEG;              // DMP only: [NTW][SEG]

    {   // stage the shared basis tables through LDS once per workgroup
        const float4* src = reinterpret_cast<const float4*>(a.A);
        float4* dst = reinterpret_cast<float4*>(sA);
        const int n4 = (NOUT * KP * TS) >> 2;
        for (int i = tid; i < n4; i += 256) dst[i] = src[i];
        const float4* s2 = reinterpret_cast<const float4*>(a.aux);
        float4* d2 = reinterpret_cast<float4*>(sAux);
        for (int i = tid; i < (TS >> 2); i += 256) d2[i] = s2[i];
    }
    __syncthreads();

    // ---- lane constants -------------------------------------------------------------------------------
    const int col = lane & 15, q = lane >> 4;
    const int bl = col >> sh, d = col & (DP - 1);
    const bool dvalid = d < D;

    // X-tile element(s) this lane fetches: e -> (column xc, k xk), consecutive lanes walk k (contiguous params)
    int xkind[4], xoff[4], xbl[4], xd[4], xlds[4];
    float xscale[4];
    const unsigned inv = 65536u / (unsigned)KP + 1u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        int kind = XK_ZERO, off = 0;
        float scale = 1.0f;
        const int xc = (int)(((unsigned)e * inv) >> 16);
        const int xk = e - xc * KP;
        const int ebl = xc >> sh, ed = xc & (DP - 1);
        if (e < 16 * KP && ed < D) {
            if (MP == MPK_MP_PRODMP) {
                const int nb = c.nb;
                if (xk < nb) {
                    if (!c.disable_weights) { kind = XK_PARAM; off = c.off + ed * c.Kloc + xk; scale = c.scale[xk]; }
                } else if (xk == nb) {
                    kind = XK_GOAL; off = c.off + ed * c.Kloc + (c.disable_weights ? 0 : nb); scale = c.scale[nb];
                } else if (xk == nb + 1) {
                    kind = XK_IPOS;
                } else if (xk == nb + 2) {
                    kind = XK_IVEL;
                }
            } else if (MP == MPK_MP_PROMP) {
                if (xk < c.nb) { kind = XK_PARAM; off = c.off + ed * c.Kloc + xk; }
                else if (xk == c.nb && c.KT > c.nb) kind = XK_IPOS;
            } else {
                if (xk < c.nb) { kind = XK_PARAM; off = c.off + ed * c.Kloc + xk; scale = c.ws; }
            }
        }
        xkind[i] = kind; xoff[i] = off; xbl[i] = ebl; xd[i] = ed; xscale[i] = scale;
        xlds[i] = e < 16 * KP ? xk * 17 + xc : -1;
    }
    auto loadx = [&](int i, int b0) -> float {
        const int b = b0 + xbl[i];
        if (xkind[i] == XK_ZERO || b >= B) return 0.0f;
        switch (xkind[i]) {
            case XK_PARAM: return a.params[(size_t)b * P + xoff[i]] * xscale[i];
            case XK_GOAL: {
                float v = c.disable_goal ? 0.0f : a.params[(size_t)b * P + xoff[i]] * xscale[i];
                if (c.relative_goal) v = v + a.init_pos[(size_t)b * D + xd[i]];
                return v;
            }
            case XK_IPOS: return a.init_pos[(size_t)b * D + xd[i]];
            default: return a.init_vel[(size_t)b * D + xd[i]] * c.tau;
        }
    };

    // store mapping: float4 chunk i of the staging area -> (segment, chunk in segment)
    const int seg4 = SEG >> 2;               // float4 per full segment (= 4*D)
    const int total4 = NTW * NST * seg4;     // <= 192
    int sseg[3], soff[3];
    {
        const unsigned inv2 = 65536u / (unsigned)seg4 + 1u;
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int i = lane + 64 * it;
            const int s = (int)(((unsigned)i * inv2) >> 16);
            sseg[it] = i < total4 ? s : -1;
            soff[it] = i - s * seg4;
        }
    }

    const int NRT = (T + 15) >> 4;
    const int gw = blockIdx.x * 4 + wave, gstride = gridDim.x * 4;
    float xv[4];
    if (gw < a.G) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = loadx(i, gw * NTW);
    }

    for (int g = gw; g < a.G; g += gstride) {
        const int b0 = g * NTW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (xlds[i] >= 0) sX[xlds[i]] = xv[i];
        const int gn = g + gstride;
        if (gn < a.G) {  // prefetch the next group's parameters under this group's matrix work
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = loadx(i, gn * NTW);
        }
        const int b = b0 + bl;
        const bool cvalid = dvalid && b < B;
        double cp = 0.0, cv = 0.0;
        if (ACT) {
            if (cvalid) { cp = a.c_pos[(size_t)b * D + d]; cv = a.c_vel[(size_t)b * D + d]; }
        }
        float ey = 0.f, ez = 0.f, eg = 0.f;   // DMP Euler state (lanes q == 0)
        if (MP == MPK_MP_DMP) {
            if (cvalid && q == 0) {
                ey = a.init_pos[(size_t)b * D + d];
                ez = a.init_vel[(size_t)b * D + d] * c.tau;
                eg = a.params[(size_t)b * P + c.off + d * c.Kloc + c.nb] * c.gs;
            }
        }
        __builtin_amdgcn_wave_barrier();
        float xb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xb[m] = m < KM ? sX[(4 * m + q) * 17 + col] : 0.0f;

        for (int rt = 0; rt < NRT; ++rt) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
            const float* ap = sA + q * TS + rt * 16 + col;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m < KM) {
                    const float* am = ap + (4 * m) * TS;
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[0], xb[m], acc0, 0, 0, 0);
                    if (NOUT > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[KP * TS], xb[m], acc1, 0, 0, 0);
                    if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[2 * KP * TS], xb[m], acc2, 0, 0, 0);
                }
            }
            const int rows = min(16, T - rt * 16);
            if (MP != MPK_MP_DMP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tl = 4 * q + r, t = rt * 16 + tl;
                    if (cvalid && t < T) {
                        const float p = acc0[r];
                        float v;
                        if (MP == MPK_MP_PRODMP) v = acc1[r] / c.tau;
                        else v = (acc1[r] - acc2[r]) / sAux[t];
                        sSt[(bl * NST + 0) * SEG + tl * D + d] = p;
                        sSt[(bl * NST + 1) * SEG + tl * D + d] = v;
                        if (ACT) {
                            double u;
                            if (act.controller_type == MPK_CTRL_MOTOR)
                                u = act.pg[d] * ((double)p - cp) + act.dg[d] * ((double)v - cv);
                            else if (act.controller_type == MPK_CTRL_POSITION) u = (double)p;
                            else u = (double)v;
                            u = fmin(fmax(u, act.lo[d]), act.hi[d]);
                            sSt[(bl * NST + 2) * SEG + tl * D + d] = (float)u;
                        }
                    }
                }
            } else {
                if (cvalid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sF[bl * SEG + (4 * q + r) * D + d] = acc0[r];
                }
                __builtin_amdgcn_wave_barrier();
                if (cvalid && q == 0) {
                    // explicit Euler in scaled time, one rounding per op (no FMA), first sample = initial condition
                    for (int tl = 0; tl < rows; ++tl) {
                        const int t = rt * 16 + tl;
                        sSt[(bl * 2 + 0) * SEG + tl * D + d] = ey;
                        sSt[(bl * 2 + 1) * SEG + tl * D + d] = ez / c.tau;
                        if (t < T - 1) {
                            const float f = sF[bl * SEG + tl * D + d], ds = sAux[t];
                            const float t1 = eg - ey;
                            const float t2 = c.dmp_beta * t1;
                            const float t3 = t2 - ez;
                            const float t4 = c.dmp_alpha * t3;
                            const float acc = t4 + f;
                            ez = ez + ds * acc;
                            ey = ey + ds * ez;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // coalesced stores: every (episode, output) segment of this row tile is contiguous in HBM
            const int len = rows * D;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int s = sseg[it];
                if (s >= 0) {
                    const int sb = s / NST, j = s - sb * NST;
                    const int bb = b0 + sb, w4 = soff[it] * 4;
                    if (bb < B && w4 < len) {
                        float* outp = j == 0 ? a.pos : (j == 1 ? a.vel : a.actions);
                        float* gp = outp + ((size_t)bb * T + rt * 16) * D + w4;
                        const float4 val = *reinterpret_cast<const float4*>(sSt + s * SEG + w4);
                        if (a.vec_ok && w4 + 4 <= len) {
                            *reinterpret_cast<float4*>(gp) = val;
                        } else {
                            const float vv[4] = {val.x, val.y, val.z, val.w};
                            for (int e = 0; e < 4 && w4 + e < len; ++e) gp[e] = vv[e];
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int MP, bool ACT>
static int launch_traj_shared_t(const TrajArgs& ta, const ActArgs& aa, int blocks, size_t lds, void* stream) {
    auto kern = k_traj_shared<MP, ACT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, (hipStream_t)stream, ta, aa);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}

int launch_traj_shared(const DevCfg& c, const SharedTables& st, const float* params, const float* init_pos,
                       const float* init_vel, float* pos, float* vel, float* actions, const RolloutDev* rc,
                       const double* c_pos, const double* c_vel, int B, int num_cu, void* stream,
                       const char** kernel_name) {
    TrajArgs ta;
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
    const int SEG = 16 * c.D;
    ta.xtile_floats = (c.KP * 17 + 3) / 4 * 4;
    ta.wave_floats = ta.xtile_floats + NTW * nst * SEG + (c.mp_type == MPK_MP_DMP ? NTW * SEG : 0);
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    ta.vec_ok = ((c.T * c.D) % 4 == 0) && aligned16(pos) && aligned16(vel) && (!act || aligned16(actions));
    ActArgs aa{};
    if (act) {
        aa.controller_type = rc->controller_type;
        for (int d = 0; d < c.D; ++d) { aa.pg[d] = rc->pg[d]; aa.dg[d] = rc->dg[d]; aa.lo[d] = rc->lo[d]; aa.hi[d] = rc->hi[d]; }
    }
    const size_t lds = ((size_t)st.n_out * c.KP * st.TS + st.TS + 4 * (size_t)ta.wave_floats) * sizeof(float);
    if (lds > 160 * 1024) { set_error("trajectory too long for the shared-table kernel's LDS budget"); return MPK_EINVAL; }
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    int blocks = (ta.G + 3) / 4;
    const int cap = num_cu * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    switch (c.mp_type) {
        case MPK_MP_PRODMP:
            if (act) { *kernel_name = "k_traj_shared<prodmp,act>"; return launch_traj_shared_t<MPK_MP_PRODMP, true>(ta, aa, blocks, lds, stream); }
            *kernel_name = "k_traj_shared<prodmp>";
            return launch_traj_shared_t<MPK_MP_PRODMP, false>(ta, aa, blocks, lds, stream);
        case MPK_MP_PROMP:
            if (act) { *kernel_name = "k_traj_shared<promp,act>"; return launch_traj_shared_t<MPK_MP_PROMP, true>(ta, aa, blocks, lds, stream); }
            *kernel_name = "k_traj_shared<promp>";
            return launch_traj_shared_t<MPK_MP_PROMP, false>(ta, aa, blocks, lds, stream);
        default:
            *kernel_name = "k_traj_shared<dmp>";
            return launch_traj_shared_t<MPK_MP_DMP, false>(ta, aa, blocks, lds, stream);
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
            if (MP == MPK_MP_PRODMP) {
                const int nb = c.nb;
                if (k < nb) {
                    if (!c.disable_weights) v = prm[c.off + dd * c.Kloc + k] * c.scale[k];
                } else if (k == nb) {
                    if (!c.disable_goal) v = prm[c.off + dd * c.Kloc + (c.disable_weights ? 0 : nb)] * c.scale[nb];
                    if (c.relative_goal) v = v + a.init_pos[(size_t)b * D + dd];
                } else if (k == nb + 1) {
                    v = a.init_pos[(size_t)b * D + dd];
                } else {
                    v = a.init_vel[(size_t)b * D + dd] * tau;
                }
            } else if (MP == MPK_MP_PROMP) {
                if (k < c.nb) v = prm[c.off + dd * c.Kloc + k];
                else v = a.init_pos[(size_t)b * D + dd];
            } else {
                v = prm[c.off + dd * c.Kloc + k] * c.ws;
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
                    prodmp_col(c, bc, idx, xi, k, &h, &hv);
                    sH[t * KT + k] = h;
                    sH[(T + t) * KT + k] = hv;
                }
            }
        } else {
            for (int t = tid; t < T; t += nt) {
                const float time = c.base_times[t] + it;
                const double x = phase_f64(c, time, tau, delay, nullptr);
                rbf_cols(c, x, MP == MPK_MP_PROMP ? (double)c.ws : x, sH + t * KT, 1);
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
                a.vel[(size_t)b * T * D + e] = accv / tau;
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
                a.vel[(size_t)b * T * D + e] = (sP[th * D + dd] - sP[tl * D + dd]) / (sT[th] - sT[tl]);
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

int launch_pd_rollout(const RolloutDev& rc, int D, const float* des_pos, const float* des_vel, double* q, double* qd,
                      const int32_t* n_steps, float* actions, int B, int T, void* stream) {
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
