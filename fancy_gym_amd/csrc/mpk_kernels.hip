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
#include <type_traits>

#include "mpk_internal.h"

// Build partitioning (compile time only; the library is the same): this file is compiled once per MPK_PART and the
// objects are linked into libmpk.so, so that the ~300 kernel instantiations build on several cores (__graft_entry__.py).
//   -1 (default)  everything in one translation unit (one_kernel.sh, MPK_TRACE development builds)
//    0            everything but the shared-phase trajectory kernel families
//    1, 2, 3      launch_traj_ct<MP = MPK_PART - 1> and the k_traj_tiles / split / stream / quad / pipe instantiations
//                 behind it (promp, dmp, prodmp)
#ifndef MPK_PART
#define MPK_PART -1
#endif
#define MPK_MAIN (MPK_PART <= 0)
#if defined(MPK_TRACE) && MPK_PART >= 0
#error "MPK_TRACE builds are single translation unit builds (the trace buffer is one device variable)"
#endif

namespace mpk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Development build only (-DMPK_TRACE): wave 0 of workgroup `MPK_TRACE_BLOCK` stamps the shader clock at labelled points
// of a kernel into a device array that tools/dev/trace_kernel.py prints -- the per-phase timeline of ONE wave.
#ifdef MPK_TRACE
#ifndef MPK_TRACE_BLOCK
#define MPK_TRACE_BLOCK 0
#endif
// slot = tag (< 256): a stamp is one s_memtime and one fire-and-forget store -- no counter to fetch, nothing to wait for
// but the clock itself (a version that appended through a counter in memory paid a memory round trip per stamp and
// stretched the traced wave by half).  A tag stamped repeatedly keeps its last value.
__device__ long long g_trace[256];
#define MPK_STAMP_AT(tag, tid)                                                                      \
    do {                                                                                            \
        if (blockIdx.x == MPK_TRACE_BLOCK && threadIdx.x == (tid))                                  \
            g_trace[(tag) & 255] = (long long)__builtin_readcyclecounter();                         \
    } while (0)
#define MPK_STAMP(tag) MPK_STAMP_AT(tag, 0)
#else
#define MPK_STAMP(tag) do { } while (0)
#define MPK_STAMP_AT(tag, tid) do { } while (0)
#endif

#define MPK_LAUNCH_CHECK()                                                          \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            set_error(std::string("kernel launch: ") + hipGetErrorString(e_));      \
            return MPK_EHIP;                                                        \
        }                                                                           \
    } while (0)

// Kernels that may take more than the default 64 KB of dynamic LDS: the function attribute is raised ONCE per kernel
// instantiation to the CU's whole LDS (160 KB), not per launch with the launch's size -- hipFuncSetAttribute rewrites state of a
// function whose earlier launches may still be in flight (round 3: one silent runtime abort per ~30 000 launches of mixed
// configurations in the fuzz soak went away with this).
#ifndef MPK_DEVICE_ONLY
template <class K>
static hipError_t allow_full_lds(K kern) {
    // per kernel instantiation AND device (the attribute belongs to the current device's copy of the function)
    static signed char done[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024);
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev] = 1;
    return e;
}
#endif

#if MPK_MAIN
size_t shared_tables_floats(const DevCfg& c, int* TS, int* n_out) {
    const int TP = (c.T + 15) / 16 * 16;
    const int ts = ((TP + 15) / 32) * 32 + 16;  // TS % 32 == 16: the two k rows of a 32-lane LDS read hit disjoint banks
    const int no = c.mp_type == MPK_MP_PRODMP ? 2 : (c.mp_type == MPK_MP_PROMP ? 3 : 1);
    *TS = ts;
    *n_out = no;
    // the k-major table A [n_out][KP][TS] (MFMA fragment loads) followed by its step-major copy At [TS][n_out * KP]
    // (one contiguous row per time step: the serial role of k_traj_split reads it with scalar loads)
    return 2 * (size_t)no * c.KP * ts;
}
#endif  // MPK_MAIN

// ------------------------------------------------------------------------------------------------------------
// device helpers shared by the shared-phase builder and the per-episode kernel
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float scaled_time(float t, float delay, float tau) {
    // left-bounded linear phase in fp32: max((t - delay) / tau, 0); IEEE division
    return fmaxf((t - delay) / tau, 0.0f);
}

// vel = z / tau for DMP (the reference divides the fp32 tensor z by tau).  The divisor is an episode / kernel constant,
// so the reciprocal is taken once and each quotient costs three instructions instead of the ~10 of an IEEE division:
//     q = z * r,   q' = fma(fma(-tau, q, z), r, q)            (Markstein's correction step)
// which is the correctly rounded quotient except for rare last-bit cases -- seven orders below the 1e-5 contract.  Every
// DMP kernel uses this helper, so they keep producing identical bits.
struct TauDiv { float tau, r; };
__device__ __forceinline__ TauDiv make_tau_div(float tau) { return TauDiv{tau, 1.0f / tau}; }
__device__ __forceinline__ float div_tau(float z, const TauDiv& t) {
    const float q = z * t.r;
    return __builtin_fmaf(__builtin_fmaf(-t.tau, q, z), t.r, q);
}

// The CORRECTLY ROUNDED fp32 quotient z / d for a divisor that is reused (an episode's tau, the table's grid step): the
// reciprocal r = RN(1 / d) is taken once with an IEEE division; then
//     q = RN(z * r),   e = RN(z - d * q)  (exact: one fma),   q' = RN(q + e * r)
// is RN(z / d) whenever the significand of d is not all ones and nothing over- or underflows (Markstein 1990; Muller et
// al., Handbook of Floating-Point Arithmetic, section 4.7: a correctly rounded reciprocal and a quotient estimate within
// one ulp make the correction step exact).  The one excluded divisor pattern takes the IEEE division.  This feeds the
// ProDMP table indices -- the integer part of the path -- so tests/test_gpu_edge_cases.py sweeps every fp32 numerator a
// BASELINE time grid can produce against the IEEE division for 64 divisors (identical, 3 x 10^9 quotients).
struct ExactDiv { float d, r; bool plain; };
__device__ __forceinline__ ExactDiv make_exact_div(float d) {
    return ExactDiv{d, 1.0f / d, (__float_as_uint(d) & 0x7fffffu) == 0x7fffffu};
}
__device__ __forceinline__ float div_exact(float z, const ExactDiv& x) {
    if (x.plain) return z / x.d;                               // wave-uniform for a per-episode divisor
    const float q = z * x.r;
    return __builtin_fmaf(__builtin_fmaf(-x.d, q, z), x.r, q);
}

// Write-through (sc1) stores are chosen while a launch's outputs still fit the memory-side cache (256 MB + the L2s): plain
// write-back stores fall off a cliff once the dirty lines exceed it, write-through stores lose once the outputs stream to
// HBM anyway.  Measured per kernel family and size (profiles/r03_streaming_wt.md, us plain vs write-through): k_traj_flat
// +actions 57.1 / 46.1 at 241 MB, 67.5 / 51.9 at 275 MB, 74.3 / 71.3 at 310 MB, 77.7 / 88.6 at 344 MB; cfg3 k_traj_quad<dmp>
// 36.5 / 33.6 at 175 MB, 58.2 / 54.3 at 262 MB, 99 / 104 at 350 MB; closed loop k_traj_duo 80.8 / 70.1 at 262 MB, 173 / 221 at
// 525 MB; per-episode kernels 44.2 / 41.2 at 175 MB, 107 / 117 at 350 MB.  (The tile-major kernels keep their own 96 MB rule:
// above it the episode-major kernels take over.)
constexpr double kWtBytes = 300.0 * 1024 * 1024;

// The integer part of BlackBoxWrapper.step's loop (black_box_wrapper.py:174,197,206) for one episode and one plan:
// how many steps this plan executes before the loop breaks (end of the horizon, or the schedule t % every == 0 while
// plan_steps < max_planning_times), and the counters after it.  k_replan_advance and the fused closed-loop kernels both
// call this, `writer` = the one lane per episode that stores the new state.
__device__ __forceinline__ int replan_rule(const ReplanDev& rp, int b, int T, bool writer) {
    // three independent loads (one memory round trip): this sits in front of a serial recurrence
    const uint8_t was_done = rp.done[b];
    const int cur = rp.traj_steps[b];
    const int plan = rp.plan_steps[b] + 1;
    // first global step g = cur + t + 1 (t >= 0) at which the loop breaks
    int g_break = rp.horizon;
    if (plan < rp.max_planning_times) {
        const int gm = (cur / rp.every + 1) * rp.every;  // next multiple of `every` strictly above cur
        g_break = gm < rp.horizon ? gm : rp.horizon;
    }
    int seg = g_break - cur;
    if (seg > T) seg = T;
    if (seg < 1) seg = 1;
    if (was_done) seg = 0;                               // a finished episode is left alone
    if (writer) {
        rp.seg_len[b] = seg;
        if (!was_done) {
            const uint8_t dn = (cur + seg) >= rp.horizon ? 1 : 0;
            rp.plan_steps[b] = plan;
            rp.traj_steps[b] = cur + seg;
            rp.done[b] = dn;
            if (rp.done_out) rp.done_out[b] = dn;
        } else if (rp.done_out) {
            rp.done_out[b] = 1;
        }
    }
    return seg;
}

// Basis tables -> LDS, once per workgroup of 256 threads: every thread issues ALL its loads (up to four chunks of the
// rows, one of the aux row) before its first LDS write -- one memory round trip instead of one per loop iteration,
// which matters for launches that give a wave a single work unit.  Longer tables take plain loops after that.
__device__ __forceinline__ void stage_tables(const float* __restrict__ A, const float* __restrict__ aux, float* sA,
                                             float* sAux, int nA4, int nX4, int tid) {
    const float4* src = reinterpret_cast<const float4*>(A);
    const float4* s2 = reinterpret_cast<const float4*>(aux);
    float4* dst = reinterpret_cast<float4*>(sA);
    float4* d2 = reinterpret_cast<float4*>(sAux);
    float4 x = {0.f, 0.f, 0.f, 0.f};
    float4 r[4] = {x, x, x, x};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k;
        if (i < nA4) r[k] = src[i];      // (a select between src[i] and a private zero would become a flat load)
    }
    if (tid < nX4) x = s2[tid];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k;
        if (i < nA4) dst[i] = r[k];
    }
    if (tid < nX4) d2[tid] = x;
    for (int i = tid + 1024; i < nA4; i += 256) dst[i] = src[i];
    for (int i = tid + 256; i < nX4; i += 256) d2[i] = s2[i];
}

__device__ __forceinline__ int prodmp_index(float s, float scaled_dt) {
    // times_to_indices: round-half-even of the fp32 quotient -- the bit-exact integer part of the path
    return (int)rintf(s / scaled_dt);
}

__device__ __forceinline__ int wave_of(unsigned tid) { return __builtin_amdgcn_readfirstlane((int)(tid >> 6)); }

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
//   k == nb+1 : xi1  (+ H_g for a relative goal: goal = scale*g + y_b; MPK_RELGOAL_BEFORE_SCALE: + scale*H_g,
//               goal = scale*(g + y_b))
//   k == nb+2 : xi2 * tau                               (v_b = tau * ydot_b)
//   k == nb+3 : H_g * goal_offset, contracted with x = 1 (MPK_GOAL_OFFSET_ADD only: goal += goal_offset)
// and the velocity row additionally carries the 1/tau of  vel = (...)/tau.  Everything is folded in float64 and
// rounded ONCE to fp32.
__device__ __forceinline__ void prodmp_col(const DevCfg& c, const ProdmpBC& bc, int idx, const double xi[4], int k,
                                           double tau, double inv_tau, float* h, float* hv) {
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
            if (c.relgoal_before_scale) {
                const double sg = (VB + (size_t)N * K)[c.nb];
                gp *= sg; gv *= sg;
            }
            p += gp; v += gv;
        }
    } else if (k == K + 1) {
        p = xi[1] * tau; v = xi[3] * tau;
    } else {
        hcol(c.nb, &p, &v);
        p *= (double)c.goal_offset; v *= (double)c.goal_offset;
    }
    *h = (float)p;
    *hv = (float)(v * inv_tau);
}

// Lean float64 helpers for the basis rows.  The library exp()/divide carry special-case handling the rows never need
// (arguments are finite and <= 0, divisors are positive and normal); these keep ~1e-15 relative accuracy, far inside
// the single rounding to fp32 that follows, at a third of the instructions.  Every basis row on the device -- shared
// tables and per-episode rows alike -- goes through the same two functions, so the two paths agree bit for bit.
// coefficients of exp_nonpos: [0] log2(e), [1..2] -ln2 split hi / lo, [3..14] Taylor 1/11! .. 1/0! (remainder < 7e-15
// for |r| <= ln2/2).  ExpLiteral folds them into the instruction stream; ExpRegs holds them in registers loaded once
// per kernel (64-bit literals cost a move per use and the scalar registers that would hold them are the scarce ones
// in the per-episode kernel).  Same values, same operation order: same bits.
static __device__ double kExpTab[15] = {   // not const: a const table would be folded back into literals
    1.4426950408889634074, -6.93147180369123816490e-01, -1.90821492927058770002e-10,
    2.50521083854417187751e-08, 2.75573192239858906526e-07, 2.75573192239858906526e-06, 2.48015873015873015873e-05,
    1.98412698412698412698e-04, 1.38888888888888888889e-03, 8.33333333333333333333e-03, 4.16666666666666666667e-02,
    1.66666666666666666667e-01, 0.5, 1.0, 1.0};

struct ExpLiteral {
    __device__ __forceinline__ double operator[](int i) const {
        constexpr double t[15] = {
            1.4426950408889634074, -6.93147180369123816490e-01, -1.90821492927058770002e-10,
            2.50521083854417187751e-08, 2.75573192239858906526e-07, 2.75573192239858906526e-06,
            2.48015873015873015873e-05, 1.98412698412698412698e-04, 1.38888888888888888889e-03,
            8.33333333333333333333e-03, 4.16666666666666666667e-02, 1.66666666666666666667e-01, 0.5, 1.0, 1.0};
        return t[i];
    }
};

struct ExpRegs {
    double t[15];
    __device__ __forceinline__ void load() {
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            t[i] = kExpTab[i];
            asm volatile("" : "+v"(t[i]));      // vector registers: the scalar file is what this kernel runs out of
        }
    }
    __device__ __forceinline__ double operator[](int i) const { return t[i]; }
};

template <class CF>
__device__ __forceinline__ double exp_nonpos(double x, const CF& cf) {
    x = fmax(x, -700.0);                                        // exp(-700) ~ 1e-304: still normal, rounds to 0.0f
    // (arguments are <= 0 everywhere but in RbfRecur's ratio, which stays far below the overflow threshold)
    const double n = rint(x * cf[0]);
    double r = fma(n, cf[1], x);
    r = fma(n, cf[2], r);
    double p = cf[3];
#pragma unroll
    for (int i = 4; i < 15; ++i) p = fma(p, r, cf[i]);
    return ldexp(p, (int)n);
}

__device__ __forceinline__ double exp_nonpos(double x) { return exp_nonpos(x, ExpLiteral()); }

// num / den for den > 0: v_rcp_f64 seed, two Newton steps, one residual fix-up.  The refined reciprocal depends on the
// divisor alone, so a divisor that is reused (an episode's tau) takes it once (PosDiv) -- the same operations on the same
// values as the one-shot form, hence the same bits.
struct PosDiv { double den, y; };
__device__ __forceinline__ PosDiv make_pos_div(double den) {
    double y = __builtin_amdgcn_rcp(den);
    y = fma(fma(-den, y, 1.0), y, y);
    y = fma(fma(-den, y, 1.0), y, y);
    return PosDiv{den, y};
}
__device__ __forceinline__ double div_pos(double num, const PosDiv& d) {
    const double q = num * d.y;
    return fma(fma(-d.den, q, num), d.y, q);
}
__device__ __forceinline__ double div_pos(double num, double den) { return div_pos(num, make_pos_div(den)); }

// bounded phase in float64 from an fp32 time value and fp32-held tau/delay (promp / dmp rows)
template <class CF>
__device__ __forceinline__ double phase_f64(const DevCfg& c, float time, const PosDiv& tau, float delay, const CF& cf) {
    const double s = div_pos((double)time - (double)delay, tau);
    if (c.phase_type == MPK_PHASE_LINEAR) return fmin(fmax(s, 0.0), 1.0);
    return exp_nonpos(-(double)c.alpha_phase * fmax(s, 0.0), cf);
}
template <class CF>
__device__ __forceinline__ double phase_f64(const DevCfg& c, float time, float tau, float delay, const CF& cf) {
    return phase_f64(c, time, make_pos_div((double)tau), delay, cf);
}



// Equally spaced centres with one bandwidth (every linear-phase configuration: the centres are equally spaced in time,
// SURVEY A.4): the Gaussians e_k = exp(-bw (x - c_k)^2 / 2), c_k = c_0 + k D, obey
//     e_{k+1} = e_k r_k,   r_k = exp(bw D (x - c_k) - bw D^2 / 2),   r_{k+1} = r_k exp(-bw D^2)
// -- TWO exponentials per row and two float64 multiplications per further basis function instead of one exponential
// each (a row of cfg5's five RBFs: 60 % of its instructions were exponentials).  Error of e_k relative to the direct
// exponential: the ratio's exponent argument is as large as ~600, so r_0 carries ~600 x 1.1e-16 = 7e-14 relative
// error (plus exp's own 1e-15), q ~2e-16; e_k = e_0 r_0^k q^(k(k-1)/2) therefore ~k 7e-14 + k^2 2e-16: 1.4e-12 at
// k = 20, 3e-10 at k = 1000 -- still more than two orders below the single rounding to fp32 (6e-8) that follows.
// Every row builder on the device goes through the same code (k_dmp_prestep included), so the shared-phase and
// per-episode kernels keep producing identical bits.  The host enables it
// (DevCfg::rbf_uniform) only where e_0 cannot underflow; constants behind the bandwidths in the device table:
// [bw D, bw D^2 / 2, exp(-bw D^2)].
struct RbfRecur {
    double e, r, q;
    template <class CF>
    __device__ __forceinline__ RbfRecur(const double* cen, const double* bw, int n_total, double x, const CF& cf) {
        const double* k3 = bw + n_total;
        const double dx0 = x - cen[0];
        e = exp_nonpos(-(dx0 * dx0 * bw[0]) * 0.5, cf);
        r = exp_nonpos(k3[0] * dx0 - k3[1], cf);
        q = k3[2];
    }
    __device__ __forceinline__ double next() { const double v = e; e *= r; r *= q; return v; }
};

// normalised RBF row: writes nb learnable columns scaled by `mul` (column zs.. of the zero-padded family)
__device__ __forceinline__ void rbf_cols(const DevCfg& c, double x, double mul, float* out, int stride) {
    const double* cen = c.tab;
    const double* bw = c.tab + c.n_total;
    if (c.rbf_uniform) {
        RbfRecur s1(cen, bw, c.n_total, x, ExpLiteral());
        double sum = 0.0;
        for (int k = 0; k < c.n_total; ++k) sum += s1.next();
        const double scale = div_pos(mul, sum);
        RbfRecur s2(cen, bw, c.n_total, x, ExpLiteral());
        for (int k = 0; k < c.zs + c.nb; ++k) {
            const double ek = s2.next();
            if (k >= c.zs) out[(size_t)(k - c.zs) * stride] = (float)(ek * scale);
        }
        return;
    }
    double sum = 0.0;
    for (int k = 0; k < c.n_total; ++k) {
        const double dx = x - cen[k];
        sum += exp_nonpos(-(dx * dx * bw[k]) * 0.5);
    }
    const double scale = c.n_total > 1 ? div_pos(mul, sum) : mul;
    for (int k = 0; k < c.nb; ++k) {
        const double dx = x - cen[c.zs + k];
        out[(size_t)k * stride] = (float)(exp_nonpos(-(dx * dx * bw[c.zs + k]) * 0.5) * scale);
    }
}

// rbf_cols into a register row of KS columns (static indices only): columns nb.. stay as the caller set them; the
// promp "+ init_pos" column nb is set to 1 when the configuration has it.  Same arithmetic as rbf_cols.  cen / bw: the
// caller's LDS copy of the centres / bandwidths (a load from c.tab would sit in the memory queue behind the stores).
template <int KS, class CF>
__device__ __forceinline__ void rbf_row(const DevCfg& c, const double* cen, const double* bw, double x, double mul,
                                        float (&h)[KS], const CF& cf) {
    constexpr int NE = KS + 2;
    if (c.zs <= 2 && c.n_total <= NE) {
        // every RBF once: the learnable columns are e[zs .. zs + nb)
        double e[NE], sum = 0.0;
        if (c.rbf_uniform) {
            RbfRecur rr(cen, bw, c.n_total, x, cf);          // the same operations as rbf_cols: same bits
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                e[k] = 0.0;
                if (k < c.n_total) { e[k] = rr.next(); sum += e[k]; }
            }
        } else {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            e[k] = 0.0;
            if (k < c.n_total) {
                const double dx = x - cen[k];
                e[k] = exp_nonpos(-(dx * dx * bw[k]) * 0.5, cf);
                sum += e[k];
            }
        }
        }
        const double scale = c.n_total > 1 ? div_pos(mul, sum) : mul;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k < c.nb) {
                const double v = c.zs == 0 ? e[k] : (c.zs == 1 ? e[k + 1] : e[k + 2]);
                h[k] = (float)(v * scale);
            } else if (k == c.nb && c.KT > c.nb) {
                h[k] = 1.0f;
            }
        }
        return;
    }
    if (c.rbf_uniform) {
        RbfRecur s1(cen, bw, c.n_total, x, cf);
        double sum = 0.0;
        for (int k = 0; k < c.n_total; ++k) sum += s1.next();
        const double scale = div_pos(mul, sum);
        RbfRecur s2(cen, bw, c.n_total, x, cf);
        for (int k = 0; k < c.zs; ++k) (void)s2.next();
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k < c.nb) h[k] = (float)(s2.next() * scale);
            else if (k == c.nb && c.KT > c.nb) h[k] = 1.0f;
        }
        return;
    }
    double sum = 0.0;
    for (int k = 0; k < c.n_total; ++k) {
        const double dx = x - cen[k];
        sum += exp_nonpos(-(dx * dx * bw[k]) * 0.5, cf);
    }
    const double scale = c.n_total > 1 ? div_pos(mul, sum) : mul;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        if (k < c.nb) {
            const double dx = x - cen[c.zs + k];
            h[k] = (float)(exp_nonpos(-(dx * dx * bw[c.zs + k]) * 0.5, cf) * scale);
        } else if (k == c.nb && c.KT > c.nb) {
            h[k] = 1.0f;
        }
    }
}

#if MPK_MAIN
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
                       st.n_out, idx_out, range_flag);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY
#endif  // MPK_MAIN

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
    // tile-major work assignment: wave w owns row tile w % NRT of groups w / NRT, + gstride, ...; nrt_magic =
    // 2^32 / NRT + 1 (NRT > 1) turns the division into a multiply-high
    unsigned nrt_magic;
    int gstride;
    // episode-major kernels: write-through (sc1) stores while the outputs are cache resident (a serial-recurrence launch
    // of a few thousand episodes: closed-loop step at B = 4096 22.7 -> 18.7 us); plain stores once they stream to HBM
    // (write-through costs 25 % there).  The tile-major kernel has the policy as a template parameter.
    int wt;
    int flat_img;          // k_traj_flat: floats per whole-trajectory array image (NTW * T * D); 0 = another kernel runs
    unsigned ser_blocks;   // k_traj_split: workgroups [0, ser_blocks) run the serial role
    // closed-loop rollout fused into the episode-major kernel (CT >= 3)
    double* q_state;       // [B, D] plant position, in/out
    double* qd_state;      // [B, D] plant velocity, in/out
    const int32_t* n_steps;  // [B] executed steps of this plan (NULL = T)
    double plant_dt;
    ReplanDev rp;            // closed loop only: integer replanning state advanced in the kernel (replaces n_steps)
};

struct ActArgs {
    double pg[kMaxD], dg[kMaxD], lo[kMaxD], hi[kMaxD];
};

constexpr int kStageStride = 256;   // floats between output arrays in the wave's LDS staging area (>= NTW*16*D)
constexpr int kStageFloats = 4 * kStageStride;   // pos | vel | actions or DMP forcing | controller constants

enum : int { XK_ZERO = 0, XK_PARAM = 1, XK_IPOS = 2, XK_IVEL = 3, XK_ONE = 4 };

// which raw input feeds element k of a DoF's extended parameter column, and its offset inside the DoF's local block
template <int MP>
__device__ __forceinline__ int x_kind(const DevCfg& c, int k, int* loc) {
    // select form (no early returns): this runs in the latency-critical prologue of every trajectory kernel
    const int nb = c.nb;
    if (MP == MPK_MP_PRODMP) {
        const bool isw = k < nb, isg = k == nb;
        // the offset is used for an UNCONDITIONAL load (the kind decides afterwards whether the value counts), so it must
        // stay inside the DoF's local block whatever is disabled: with disable_goal the block has nb entries (no goal at
        // [nb]), with disable_weights one (the goal at [0]).  Round 3's fuzz soak found the old `isw ? k : ...`: the last
        // DoF of the last episode read one float (disable_goal) or up to nb - 1 floats (disable_weights) past the end of
        // `params` -- a memory fault once every ~10^4 random configurations, when the buffer ends on a page boundary.
        *loc = (isw && !c.disable_weights) ? k : ((isg && !c.disable_goal && !c.disable_weights) ? nb : 0);
        const int kw = c.disable_weights ? XK_ZERO : XK_PARAM, kg = c.disable_goal ? XK_ZERO : XK_PARAM;
        const int klast = (k == nb + 3 && c.goal_off_on) ? XK_ONE : XK_ZERO;
        return isw ? kw : (isg ? kg : (k == nb + 1 ? XK_IPOS : (k == nb + 2 ? XK_IVEL : klast)));
    } else if (MP == MPK_MP_PROMP) {
        const bool isw = k < nb;
        *loc = isw ? k : 0;
        return isw ? XK_PARAM : ((k == nb && c.KT > nb) ? XK_IPOS : XK_ZERO);
    } else {
        const bool isw = k < nb;
        *loc = isw ? k : 0;
        return isw ? XK_PARAM : XK_ZERO;
    }
}

// lane-constant description of a lane's role in the 16x16 tile machinery
template <int KM>
struct LaneMap {
    int col, q, bl, d, dsafe, NTW;
    bool dvalid;
    bool isp[KM], isip[KM], isiv[KM];
    float cst[KM];       // what an element that is no input carries: 0, or 1 for the goal-offset column
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
        L.cst[m] = L.dvalid && kind == XK_ONE ? 1.0f : 0.0f;
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

template <class T>
__device__ __forceinline__ T ld_off(const T* base, unsigned byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

template <int MP, bool ACT, int KM>
__device__ __forceinline__ GroupIn<KM> load_group(const TrajArgs& a, const LaneMap<KM>& L, int g) {
    const DevCfg& c = a.c;
    GroupIn<KM> in;
    // the last group may be ragged: clamp its missing episodes onto the group's first one (computed, never stored)
    const int b0 = g * L.NTW;
    const bool bv = b0 + L.bl < a.B;
    // wave-uniform block bases + 32-bit per-lane BYTE offsets: the loads take the (scalar base, vector offset) form
    // instead of a 64-bit address addition per load on the vector ALU
    const float* pb = a.params + (size_t)b0 * c.P;
    const unsigned io = bv ? L.ioff : (unsigned)L.dsafe;
    const unsigned pclamp = bv ? 0u : (unsigned)(L.bl * c.P);
#pragma unroll
    for (int m = 0; m < KM; ++m) in.raw[m] = ld_off(pb, 4u * (L.poff[m] - pclamp));
    in.ip = MP != MPK_MP_DMP ? ld_off(a.init_pos + (size_t)b0 * c.D, 4u * io) : 0.0f;
    in.iv = MP == MPK_MP_PRODMP ? ld_off(a.init_vel + (size_t)b0 * c.D, 4u * io) : 0.0f;
    in.cp = 0.0; in.cv = 0.0;
    if (ACT) { in.cp = ld_off(a.c_pos + (size_t)b0 * c.D, 8u * io); in.cv = ld_off(a.c_vel + (size_t)b0 * c.D, 8u * io); }
    return in;
}

template <int KM>
__device__ __forceinline__ void finish_group(const LaneMap<KM>& L, const GroupIn<KM>& in, float (&xb)[KM]) {
#pragma unroll
    for (int m = 0; m < KM; ++m) xb[m] = L.isp[m] ? in.raw[m] : (L.isip[m] ? in.ip : (L.isiv[m] ? in.iv : L.cst[m]));
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

// controller constants of one lane's DoF
struct Gains { double pg, dg, lo, hi; };

__device__ __forceinline__ Gains parked_gains(const double* sg) { return Gains{sg[0], sg[16], sg[32], sg[48]}; }

// The same constants straight from the kernel-argument segment with per-lane (vector) loads: the segment is ordinary
// device memory, so a lane-dependent index costs four 8-byte loads issued next to the kernel's first input loads,
// where selecting among scalar kernarg registers costs eight dependent s_load round trips and 16 exec-masked moves
// before any input load is issued (the tile-major kernel's whole life is ~8 us: its prologue is not free).
// `act` is the second kernel argument of every trajectory kernel.
constexpr size_t kActArgsOffset = (sizeof(TrajArgs) + alignof(ActArgs) - 1) / alignof(ActArgs) * alignof(ActArgs);
__device__ __forceinline__ Gains kernarg_gains(int d) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + kActArgsOffset;
    typedef const __attribute__((address_space(4))) double* dptr;
    Gains gn;
    gn.pg = ((dptr)(base + offsetof(ActArgs, pg)))[d];
    gn.dg = ((dptr)(base + offsetof(ActArgs, dg)))[d];
    gn.lo = ((dptr)(base + offsetof(ActArgs, lo)))[d];
    gn.hi = ((dptr)(base + offsetof(ActArgs, hi)))[d];
    return gn;
}

// The step loop of black_box_wrapper.py:175-203 on the reference's torque double integrator (base_reacher_torque.py:25-26)
// for the 16 steps of one row tile of ONE (episode, DoF) lane: float64, no FMA contraction -- numpy's promotion in
// pd_controller.py:21-29.  The desired states of the tile are pulled into registers first, then the chain runs as
// straight-line code WITHOUT control flow: a step past the executed ones (t >= nst) is computed and discarded by selects
// (its action is written as 0).  Measured on one wave (tools/dev/trace_kernel.py, tools/dev/rec_latency.hip,
// profiles/r02_closed_loop.md): with two exec-mask branches per step (t == tcond, t < nst) a step cost 260 cycles; this
// form costs 83 in isolation (57 for the bare chain of 11 float64 operations, the rest conversions and the LDS write).
// Feeding the chain float64 values from LDS (conversions done by all 64 lanes beforehand) measured the same 83 in
// isolation and SLOWER in the kernel (an extra LDS pass and barrier per tile: 18.3 vs 14.4 us), so it stays as it is.
// MASKED = false is the version for a tile every step of which is executed by every lane of the wave (the caller tests
// that wave-uniformly).  sP / sV / sA: the lane's (row 0, column) slots of the desired pos / vel / action images,
// `stride` floats per row.
template <int CTRL, bool MASKED, bool INTEGRATE = true, bool KEEP64 = false>
__device__ __forceinline__ void pd_tile_steps(const float* __restrict__ sP, const float* __restrict__ sV,
                                              float* __restrict__ sA, const int stride, const int t0, const int nst,
                                              const double pgd, const double dgd, const double lod, const double hid,
                                              const double dtp, double& qs, double& qds, double* __restrict__ q64 = nullptr,
                                              double* __restrict__ u64 = nullptr) {
    // INTEGRATE = false: MPK_PLANT_STATIC (the state never changes).  KEEP64: the plant position after the step and the
    // clipped action also stay in LDS as float64, 16 doubles per step (the reward pass of the reacher rollout reads them)
    float pr[16], vr[16];
#pragma unroll
    for (int tl = 0; tl < 16; ++tl) { pr[tl] = sP[tl * stride]; vr[tl] = sV[tl * stride]; }
#pragma unroll
    for (int tl = 0; tl < 16; ++tl) {
        const double dp = (double)pr[tl], dv = (double)vr[tl];
        double u;
        if (CTRL == MPK_CTRL_MOTOR) u = pgd * (dp - qs) + dgd * (dv - qds);
        else if (CTRL == MPK_CTRL_POSITION) u = dp;
        else u = dv;
        u = fmin(fmax(u, lod), hid);
        const double qds_n = INTEGRATE ? qds + dtp * u : qds;
        const double qs_n = INTEGRATE ? qs + dtp * qds_n : qs;
        if (MASKED) {
            const bool live = t0 + tl < nst;
            qds = live ? qds_n : qds;
            qs = live ? qs_n : qs;
            u = live ? u : 0.0;
            sA[tl * stride] = (float)u;
        } else {
            qds = qds_n; qs = qs_n;
            sA[tl * stride] = (float)u;
        }
        if (KEEP64) { q64[tl * 16] = qs; u64[tl * 16] = u; }
    }
}

// clip(u, lo, hi) of the step loop as the two instructions it is: fmin / fmax make the compiler re-quiet a loop-invariant
// bound before every use (a v_max_f64 x, x per bound and step -- two of the ~14 float64 operations of a step).  lo / hi
// are finite controller bounds or +-inf, never NaN; u is quieted by the instructions themselves (IEEE mode).
__device__ __forceinline__ double clip_f64(double u, double lo, double hi) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(u), "v"(lo));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(r), "v"(hi));
    return r;
}

// all lanes of the wave: does every serial lane execute every step of the tile [t0, t0 + 16)?  (wave-uniform)
__device__ __forceinline__ bool tile_fully_executed(bool serial, int nst, int t0) {
    return __all(!serial || nst >= t0 + 16) != 0;
}

// DMP's explicit Euler recurrence for the 16 steps of one row tile of one (episode, DoF) lane, fp32, one
// rounding per op, branch-free like pd_tile_steps: a step at or past T - 1 leaves the state alone by select.
__device__ __forceinline__ void dmp_tile_steps(const float* __restrict__ sF, float* __restrict__ sP, float* __restrict__ sV,
                                               const float* __restrict__ ds16, const int stride, const int t0,
                                               const int T, const float alpha, const float beta, const float eg,
                                               const TauDiv& td, float& ey, float& ez) {
    float fr[16], dsr[16];
#pragma unroll
    for (int tl = 0; tl < 16; ++tl) fr[tl] = sF[tl * stride];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 x = reinterpret_cast<const float4*>(ds16)[j];
        dsr[4 * j] = x.x; dsr[4 * j + 1] = x.y; dsr[4 * j + 2] = x.z; dsr[4 * j + 3] = x.w;
    }
#pragma unroll
    for (int tl = 0; tl < 16; ++tl) {
        sP[tl * stride] = ey;
        sV[tl * stride] = div_tau(ez, td);               // vel = z / tau, off the dependent chain
        const float t1 = eg - ey;
        const float t2 = beta * t1;
        const float t3 = t2 - ez;
        const float t4 = alpha * t3;
        const float acc = t4 + fr[tl];
        const float ez_n = ez + dsr[tl] * acc;
        const float ey_n = ey + dsr[tl] * ez_n;
        const bool live = t0 + tl < T - 1;
        ez = live ? ez_n : ez;
        ey = live ? ey_n : ey;
    }
}

// epilogue of one C tile into the wave-private LDS transpose buffer (rows beyond T land in rows never stored)
template <int MP, int CT>
__device__ __forceinline__ void tile_epilogue(const f32x4& acc0, const f32x4& acc1, const f32x4& acc2,
                                              const float (&dtd)[4], double cp, double cv, const Gains& gn,
                                              float* sSt, unsigned wofs, int D, const int astride = kStageStride,
                                              const int nrows = 4) {
    // astride: floats between the pos / vel / action images; nrows: rows of this lane's four that exist (k_traj_flat's
    // whole-trajectory images have no spare rows behind step T - 1; the transpose buffers do: 4)
    const double pgd = gn.pg, dgd = gn.dg, lod = gn.lo, hid = gn.hi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r >= nrows) break;
        const float p = acc0[r];
        float v;
        if (MP == MPK_MP_PRODMP) v = acc1[r];            // 1/tau is folded into the velocity rows
        else v = (acc1[r] - acc2[r]) * dtd[r];           // forward difference of fp32 positions x (1 / dt)
        float* w = sSt + wofs + r * D;
        w[0] = p;
        w[astride] = v;
        if (CT >= 0 && CT < 3) {
            // float64 without FMA: numpy's promotion in pd_controller.py:21-29 (fp32 desired (+) fp64 state)
            double u;
            if (CT == MPK_CTRL_MOTOR) u = pgd * ((double)p - cp) + dgd * ((double)v - cv);
            else if (CT == MPK_CTRL_POSITION) u = (double)p;
            else u = (double)v;
            u = fmin(fmax(u, lod), hid);
            w[2 * astride] = (float)u;
        }
        // closed loop: actions of steps the plan does not execute are 0; the recurrence lanes overwrite the executed ones
        if (CT >= 3) w[2 * astride] = 0.0f;
    }
}

// generic (slow) tile store: partial last row tile whose length is not a multiple of 4, or unaligned outputs.
// Takes plain values (a reference to the kernarg struct would force the whole struct into scratch).
__device__ __noinline__ void store_tile_generic(float* pos, float* vel, float* actions, int mask, int B, int T, int D,
                                                int NTW, const float* sSt, int lane, int b0, int rt, int rows) {
    const int SEG = 16 * D, len = rows * D;      // generic path: never shifted, pitch == SEG
    for (int j = 0; j < 3; ++j) {
        if (!((mask >> j) & 1)) continue;
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
// Write-through (sc1) stores.  Default: inline-asm global stores (hipcc does not count them in its vmcnt bookkeeping).
// Round 3 tested the alternative on the suspicion that a later `s_waitcnt vmcnt(N)` for prefetched inputs -- N short by the
// uncounted stores, the queue retiring in order -- makes waves wait for store acknowledgements: (a) the same stores as
// compiler-visible buffer stores (MPK_WT_ASM=0: resource built per store from a wave-uniform base), (b) range-check-
// predicated straight-line stores so that no branch hides them from the count (MPK_WT_PRED=1), (c) the tile-major loop
// re-ordered to collect the next item's inputs before its stores.  Headline launch, alternating builds on one box: asm
// 8.18 us, (a) 8.27 - 8.30, (a + b) 12.1, (a + b + c) 11.2 - 11.4: with seven waves per SIMD the wave that waits is covered
// by the others, while anything that delays or fattens the store issue costs directly.  Kept as build knobs, default off.
#ifndef MPK_WT_PRED
#define MPK_WT_PRED 0            // 1: range-check-predicated straight-line stores in tile_store_sel (A/B build knob)
#endif
#ifndef MPK_WT_ASM
#define MPK_WT_ASM 1             // 0: compiler-visible buffer stores instead of the inline-asm global stores
#endif
#ifndef MPK_STORE_AUX
#define MPK_STORE_AUX 16         // sc1 (build-time knob for A/B runs: 17 = sc0 sc1, 2 = nt, 0 = plain)
#endif
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x2_t __attribute__((ext_vector_type(2)));
struct WtDst { __amdgpu_buffer_rsrc_t rsrc; unsigned off; };
__device__ __forceinline__ WtDst wt_dst(const float* p) {
    const unsigned long long pu = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pu), hi = __builtin_amdgcn_readfirstlane((unsigned)(pu >> 32));
    const unsigned long long base = (((unsigned long long)hi << 32) | lo) - (1ull << 30);
    WtDst d;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, -1, 0x00020000);
    d.off = (unsigned)pu - (unsigned)base;
    return d;
}

template <bool WT>
__device__ __forceinline__ void store16(float* p, const f32x4& v) {
    if (WT && MPK_WT_ASM) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else if (WT) {
        const WtDst d = wt_dst(p);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), d.rsrc, (int)d.off, 0, MPK_STORE_AUX);
    } else {
        *reinterpret_cast<f32x4*>(p) = v;
    }
}

template <bool WT>
__device__ __forceinline__ void store8(float* p, const f32x2& v) {
    if (WT && MPK_WT_ASM) {
        // same cache policy as the 16-byte stores next to it: plain stores into lines that also take write-through
        // stores cost the tile-major kernel half its bandwidth (cfg5 at B = 1024: 14.2 vs 8 us)
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else if (WT) {
        const WtDst d = wt_dst(p);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, v), d.rsrc, (int)d.off, 0, MPK_STORE_AUX);
    } else {
        *reinterpret_cast<f32x2*>(p) = v;
    }
}

template <bool WT>
__device__ __forceinline__ void store4(float* p, float v) {
    if (WT && MPK_WT_ASM) {
        asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else if (WT) {
        const WtDst d = wt_dst(p);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), d.rsrc, (int)d.off, 0, MPK_STORE_AUX);
    } else {
        *p = v;
    }
}

// Write-through store of 16 bytes at `base + off` bytes, PREDICATED by the buffer's range check instead of a branch: a lane
// that must not store passes kWtSkip (beyond num_records = 2 GiB: the hardware discards the store).  Straight-line stores
// are what lets the compiler count them (s_waitcnt vmcnt of a later load wait stays exact); `base` is wave-uniform (an
// output array of the launch -- write-through launches write < 2 GiB per array, enforced by the launchers).
constexpr unsigned kWtSkip = 0xFFFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(const float* base_uniform) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, (int)0x80000000u, 0x00020000);
}
__device__ __forceinline__ void wt_store16(__amdgpu_buffer_rsrc_t r, unsigned off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), r, (int)off, 0, MPK_STORE_AUX);
}

// MASK: which output arrays of the staging image leave (bit 0 pos, bit 1 vel, bit 2 actions)
template <int MASK, int KM, bool WT>
__device__ __forceinline__ void tile_store_sel(const TrajArgs& a, const LaneMap<KM>& L, const float* sSt, int lane,
                                               int b0, int rt, int rows) {
    constexpr bool SP = (MASK & 1) != 0, SV = (MASK & 2) != 0, SA = (MASK & 4) != 0;
    const int D = a.c.D, T = a.c.T, len = rows * D;
    if (a.vec_ok) {
        const int bb = b0 + L.sseg;
        const int lo = (int)ep_shift(a, bb), hi = lo + len, c0 = L.w4;   // valid elements of the padded segment
        const bool in_seg = L.sseg < L.NTW && bb < a.B && c0 < hi && c0 + 4 > lo;
        if (WT && MPK_WT_PRED) {
            // whole chunks: straight-line, range-check-predicated buffer stores (no branch between the wave's loads and
            // its stores: the compiler's vmcnt bookkeeping stays exact, see wt_store16)
            const bool whole = in_seg && c0 >= lo && c0 + 4 <= hi;
            const unsigned off = whole ? (unsigned)((((size_t)bb * T + rt * 16) * D - lo + c0) * sizeof(float)) : kWtSkip;
            const unsigned ro = L.sseg < L.NTW ? L.rofs : 0u;                  // (lanes without a segment read slot 0)
            if (SP) wt_store16(wt_rsrc(a.pos), off, *reinterpret_cast<const f32x4*>(sSt + ro));
            if (SV) wt_store16(wt_rsrc(a.vel), off, *reinterpret_cast<const f32x4*>(sSt + kStageStride + ro));
            if (SA) wt_store16(wt_rsrc(a.actions), off, *reinterpret_cast<const f32x4*>(sSt + 2 * kStageStride + ro));
            if (!a.shifted) return;                                            // T * D % 4 == 0: every chunk is whole
        }
        if (in_seg) {
            const size_t go = ((size_t)bb * T + rt * 16) * D - lo + c0;    // 16-byte aligned by construction
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0;
            if (SP) d0 = *reinterpret_cast<const f32x4*>(sSt + L.rofs);
            if (SV) d1 = *reinterpret_cast<const f32x4*>(sSt + kStageStride + L.rofs);
            if (SA) d2 = *reinterpret_cast<const f32x4*>(sSt + 2 * kStageStride + L.rofs);
            if (c0 >= lo && c0 + 4 <= hi) {
                if (!(WT && MPK_WT_PRED)) {
                    if (SP) store16<WT>(a.pos + go, d0);
                    if (SV) store16<WT>(a.vel + go, d1);
                    if (SA) store16<WT>(a.actions + go, d2);
                }
            } else if (a.td3 == 2) {
                // T*D = 2 mod 4 (e.g. 350 x 7): segment starts and lengths are even, so a partial chunk is exactly its
                // upper half (the chunk straddles the segment start) or its lower half (the end): ONE 8-byte store per
                // array for the head and tail lanes together instead of up to four scalar stores in four branches
                const bool head = c0 < lo;
                const int o = head ? 2 : 0;
                const f32x2 p2 = {head ? d0[2] : d0[0], head ? d0[3] : d0[1]};
                const f32x2 v2 = {head ? d1[2] : d1[0], head ? d1[3] : d1[1]};
                if (SP) store8<WT>(a.pos + go + o, p2);
                if (SV) store8<WT>(a.vel + go + o, v2);
                if (SA) {
                    const f32x2 a2 = {head ? d2[2] : d2[0], head ? d2[3] : d2[1]};
                    store8<WT>(a.actions + go + o, a2);
                }
            } else {                                   // the (at most two) partial chunks of a segment
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c0 + e >= lo && c0 + e < hi) {
                        if (SP) store4<WT>(a.pos + go + e, d0[e]);
                        if (SV) store4<WT>(a.vel + go + e, d1[e]);
                        if (SA) store4<WT>(a.actions + go + e, d2[e]);
                    }
                }
            }
        }
    } else {
        store_tile_generic(a.pos, a.vel, a.actions, MASK, a.B, T, D, L.NTW, sSt, lane, b0, rt, rows);
    }
}

template <int NST, int KM, bool WT>
__device__ __forceinline__ void tile_store(const TrajArgs& a, const LaneMap<KM>& L, const float* sSt, int lane,
                                           int b0, int rt, int rows) {
    tile_store_sel<(NST > 2 ? 7 : 3), KM, WT>(a, L, sSt, lane, b0, rt, rows);
}

// Every kernel-argument field the tile-major prologue needs, demanded in scalar registers at the top of the kernel:
// the compiler then issues ALL their scalar loads in one batch (one scalar-cache miss round trip) instead of where
// each field is first used, which chains two or three dependent misses (~0.2 us each) in front of the first input
// load.  The tile-major kernel lives for ~8 us, so that is worth removing.
__device__ __forceinline__ void demand_args(const TrajArgs& a, unsigned grid_x) {
    asm volatile("" ::"s"(grid_x), "s"(a.c.D), "s"(a.c.nb), "s"(a.c.KT), "s"(a.c.P), "s"(a.c.Kloc), "s"(a.c.off), "s"(a.c.T),
                 "s"(a.c.disable_weights), "s"(a.c.disable_goal), "s"(a.c.goal_off_on), "s"(a.A), "s"(a.aux), "s"(a.TS), "s"(a.params),
                 "s"(a.init_pos), "s"(a.init_vel), "s"(a.pos), "s"(a.vel), "s"(a.actions), "s"(a.c_pos), "s"(a.c_vel),
                 "s"(a.sh), "s"(a.G), "s"(a.vec_ok), "s"(a.pitch), "s"(a.cps), "s"(a.shifted), "s"(a.td3),
                 "s"(a.inv_cps), "s"(a.nrt_magic), "s"(a.gstride));
}

// ---- tile-major ------------------------------------------------------------------------------------------------
// the tile-major body for workgroup `bid` of `nblk` (k_traj_tiles: the whole grid; k_traj_split: the workgroups after the
// serial-role ones)
template <int MP, int CT, int KM, bool WT>
__device__ __forceinline__ void tiles_body(const TrajArgs& a, float* smem, const unsigned bid, const unsigned nblk) {
    static_assert(MP != MPK_MP_DMP, "dmp runs in k_traj_stream");
    static_assert(CT < 3, "closed-loop rollouts run in k_traj_stream / k_traj_split");
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
    // Wn % NRT == 0: this wave owns row tile rt for every item.  wid / NRT by multiply-high with the host's magic
    // number (exact for wid < 2^32 / NRT, which the launcher guarantees): the generic division is ~25 instructions
    // XCD-contiguous virtual block id (workgroup b runs on XCD b % 8): the row tiles of an episode group -- which read
    // the same parameters and write one contiguous trajectory -- stay behind one L2
    const int nb8 = (int)(nblk >> 3);
    const int vb = (nblk & 7) == 0 ? (int)(bid & 7) * nb8 + (int)(bid >> 3) : (int)bid;
    const int wid = vb * 4 + wave;
    const int gstride = a.gstride;
    int g = a.nrt_magic ? (int)__umulhi((unsigned)wid, a.nrt_magic) : wid;      // magic 0: NRT == 1
    const int rt = wid - g * NRT;
    if (g >= a.G) return;
    MPK_STAMP(1);
    // first group's inputs and the controller constants: issued before everything else (latency-bound prologue)
    GroupIn<KM> cur = load_group<MP, ACT, KM>(a, L, g);
    Gains kg{0.0, 0.0, 0.0, 0.0};
    if (ACT) kg = kernarg_gains(L.dvalid ? L.d : 0);
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

    float xb[KM];
    finish_group<KM>(L, cur, xb);
    double cp = cur.cp, cv = cur.cv;
    MPK_STAMP(2);
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
            tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, cp, cv, kg, sSt, L.wofs + ep_shift(a, g * L.NTW + L.bl), D);
        __builtin_amdgcn_wave_barrier();
        MPK_STAMP(10);
        tile_store<NST, KM, WT>(a, L, sSt, lane, g * L.NTW, rt, rows);
        __builtin_amdgcn_wave_barrier();
        MPK_STAMP(11);
        // 5. finish the prefetched fragments for the next iteration
        finish_group<KM>(L, nxt, xb);
        cp = nxt.cp; cv = nxt.cv;
        g = gn;
    }
#ifdef MPK_TRACE
    MPK_STAMP(20);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store of this wave acknowledged
    MPK_STAMP(21);
#endif
}

#ifndef MPK_TILES_OCC
#define MPK_TILES_OCC 7      // waves per SIMD the tile-major kernel is compiled for (A/B build knob)
#endif
template <int MP, int CT, int KM, bool WT>
__global__ void __launch_bounds__(256, (KM <= 2 ? MPK_TILES_OCC : 1)) k_traj_tiles(const TrajArgs a, const ActArgs act) {
    __shared__ __attribute__((aligned(16))) float smem[4 * kStageFloats];
    demand_args(a, gridDim.x);
    tiles_body<MP, CT, KM, WT>(a, smem, blockIdx.x, gridDim.x);
}

// ---- tile-major with a serial role: the fused closed-loop step for cache-resident batches ------------------------------
// BlackBoxWrapper.step on a GPU-resident plant (black_box_wrapper.py:150-217) is serial in t only through the plant
// state: (pos, vel) of every row tile are independent of it, the ACTIONS of the executed steps are not.  So the launch
// has two roles, by workgroup:
//   tiles role   (workgroups >= a.ser_blocks)  exactly k_traj_tiles without a controller: one row tile per wave, pos and
//                vel leave as soon as their tile is contracted -- the parallelism (7 waves per SIMD) that the
//                episode-major closed-loop kernels lack at a few thousand episodes (two waves per SIMD, every LDS / MFMA /
//                barrier latency of 7 sequential row tiles exposed: 19 us at B = 4096, 17-22 us at 8192);
//   serial role  (workgroups <  a.ser_blocks, dispatched first)  a wave owns an episode group: advances the integer
//                replanning state, re-contracts only the row tiles that hold executed steps (a plan that executes 25 of
//                100 steps: 2 of 7), runs the controller + plant recurrence on them (float64, no FMA: the same operations
//                as k_traj_stream, bit for bit), gathers the next boundary condition, and writes the ACTIONS of every tile
//                (zeros past the executed steps) plus the plant state.  Nothing else touches actions or state, so the two
//                roles never race; pos / vel come from the tiles role only.
// The serial role is latency-bound and hides under the store-bound tiles role.
// One lane per (episode, DoF): 64 / DP episodes per wave, every lane busy.  Per step the lane contracts ITS column with the
// step's basis row -- an fp32 fmaf chain in ascending k, i.e. the accumulation order of the MFMA, so the desired state has
// the bits the tiles role stores (the per-episode kernels rely on the same equality) -- and feeds it to the float64
// controller + plant chain.  The basis rows come from the step-major table copy through scalar loads (the row of a step is
// wave-uniform); nothing but the actions passes through LDS.
template <int MP, int CT, int KM, bool WT>
__device__ __forceinline__ void serial_body(const TrajArgs& a, const float* __restrict__ At,
                                            const float* __restrict__ aux, float* smem) {
    static_assert(CT >= 3, "closed loop only");
    constexpr int KP = 4 * KM;
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    constexpr int RS = NOUT * KP;
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = c.D, T = c.T, B = a.B, P = c.P;
    const int DP = 1 << a.sh, EPW = 64 >> a.sh;
    const int e = lane >> a.sh, d = lane & (DP - 1);
    const bool dvalid = d < D;
    float* sAct = smem + wave * (kStageFloats + 16 * RS);   // [EPW][16 * D] action tile of the wave's episodes | rows
    const int NRT = (T + 15) >> 4;
    const int units = (B + EPW - 1) / EPW;
    const int ustride = (int)a.ser_blocks * 4;
    const Gains gn = kernarg_gains(dvalid ? d : 0);
    const double pgd = gn.pg, dgd = gn.dg, lod = __builtin_canonicalize(gn.lo), hid = __builtin_canonicalize(gn.hi),
                 dtp = a.plant_dt;
    // store geometry: an episode's row tile is 16 * D contiguous floats = cps float4 chunks; chunk ids lane + 64 i
    const int cps = a.cps;
    MPK_STAMP(1);
    MPK_STAMP(2);
    for (int u = (int)blockIdx.x * 4 + wave; u < units; u += ustride) {
        const int b = u * EPW + e;
        const bool on = dvalid && b < B;
        const int bs = on ? b : 0, ds = dvalid ? d : 0;
        // every input of the unit is requested before the first one is used (plain loads, no control flow: a branch per
        // column kind made this a chain of eight dependent cache misses -- 9 000 cycles on the trace)
        float raw[KP];
        const float* prm = a.params + (size_t)bs * P + c.off + ds * c.Kloc;
        int kinds[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            int loc;
            kinds[k] = x_kind<MP>(c, k, &loc);           // wave-uniform
            raw[k] = prm[loc];
        }
        const float ipv = a.init_pos[(size_t)bs * D + ds];
        const float ivv = MP == MPK_MP_PRODMP ? a.init_vel[(size_t)bs * D + ds] : 0.0f;
        const size_t si0 = (size_t)bs * D + ds;
        double qs = a.q_state[si0], qds = a.qd_state[si0];
        int nst = 0;
        if (on) {
            nst = T;
            if (a.rp.traj_steps) nst = replan_rule(a.rp, b, T, d == 0);
            else if (a.n_steps) nst = min(a.n_steps[b], T);
        }
        float (&x)[KP] = raw;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const float v = kinds[k] == XK_PARAM ? raw[k] : (kinds[k] == XK_IPOS ? ipv : (kinds[k] == XK_IVEL ? ivv : (kinds[k] == XK_ONE ? 1.0f : 0.0f)));
            x[k] = on ? v : 0.0f;
        }
        const int tcond = (on && a.rp.cond_pos) ? min(max(nst - 1, 0), T - 1) : -1;
        int need = on ? max(nst, tcond + 1) : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) need = max(need, __shfl_xor(need, m));
        const int n_ser = __builtin_amdgcn_readfirstlane((need + 15) >> 4);      // row tiles that hold an executed step
        float cpos = 0.0f, cvel = 0.0f;
        // action stores: float4 chunk ids lane + 64 i -> (episode of the unit, offset in its 16 * D tile segment)
        float* const ub = a.actions + (size_t)u * EPW * T * D;
        unsigned sgo[4], slds[4], sw4[4];
        bool sval[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = lane + 64 * i;
            const int sg = (int)(((unsigned)ch * a.inv_cps) >> 16);
            const int w4 = (ch - sg * cps) * 4;
            sval[i] = sg < EPW && u * EPW + sg < B;
            sw4[i] = (unsigned)w4;
            sgo[i] = (unsigned)(sg * T * D + w4);
            slds[i] = (unsigned)(sg * 16 * D + w4);
        }
        MPK_STAMP(3);
        // lanes of a padding DoF (d >= D) park their action in the spare floats behind the image (there are >= 64 of them
        // whenever D < DP): one address select per unit instead of an exec-mask branch per step
        float* const slot = sAct + (dvalid ? e * (16 * D) + d : EPW * 16 * D);
        const int sstep = dvalid ? D : 0;
        // basis rows: the 16 step-major rows of a row tile are 16 * RS contiguous floats of At -- one coalesced float4 load
        // per lane (two for promp), parked in the wave's LDS slice one tile ahead; a step reads its row with broadcast LDS
        // reads, one step ahead.  (Scalar loads of the rows -- a wave-uniform address through the constant address space
        // -- measured 380 cycles per step even with a warm scalar cache: profiles/r02_closed_loop.md.)
        float* const sRow = sAct + kStageFloats;                       // [16][RS]
        constexpr int NR4 = 16 * RS / 4;                               // float4 per row tile
        constexpr int NRR = (NR4 + 63) / 64;                           // float4 per lane (promp with 12+ columns: 3)
        f32x4 rr[NRR];
#pragma unroll
        for (int i = 0; i < NRR; ++i) rr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto fetch_rows = [&](int rt) {
            const f32x4* src = reinterpret_cast<const f32x4*>(At + (size_t)rt * 16 * RS);
#pragma unroll
            for (int i = 0; i < NRR; ++i)
                if (lane + 64 * i < NR4) rr[i] = src[lane + 64 * i];
        };
        auto park_rows = [&]() {
#pragma unroll
            for (int i = 0; i < NRR; ++i)
                if (lane + 64 * i < NR4) reinterpret_cast<f32x4*>(sRow)[lane + 64 * i] = rr[i];
        };
        if (n_ser > 0) fetch_rows(0);
        for (int rt = 0; rt < NRT; ++rt) {
            const int rows = min(16, T - rt * 16);
            if (rt < n_ser) {
                park_rows();
                __builtin_amdgcn_wave_barrier();
                if (rt + 1 < n_ser) fetch_rows(rt + 1);                // in flight under this tile's 16 steps
                // a row of the tile from LDS (broadcast reads)
                auto read_row = [&](int tl, float (&r)[RS]) {
#pragma unroll
                    for (int k4 = 0; k4 < RS / 4; ++k4) {
                        const f32x4 q4 = reinterpret_cast<const f32x4*>(sRow + tl * RS)[k4];
                        r[4 * k4] = q4[0]; r[4 * k4 + 1] = q4[1]; r[4 * k4 + 2] = q4[2]; r[4 * k4 + 3] = q4[3];
                    }
                };
                // one step: the lane's column against the step's row (fp32 fmaf chains in ascending k = the MFMA's
                // accumulation order), then the float64 controller + plant chain.  MASKED: steps past the executed ones and
                // the gathered step are handled by selects; the unmasked form serves a tile every lane executes in full
                auto one_step = [&](auto masked_tag, int tl, const float (&rc)[RS]) {
                    constexpr bool MASKED = decltype(masked_tag)::value;
                    const int t = rt * 16 + tl;
                    float p = 0.0f, v = 0.0f;
                    if (MP == MPK_MP_PRODMP) {
                        // rows interleaved (pos_k, vel_k): both chains in packed FMAs; 1/tau is folded into the vel rows
#pragma unroll
                        for (int k = 0; k < KP; ++k) { p = fmaf(rc[2 * k], x[k], p); v = fmaf(rc[2 * k + 1], x[k], v); }
                    } else {
                        float ph = 0.0f, pl = 0.0f;
#pragma unroll
                        for (int k = 0; k < KP; ++k) {
                            p = fmaf(rc[k], x[k], p);
                            ph = fmaf(rc[KP + k], x[k], ph); pl = fmaf(rc[2 * KP + k], x[k], pl);
                        }
                        v = (ph - pl) * aux[t];                  // forward difference of fp32 positions x (1 / dt)
                    }
                    if (MASKED) {
                        const bool at_cond = t == tcond;
                        cpos = at_cond ? p : cpos; cvel = at_cond ? v : cvel;
                    }
                    const double dp = (double)p, dv = (double)v;
                    double uu;
                    if (CT - 3 == MPK_CTRL_MOTOR) uu = pgd * (dp - qs) + dgd * (dv - qds);
                    else if (CT - 3 == MPK_CTRL_POSITION) uu = dp;
                    else uu = dv;
                    uu = clip_f64(uu, lod, hid);
                    const double qds_n = qds + dtp * uu;
                    const double qs_n = qs + dtp * qds_n;
                    if (MASKED) {
                        const bool live = t < nst;
                        qds = live ? qds_n : qds;
                        qs = live ? qs_n : qs;
                        slot[tl * sstep] = live ? (float)uu : 0.0f;
                    } else {
                        qds = qds_n; qs = qs_n;
                        slot[tl * sstep] = (float)uu;
                    }
                };
                // two row buffers in turn (no copies): the row of step tl + 1 is requested before step tl is computed
                auto tile_steps = [&](auto masked_tag) {
                    float r0[RS], r1[RS];
                    read_row(0, r0);
#pragma unroll 1
                    for (int tl = 0; tl < 16; tl += 2) {
                        read_row(tl + 1, r1);
                        one_step(masked_tag, tl, r0);
                        read_row(tl + 2 < 16 ? tl + 2 : 15, r0);
                        one_step(masked_tag, tl + 1, r1);
                    }
                };
                // every lane executes every step of the tile and none gathers its boundary condition here? (wave-uniform)
                const bool plain = __all(!on || (nst >= rt * 16 + 16 && (tcond < rt * 16 || tcond >= rt * 16 + 16))) != 0;
                if (plain) tile_steps(std::false_type());
                else tile_steps(std::true_type());
                __builtin_amdgcn_wave_barrier();
                MPK_STAMP(10 + rt);
            }
            // the wave's EPW action segments of this row tile: coalesced float4 stores (zeros past the executed tiles); the
            // lane's chunk geometry was worked out once per unit
            {
                float* const tb = ub + (size_t)rt * 16 * D;
                const int lim = rows * D;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (sval[i] && (int)sw4[i] < lim) {
                        f32x4 val = {0.f, 0.f, 0.f, 0.f};
                        if (rt < n_ser) val = *reinterpret_cast<const f32x4*>(sAct + slds[i]);
                        store16<WT>(tb + sgo[i], val);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            MPK_STAMP(30 + rt);
        }
        if (on) {
            const size_t si = (size_t)b * D + d;
            a.q_state[si] = qs; a.qd_state[si] = qds;
            if (tcond >= 0) { a.rp.cond_pos[si] = cpos; a.rp.cond_vel[si] = cvel; }
        }
        MPK_STAMP(90);
    }
}

#ifndef MPK_SPLIT_OCC
// waves per SIMD the split kernel is compiled for.  The tile-major body does not need more: with dynamic-LDS padding
// capping the workgroups per CU it runs 9.3 / 9.2 / 9.1 / 9.8 / 9.6 us at 8 / 7 / 6 / 5 / 4 waves per SIMD (B = 4096,
// tools/occ_probe.py), and 128 registers let the serial role keep its rows, columns and float64 state without scratch.
#define MPK_SPLIT_OCC 4
#endif
template <int MP, int CT, int KM, bool WT>
__global__ void __launch_bounds__(256, (KM <= 2 ? MPK_SPLIT_OCC : 1)) k_traj_split(const TrajArgs a, const ActArgs act) {
    constexpr int kWaveFloats = kStageFloats + 16 * (MP == MPK_MP_PRODMP ? 2 : 3) * 4 * KM;   // staging + one tile of rows
    __shared__ __attribute__((aligned(16))) float smem[4 * kWaveFloats];
    demand_args(a, gridDim.x);
    if (blockIdx.x < a.ser_blocks) serial_body<MP, CT, KM, WT>(a, a.A + (size_t)(MP == MPK_MP_PRODMP ? 2 : 3) * (4 * KM) * a.TS, a.aux, smem);
    else tiles_body<MP, -1, KM, WT>(a, smem, blockIdx.x - a.ser_blocks, gridDim.x - a.ser_blocks);
}

// 16 consecutive floats at a wave-uniform, 16-byte aligned LDS address (the scaled-time steps of a row tile)
__device__ __forceinline__ void load_ds16(const float* __restrict__ p, float (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 x = reinterpret_cast<const float4*>(p)[j];
        v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
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
    // step whose desired state is gathered for the next plan's boundary condition (k_condition_gather's clamp); -1 = off
    const int tcond = (CLOSED && a.rp.cond_pos) ? min(max(nst - 1, 0), T - 1) : -1;
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
            if (L.dvalid) {
                Gains gn{0.0, 0.0, 0.0, 0.0};
                if (CT >= 0 && CT < 3) gn = parked_gains(sg);
                tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, cp, cv, gn, sSt, wofs, D);
            }
            if (CLOSED) {
                // the step loop of black_box_wrapper.py:175-203 on the reference's torque double integrator
                // (base_reacher_torque.py:25-26), serial in t on the lanes (q == 0); float64, no FMA
                __builtin_amdgcn_wave_barrier();
                const bool full_tile = tile_fully_executed(serial, nst, rt * 16);
                // row tiles past the executed steps (and past the gathered step) have nothing serial to do: a replanning
                // plan that executes 25 of its 100 steps runs the recurrence on 2 of 7 tiles
                if (serial && rt * 16 < max(nst, tcond + 1)) {
                    // canonical once: fmin / fmax otherwise quiet their bound operands again at every step
                    const double pgd = sg[0], dgd = sg[16], lod = __builtin_canonicalize(sg[32]),
                                 hid = __builtin_canonicalize(sg[48]), dtp = a.plant_dt;
                    if (tcond >= rt * 16 && tcond < rt * 16 + 16) {   // condition_on_desired: the desired state at the
                        const size_t si = (size_t)(b0 + L.bl) * D + L.d;     // last executed step
                        a.rp.cond_pos[si] = sSt[o0 + (tcond - rt * 16) * D];
                        a.rp.cond_vel[si] = sSt[kStageStride + o0 + (tcond - rt * 16) * D];
                    }
                    if (full_tile)
                        pd_tile_steps<CT - 3, false>(sSt + o0, sSt + kStageStride + o0, sSt + 2 * kStageStride + o0, D,
                                                     rt * 16, nst, pgd, dgd, lod, hid, dtp, qs, qds);
                    else
                        pd_tile_steps<CT - 3, true>(sSt + o0, sSt + kStageStride + o0, sSt + 2 * kStageStride + o0, D,
                                                    rt * 16, nst, pgd, dgd, lod, hid, dtp, qs, qds);
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
            if (eul)
                dmp_tile_steps(sF + o0, sSt + o0, sSt + kStageStride + o0, sAux + rt * 16, D, rt * 16, T, c.dmp_alpha,
                               c.dmp_beta, eg, make_tau_div(c.tau), ey, ez);
            // (vel = z / tau is written by the recurrence lanes themselves)
        }
        __builtin_amdgcn_wave_barrier();
        if (a.wt) tile_store<NST, KM, true>(a, L, sSt, lane, b0, rt, rows);      // cache-resident outputs (wave-uniform)
        else tile_store<NST, KM, false>(a, L, sSt, lane, b0, rt, rows);
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
    stage_tables(a.A, a.aux, sA, sAux, (NOUT * KP * TS) >> 2, TS >> 2, threadIdx.x);   // once per workgroup
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
                    if (a.rp.traj_steps) nst = replan_rule(a.rp, b0 + L.bl, c.T, L.d == 0);
                    else if (a.n_steps) nst = min(a.n_steps[b0 + L.bl], c.T);
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
                    xb[m] = L.isp[m] ? raw : (L.isip[m] ? ip : (L.isiv[m] ? iv : L.cst[m]));
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
                        if (a.rp.traj_steps) nst = replan_rule(a.rp, b0 + L.bl, c.T, L.d == 0);
                        else if (a.n_steps) nst = min(a.n_steps[b0 + L.bl], c.T);
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

// ---- episode-major with WHOLE-TRAJECTORY images: k_traj_flat (round 3) ---------------------------------------------------
// The HBM-streaming case of the open-loop step (promp / prodmp, trajectory [+ actions], outputs far beyond the caches).
// Measured on the streaming row (profiles/r03_streaming.md): the launch is bound by how the CU's store path is fed, not by
// DRAM (per-channel write requests uniform, 5 % credit stalls) -- FEWER resident workgroups are faster (12 -> 8 waves per
// CU: 542 -> 500 us) and longer contiguous runs per store instruction are faster (profiles/r01_store_patterns.md: 448-byte
// pieces 4.9 TB/s, whole episodes 5.1, 44.8 KB runs 5.6).  So here a wave contracts ALL row tiles of its episode group
// into an LDS image of whole trajectories [pos | vel | act][NTW episodes][T * D] (no stores, no barriers in between),
// then streams each episode's T * D floats out as full-width float4 stores -- 1 KB contiguous per instruction, 2.8 KB
// per episode and array, neighbouring waves writing neighbouring episodes -- while the inputs of the next group,
// requested BEFORE the flush entered the in-order memory queue, are already on their way.  Two 4-wave workgroups per CU.
// Same tile arithmetic as k_traj_stream (same functions): same bits.
template <int MP, int CT, int KM>
__global__ void __launch_bounds__(256, 2) k_traj_flat(const TrajArgs a, const ActArgs act) {
    static_assert(MP != MPK_MP_DMP && CT < 3, "open loop, promp / prodmp");
    extern __shared__ __attribute__((aligned(16))) float sTab[];   // [NOUT][KP][TS] rows + [TS] aux + 4 x image
    constexpr bool ACT = CT >= 0;
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    constexpr int NST = 2 + (ACT ? 1 : 0);
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = 4 * KM, TS = a.TS, D = c.D, T = c.T, TD = T * D;
    (void)act;
    float* sA = sTab;
    float* sAux = sTab + NOUT * KP * TS;
    stage_tables(a.A, a.aux, sA, sAux, (NOUT * KP * TS) >> 2, TS >> 2, threadIdx.x);   // once per workgroup
    const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
    const int NTW = L.NTW, NRT = (T + 15) >> 4;
    const int IMG = a.flat_img;                                   // floats per array image: NTW * T * D rounded up to 4
    float* sI = sAux + TS + wave * (NST * IMG);
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int wstride = gridDim.x * 4;
    int g = vb * 4 + wave;
    const bool active = g < a.G;
    GroupIn<KM> cur;
    if (active) cur = load_group<MP, ACT, KM>(a, L, g);
    Gains gn{0.0, 0.0, 0.0, 0.0};
    if (ACT) gn = kernarg_gains(L.dvalid ? L.d : 0);
    __syncthreads();                                              // the tables are in LDS
    if (!active) return;
    const float* ap = sA + L.q * TS + L.col;
    const unsigned wbase = (unsigned)(L.bl * TD + 4 * L.q * D + L.d);   // (episode, row 4q, column) inside an image
    float xb[KM];
    finish_group<KM>(L, cur, xb);
    double cp = cur.cp, cv = cur.cv;
    const int TD4 = TD >> 2;
    while (g < a.G) {
        const int b0 = g * NTW;
        const int gn_ = g + wstride;
        const GroupIn<KM> nxt = load_group<MP, ACT, KM>(a, L, gn_ < a.G ? gn_ : g);   // in flight across the whole group
        for (int rt = 0; rt < NRT; ++rt) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < KM; ++m) {
                const float* am = ap + (4 * m) * TS + rt * 16;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[0], xb[m], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[KP * TS], xb[m], acc1, 0, 0, 0);
                if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[(NOUT > 2 ? 2 : 0) * KP * TS], xb[m], acc2, 0, 0, 0);
            }
            float dtd[4] = {1.f, 1.f, 1.f, 1.f};
            if (MP == MPK_MP_PROMP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dtd[r] = sAux[rt * 16 + 4 * L.q + r];
            }
            const int nrows = min(4, T - rt * 16 - 4 * L.q);      // rows of this lane that exist (<= 0: none)
            if (L.dvalid && nrows > 0)
                tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, cp, cv, gn, sI, wbase + (unsigned)(rt * 16 * D), D, IMG, nrows);
        }
        __builtin_amdgcn_wave_barrier();
        // flush: each episode's T * D floats of each array are one contiguous, 16-byte aligned run in HBM
        for (int e = 0; e < NTW; ++e) {
            const int bb = b0 + e;
            if (bb >= a.B) break;
            const size_t go = (size_t)bb * TD;
            const float* se = sI + e * TD;
#ifndef MPK_FLAT_ARRAY_MAJOR
            for (int i = lane; i < TD4; i += 64) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(se + 4 * i);
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(se + IMG + 4 * i);
                if (a.wt) { store16<true>(a.pos + go + 4 * i, p4); store16<true>(a.vel + go + 4 * i, v4); }
                else { store16<false>(a.pos + go + 4 * i, p4); store16<false>(a.vel + go + 4 * i, v4); }
                if (ACT) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(se + 2 * IMG + 4 * i);
                    if (a.wt) store16<true>(a.actions + go + 4 * i, a4); else store16<false>(a.actions + go + 4 * i, a4);
                }
            }
#else
            // array by array: the episode's 2.8 KB of one array leave back to back before the next array starts (A/B build:
            // 421 vs 406 - 411 us at B = 262144 on a fast box, equal on a slow one -- interleaved is the default)
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                float* const outp = (j == 0 ? a.pos : (j == 1 ? a.vel : a.actions)) + go;
                const float* sj = se + j * IMG;
                for (int i = lane; i < TD4; i += 64) {
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(sj + 4 * i);
                    if (a.wt) store16<true>(outp + 4 * i, x4); else store16<false>(outp + 4 * i, x4);
                }
            }
#endif
        }
        __builtin_amdgcn_wave_barrier();                          // the image is free again
        finish_group<KM>(L, nxt, xb);
        cp = nxt.cp; cv = nxt.cv;
        g = gn_;
    }
}

// ---- episode-major, four groups per wave: the serial-recurrence variants ---------------------------------------------
// DMP (Euler recurrence) and the closed-loop rollout (controller + plant recurrence) are serial in t and run on the
// 16 lanes that hold row 0 of a column.  Here a wave owns FOUR consecutive episode groups at once: per row tile it
// produces the four C tiles back to back on the matrix cores, then lane quarter q runs group q's recurrence, so the four
// recurrences advance in parallel (4x fewer serial instructions per episode), then the four tiles leave as coalesced
// float4 stores.  Same arithmetic and bits as k_traj_stream.
constexpr int kQuad = 4;
// floats per group image (pos | vel | act or force): 8 floats past a multiple of the 32 LDS banks, so that the four lane
// quarters -- which walk the four images with the same in-image offsets during the recurrences -- fall on disjoint banks
// (measured before the skew: 43 % of the kernel's LDS cycles were bank conflicts)
constexpr int kQuadImg = 3 * kStageStride + 8;

template <int MP, int CT, int KM, int NQ>
__global__ void __launch_bounds__(256) k_traj_quad(const TrajArgs a, const ActArgs act) {
    static_assert(NQ == 1 || NQ == 2 || NQ == 4, "one, two or four groups per wave");
    __shared__ __attribute__((aligned(16))) float smem[4 * NQ * kQuadImg];   // per wave: 4 x (pos|vel|act or force)
    extern __shared__ __attribute__((aligned(16))) float sTab[];   // [NOUT][KP][TS] rows + [TS] aux
    constexpr bool CLOSED = CT >= 3;

    static_assert(MP == MPK_MP_DMP || CLOSED, "k_traj_quad is for the serial-recurrence variants");
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : (MP == MPK_MP_PROMP ? 3 : 1);
    constexpr int NST = CLOSED ? 3 : 2;
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = 4 * KM, TS = a.TS, D = c.D, B = a.B, P = c.P, T = c.T;
    float* sW = smem + wave * (NQ * kQuadImg);
    float* sA = sTab;
    float* sAux = sTab + NOUT * KP * TS;
    const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
    const int NTW = L.NTW, NRT = (T + 15) >> 4;
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int ustride = gridDim.x * 4;
    const int NU = (a.G + NQ - 1) / NQ;
    int u = vb * 4 + wave;

    // A lane's serial-recurrence inputs for one unit: group u * 4 + q, column (bl, d).  Fetched one unit ahead, like the
    // B-fragment inputs (the integer replanning state of the episode is advanced at fetch time by its d == 0 lane).
    struct SerialIn { double qs, qds; int nst; float ey, ez, eg; bool on; };
    auto load_serial = [&](int uu) {
        SerialIn si{0.0, 0.0, T, 0.f, 0.f, 0.f, false};
        const int gq = uu * NQ + L.q, bq = gq * NTW + L.bl;
        si.on = L.dvalid && L.q < NQ && gq < a.G && bq < B;
        if (si.on) {
            const size_t ix = (size_t)bq * D + L.d;
            if (CLOSED) {
                si.qs = a.q_state[ix]; si.qds = a.qd_state[ix];
                if (a.rp.traj_steps) si.nst = replan_rule(a.rp, bq, T, L.d == 0);
                else if (a.n_steps) si.nst = min(a.n_steps[bq], T);
            } else {
                si.ey = a.init_pos[ix];
                si.ez = a.init_vel[ix] * c.tau;
                si.eg = a.params[(size_t)bq * P + c.off + L.d * c.Kloc + c.nb] * c.gs;
            }
        }
        return si;
    };
    // the first unit's inputs are in flight while the workgroup stages the basis tables
    GroupIn<KM> nx[NQ];
    SerialIn sn{0.0, 0.0, T, 0.f, 0.f, 0.f, false};
    MPK_STAMP(1);
    if (u < NU) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int g = u * NQ + j;
            nx[j] = load_group<MP, false, KM>(a, L, g < a.G ? g : a.G - 1);
        }
        sn = load_serial(u);
    }
    stage_tables(a.A, a.aux, sA, sAux, (NOUT * KP * TS) >> 2, TS >> 2, threadIdx.x);
    __syncthreads();
    MPK_STAMP(2);
    if (u >= NU) return;
    const float* ap = sA + L.q * TS + L.col;
    const TauDiv td = make_tau_div(c.tau);
    double pgd = 0.0, dgd = 0.0, lod = 0.0, hid = 0.0;
    if (CLOSED) {
        // four vector loads from the kernel-argument segment (see kernarg_gains) instead of 64 exec-masked selects
        const Gains gq = kernarg_gains(L.dvalid ? L.d : 0);
        pgd = gq.pg; dgd = gq.dg;
        lod = __builtin_canonicalize(gq.lo); hid = __builtin_canonicalize(gq.hi);   // not again at every step's fmin / fmax
    }
    (void)act;

    float xb[NQ][KM];
    while (u < NU) {
        const int g0 = u * NQ;
#pragma unroll
        for (int j = 0; j < NQ; ++j) finish_group<KM>(L, nx[j], xb[j]);
        const SerialIn sc = sn;
        MPK_STAMP(3);
        const int un = u + ustride;
        if (un < NU) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int g = un * NQ + j;
                nx[j] = load_group<MP, false, KM>(a, L, g < a.G ? g : a.G - 1);
            }
            sn = load_serial(un);
        }
        // this lane's recurrence: group g0 + q, column (bl, d)
        const int gq = g0 + L.q, bq = gq * NTW + L.bl;
        const bool serial = sc.on;
        const int oq = L.bl * a.pitch + L.d + (int)ep_shift(a, bq);      // (row 0, this column) in group q's image
        float* sQ = sW + L.q * kQuadImg;
        double qs = sc.qs, qds = sc.qds;
        const int nst = sc.nst;
        float ey = sc.ey, ez = sc.ez, eg = sc.eg;
        const int tcond = (CLOSED && a.rp.cond_pos) ? min(max(nst - 1, 0), T - 1) : -1;
        // A fragments (basis rows of a row tile) are the same for the four groups: read from LDS once per tile, one tile
        // ahead, into registers.  Left to the compiler they are re-read in front of every MFMA (it cannot prove that
        // the staging writes do not alias the tables), and with one or two waves per SIMD each of those LDS round trips
        // is exposed.
        float afn[NOUT][KM];
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int m = 0; m < KM; ++m) afn[o][m] = ap[(o * KP + 4 * m) * TS];
        for (int rt = 0; rt < NRT; ++rt) {
            const int rows = min(16, T - rt * 16);
            float af[NOUT][KM];
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int m = 0; m < KM; ++m) af[o][m] = afn[o][m];
            if (rt + 1 < NRT) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
#pragma unroll
                    for (int m = 0; m < KM; ++m) afn[o][m] = ap[(o * KP + 4 * m) * TS + (rt + 1) * 16];
            }
            // 1. four C tiles on the matrix cores -> four staging images
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (g0 + j < a.G) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < KM; ++m) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][m], xb[j][m], acc0, 0, 0, 0);
                        if (NOUT > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[NOUT > 1 ? 1 : 0][m], xb[j][m], acc1, 0, 0, 0);
                        if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[NOUT > 2 ? 2 : 0][m], xb[j][m], acc2, 0, 0, 0);
                    }
                    float* sJ = sW + j * kQuadImg;
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
                            tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, 0.0, 0.0, Gains{0.0, 0.0, 0.0, 0.0}, sJ, wofs, D);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            MPK_STAMP(10 + rt);
            // 2. four recurrences in parallel, one per lane quarter (float64 / fp32 without FMA, as k_traj_stream)
            const bool full_tile = CLOSED && tile_fully_executed(serial, nst, rt * 16);
            if (serial && (!CLOSED || rt * 16 < max(nst, tcond + 1))) {
                if (CLOSED) {
                    if (tcond >= rt * 16 && tcond < rt * 16 + 16) {   // condition_on_desired: the desired state at the
                        const size_t si = (size_t)bq * D + L.d;          // last executed step
                        a.rp.cond_pos[si] = sQ[oq + (tcond - rt * 16) * D];
                        a.rp.cond_vel[si] = sQ[kStageStride + oq + (tcond - rt * 16) * D];
                    }
                    if (full_tile)
                        pd_tile_steps<(CLOSED ? CT - 3 : 0), false>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D,
                                                                    rt * 16, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds);
                    else
                        pd_tile_steps<(CLOSED ? CT - 3 : 0), true>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D,
                                                                   rt * 16, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds);
                } else {
                    dmp_tile_steps(sQ + 2 * kStageStride + oq, sQ + oq, sQ + kStageStride + oq, sAux + rt * 16, D, rt * 16, T,
                                   c.dmp_alpha, c.dmp_beta, eg, td, ey, ez);
                }
            }
            __builtin_amdgcn_wave_barrier();
            MPK_STAMP(30 + rt);
            // 3. coalesced stores of the four tiles
#pragma unroll
            for (int j = 0; j < NQ; ++j)
                if (g0 + j < a.G)
                {
                    if (a.wt) tile_store<NST, KM, true>(a, L, sW + j * kQuadImg, lane, (g0 + j) * NTW, rt, rows);
                    else tile_store<NST, KM, false>(a, L, sW + j * kQuadImg, lane, (g0 + j) * NTW, rt, rows);
                }
            __builtin_amdgcn_wave_barrier();
            MPK_STAMP(50 + rt);
        }
        if (CLOSED) {
            if (serial) {
                const size_t si = (size_t)bq * D + L.d;
                a.q_state[si] = qs; a.qd_state[si] = qds;
            }
        }
        MPK_STAMP(90);
        u = un;
    }
}

// ---- episode-major, wave-specialised: the closed-loop step as a producer / consumer pipeline -------------------------
// What bounds k_traj_quad / k_traj_stream<closed> at a few thousand episodes is not arithmetic but ONE wave doing
// everything in sequence, 7 row tiles x (contract -> LDS -> recurrence -> LDS -> store), every latency exposed (PMC at
// B = 4096, profiles/r02_closed_loop.md: 2 590 VALU + 408 LDS + 564 scalar instructions per wave, 42 % of the wave's
// cycles in s_waitcnt).  Here a workgroup of FIVE waves owns four consecutive episode groups:
//   waves 1..4  (producers)  contract row tile rt + 1 of "their" group on the matrix cores into LDS image (rt + 1) & 1
//               and store tile rt (pos, vel, actions) from image rt & 1;
//   wave 0      (consumer)   runs the controller + plant recurrence of tile rt for all four groups at once, one group per
//               lane quarter (float64, no FMA: pd_tile_steps, the operations of every other closed-loop kernel, bit for
//               bit), while the producers are busy with tile rt + 1 and with the stores of tile rt - 1.
// One workgroup barrier per row tile hands the images over.  The integer replanning state, the boundary-condition gather
// and the plant state are the consumer's, exactly as in k_traj_quad.
constexpr int kPipeGroups = 4;

template <int MP, int CT, int KM>
__global__ void __launch_bounds__(320) k_traj_pipe(const TrajArgs a, const ActArgs act) {
    static_assert(CT >= 3 && MP != MPK_MP_DMP, "closed loop, promp / prodmp");
    __shared__ __attribute__((aligned(16))) float smem[2 * kPipeGroups * kQuadImg];   // [buffer][group] pos | vel | act
    extern __shared__ __attribute__((aligned(16))) float sTab[];                      // [NOUT][KP][TS] rows + [TS] aux
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = 4 * KM, TS = a.TS, D = c.D, B = a.B, T = c.T;
    float* sA = sTab;
    float* sAux = sTab + NOUT * KP * TS;
    const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
    const int NTW = L.NTW, NRT = (T + 15) >> 4;
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int NU = (a.G + kPipeGroups - 1) / kPipeGroups;
    (void)act;
    MPK_STAMP_AT(1, 0); MPK_STAMP_AT(101, 64);
    // head of the critical path: the producers' first inputs and the basis rows of row tile 0 are requested before the
    // table copy (tile 0 is contracted from registers while the LDS copy lands; later tiles read the copy)
    GroupIn<KM> nx;
    float a0[NOUT][KM];
    if (wave != 0) {
        if (vb < NU) {
            const int g = vb * kPipeGroups + wave - 1;
            nx = load_group<MP, false, KM>(a, L, g < a.G ? g : a.G - 1);
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int m = 0; m < KM; ++m) a0[o][m] = a.A[(o * KP + 4 * m + L.q) * TS + L.col];
    }
    // basis tables -> LDS by all five waves (a 256-thread loop shape: threads 256.. take the tail)
    {
        const float4* src = reinterpret_cast<const float4*>(a.A);
        float4* dst = reinterpret_cast<float4*>(sA);
        const int nA4 = (NOUT * KP * TS) >> 2, nX4 = TS >> 2;
        for (int i = threadIdx.x; i < nA4; i += 320) dst[i] = src[i];
        for (int i = threadIdx.x; i < nX4; i += 320) reinterpret_cast<float4*>(sAux)[i] = reinterpret_cast<const float4*>(a.aux)[i];
    }
    if (wave == 0) {
        __syncthreads();                                                    // (the table copy: the producers' barrier)
        // ---------------- consumer: four recurrences, one per lane quarter ----------------
        const Gains gq = kernarg_gains(L.dvalid ? L.d : 0);
        const double pgd = gq.pg, dgd = gq.dg, lod = __builtin_canonicalize(gq.lo), hid = __builtin_canonicalize(gq.hi);
        for (int u = vb; u < NU; u += (int)gridDim.x) {
            const int gsel = u * kPipeGroups + L.q, bq = gsel * NTW + L.bl;
            const bool serial = L.dvalid && gsel < a.G && bq < B;
            double qs = 0.0, qds = 0.0;
            int nst = 0;
            if (serial) {
                const size_t ix = (size_t)bq * D + L.d;
                qs = a.q_state[ix]; qds = a.qd_state[ix];
                nst = T;
                if (a.rp.traj_steps) nst = replan_rule(a.rp, bq, T, L.d == 0);
                else if (a.n_steps) nst = min(a.n_steps[bq], T);
            }
            const int tcond = (serial && a.rp.cond_pos) ? min(max(nst - 1, 0), T - 1) : -1;
            const int oq = L.bl * a.pitch + L.d + (int)ep_shift(a, bq);      // (row 0, this column) in group q's image
            __syncthreads();                                                // tile 0 is in image 0
            for (int rt = 0; rt < NRT; ++rt) {
                float* sQ = smem + ((rt & 1) * kPipeGroups + L.q) * kQuadImg;
                const bool full_tile = tile_fully_executed(serial, nst, rt * 16);
                if (serial && rt * 16 < max(nst, tcond + 1)) {
                    if (tcond >= rt * 16 && tcond < rt * 16 + 16) {   // condition_on_desired: the desired state at the
                        const size_t si = (size_t)bq * D + L.d;          // last executed step
                        a.rp.cond_pos[si] = sQ[oq + (tcond - rt * 16) * D];
                        a.rp.cond_vel[si] = sQ[kStageStride + oq + (tcond - rt * 16) * D];
                    }
                    if (full_tile)
                        pd_tile_steps<CT - 3, false>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D, rt * 16, nst,
                                                     pgd, dgd, lod, hid, a.plant_dt, qs, qds);
                    else
                        pd_tile_steps<CT - 3, true>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D, rt * 16, nst,
                                                    pgd, dgd, lod, hid, a.plant_dt, qs, qds);
                }
                __syncthreads();                                            // tile rt's actions are final; tile rt + 1 is in
            }
            if (serial) {
                const size_t si = (size_t)bq * D + L.d;
                a.q_state[si] = qs; a.qd_state[si] = qds;
            }
        }
    } else {
        // ---------------- producers: wave j + 1 owns group u * 4 + j ----------------
        const int j = wave - 1;
        const float* ap = sA + L.q * TS + L.col;
        int u = vb;
        for (; u < NU; u += (int)gridDim.x) {
            const int g = u * kPipeGroups + j;
            const bool have = g < a.G;
            float xb[KM];
            finish_group<KM>(L, nx, xb);
            const int un = u + (int)gridDim.x;
            if (un < NU) {
                const int gn = un * kPipeGroups + j;
                nx = load_group<MP, false, KM>(a, L, gn < a.G ? gn : a.G - 1);
            }
            const unsigned wofs = L.wofs + ep_shift(a, g * NTW + L.bl);
            auto produce = [&](int rt, auto first_tag) {
                constexpr bool FIRST = decltype(first_tag)::value;          // rows of tile 0 of the first unit: registers
                float* sJ = smem + ((rt & 1) * kPipeGroups + j) * kQuadImg;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < KM; ++m) {
                    const float* am = ap + (4 * m) * TS + rt * 16;
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(FIRST ? a0[0][m] : am[0], xb[m], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(FIRST ? a0[1][m] : am[KP * TS], xb[m], acc1, 0, 0, 0);
                    if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(FIRST ? a0[NOUT > 2 ? 2 : 0][m] : am[(NOUT > 2 ? 2 : 0) * KP * TS], xb[m], acc2, 0, 0, 0);
                }
                float dtd[4] = {1.f, 1.f, 1.f, 1.f};
                if (MP == MPK_MP_PROMP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dtd[r] = FIRST ? a.aux[4 * L.q + r] : sAux[rt * 16 + 4 * L.q + r];
                }
                if (L.dvalid) tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, 0.0, 0.0, Gains{0.0, 0.0, 0.0, 0.0}, sJ, wofs, D);
            };
            // pos and vel of a tile leave as soon as it is contracted (nothing of theirs waits for the consumer: two thirds
            // of the store stream are independent of the recurrence); the actions follow after the barrier
            auto store_arrays = [&](auto mask_tag, int rt) {
                constexpr int MASK = decltype(mask_tag)::value;
                const float* sJ = smem + ((rt & 1) * kPipeGroups + j) * kQuadImg;
                const int rows = min(16, T - rt * 16);
                if (a.wt) tile_store_sel<MASK, KM, true>(a, L, sJ, lane, g * NTW, rt, rows);
                else tile_store_sel<MASK, KM, false>(a, L, sJ, lane, g * NTW, rt, rows);
            };
            // tile 0 is handed to the consumer before its pos / vel are stored: the recurrence is the critical path
            if (u == vb) {
                if (have) produce(0, std::true_type());
                __syncthreads();                                            // the table copy has landed (all five waves;
            } else if (have) {                                              // every workgroup owns at least one unit)
                produce(0, std::false_type());
            }
            __syncthreads();                                                // tile 0 is in image 0
            if (have) store_arrays(std::integral_constant<int, 3>(), 0);
            for (int rt = 0; rt < NRT; ++rt) {
                if (have && rt + 1 < NRT) {
                    produce(rt + 1, std::false_type());
                    __builtin_amdgcn_wave_barrier();
                    store_arrays(std::integral_constant<int, 3>(), rt + 1);
                }
                __syncthreads();                                            // tile rt's actions are final
                if (have) store_arrays(std::integral_constant<int, 4>(), rt);
            }
        }
    }
}

#if !defined(MPK_DEVICE_ONLY) && MPK_PART != 0
template <int MP, int CT>
static int launch_traj_t(const TrajArgs& ta, const ActArgs& aa, bool stream_mode, bool write_through, bool bulk,
                         int quad, int blocks, size_t lds, void* stream, bool split = false, bool pipe = false) {
    const dim3 g(blocks), b(256);
    if (pipe) {
        if constexpr (MP != MPK_MP_DMP && CT >= 3) {
            const dim3 b5(320);
            hipStream_t s5 = (hipStream_t)stream;
            switch (ta.c.KP / 4) {
                case 1: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 1>), g, b5, lds, s5, ta, aa); break;
                case 2: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 2>), g, b5, lds, s5, ta, aa); break;
                case 3: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 3>), g, b5, lds, s5, ta, aa); break;
                default: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 4>), g, b5, lds, s5, ta, aa); break;
            }
        }
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    }
    // tile-major / split: no dynamic LDS of their own; `lds` then is the occupancy-experiment padding ("lds_pad" option)
    const size_t pad = (!stream_mode || split) ? lds : 0;
    hipStream_t s = (hipStream_t)stream;
    const int km = ta.c.KP / 4;
    if (ta.flat_img > 0) {
        if constexpr (MP != MPK_MP_DMP && CT < 3) {
            auto go = [&](auto kern) {
                if (lds > 48 * 1024) (void)allow_full_lds(kern);
                hipLaunchKernelGGL(kern, g, b, lds, s, ta, aa);
            };
            switch (km) {
                case 1: go(k_traj_flat<MP, CT, 1>); break;
                case 2: go(k_traj_flat<MP, CT, 2>); break;
                case 3: go(k_traj_flat<MP, CT, 3>); break;
                default: go(k_traj_flat<MP, CT, 4>); break;
            }
        }
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    }
    if (split) {
        if constexpr (MP != MPK_MP_DMP && CT >= 3) {
            if (write_through) {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_split<MP, CT, 1, true>), g, b, pad, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_split<MP, CT, 2, true>), g, b, pad, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_split<MP, CT, 3, true>), g, b, pad, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_split<MP, CT, 4, true>), g, b, pad, s, ta, aa); break;
                }
            } else {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_split<MP, CT, 1, false>), g, b, pad, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_split<MP, CT, 2, false>), g, b, pad, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_split<MP, CT, 3, false>), g, b, pad, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_split<MP, CT, 4, false>), g, b, pad, s, ta, aa); break;
                }
            }
        }
    } else if (stream_mode && quad) {
        if constexpr (MP == MPK_MP_DMP || CT >= 3) {
            if (quad == 1) {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_quad<MP, CT, 1, 1>), g, b, lds, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_quad<MP, CT, 2, 1>), g, b, lds, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_quad<MP, CT, 3, 1>), g, b, lds, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_quad<MP, CT, 4, 1>), g, b, lds, s, ta, aa); break;
                }
            } else if (quad == 2) {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_quad<MP, CT, 1, 2>), g, b, lds, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_quad<MP, CT, 2, 2>), g, b, lds, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_quad<MP, CT, 3, 2>), g, b, lds, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_quad<MP, CT, 4, 2>), g, b, lds, s, ta, aa); break;
                }
            } else {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_quad<MP, CT, 1, 4>), g, b, lds, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_quad<MP, CT, 2, 4>), g, b, lds, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_quad<MP, CT, 3, 4>), g, b, lds, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_quad<MP, CT, 4, 4>), g, b, lds, s, ta, aa); break;
                }
            }
        }
    } else if (stream_mode) {
        if (bulk) {
            // more than 48 KB of dynamic LDS only happens with the "lds_pad" occupancy knob (one workgroup per CU)
            auto big = [&](auto kern) {
                if (lds > 48 * 1024) (void)allow_full_lds(kern);
            };
            switch (km) {
                case 1: big(k_traj_stream<MP, CT, 1, true>); hipLaunchKernelGGL((k_traj_stream<MP, CT, 1, true>), g, b, lds, s, ta, aa); break;
                case 2: big(k_traj_stream<MP, CT, 2, true>); hipLaunchKernelGGL((k_traj_stream<MP, CT, 2, true>), g, b, lds, s, ta, aa); break;
                case 3: big(k_traj_stream<MP, CT, 3, true>); hipLaunchKernelGGL((k_traj_stream<MP, CT, 3, true>), g, b, lds, s, ta, aa); break;
                default: big(k_traj_stream<MP, CT, 4, true>); hipLaunchKernelGGL((k_traj_stream<MP, CT, 4, true>), g, b, lds, s, ta, aa); break;
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
                    case 1: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 1, true>), g, b, pad, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 2, true>), g, b, pad, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 3, true>), g, b, pad, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 4, true>), g, b, pad, s, ta, aa); break;
                }
            } else {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 1, false>), g, b, pad, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 2, false>), g, b, pad, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 3, false>), g, b, pad, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 4, false>), g, b, pad, s, ta, aa); break;
                }
            }
        }
    }
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

#if !defined(MPK_DEVICE_ONLY) && MPK_PART != 0
template <int MP>
int launch_traj_ct(const TrajArgs& ta, const ActArgs& aa, int ct, bool stream_mode, bool write_through,
                   bool bulk, int quad, int blocks, size_t lds, void* stream, bool split, bool pipe) {
    if constexpr (MP != MPK_MP_DMP) {
        if (pipe) {
            switch (ct) {
                case 3 + MPK_CTRL_MOTOR: return launch_traj_t<MP, 3 + MPK_CTRL_MOTOR>(ta, aa, true, false, false, 0, blocks, lds, stream, false, true);
                case 3 + MPK_CTRL_VELOCITY: return launch_traj_t<MP, 3 + MPK_CTRL_VELOCITY>(ta, aa, true, false, false, 0, blocks, lds, stream, false, true);
                default: return launch_traj_t<MP, 3 + MPK_CTRL_POSITION>(ta, aa, true, false, false, 0, blocks, lds, stream, false, true);
            }
        }
        if (split) {
            switch (ct) {
                case 3 + MPK_CTRL_MOTOR: return launch_traj_t<MP, 3 + MPK_CTRL_MOTOR>(ta, aa, false, write_through, false, 0, blocks, lds, stream, true);
                case 3 + MPK_CTRL_VELOCITY: return launch_traj_t<MP, 3 + MPK_CTRL_VELOCITY>(ta, aa, false, write_through, false, 0, blocks, lds, stream, true);
                default: return launch_traj_t<MP, 3 + MPK_CTRL_POSITION>(ta, aa, false, write_through, false, 0, blocks, lds, stream, true);
            }
        }
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
#endif  // MPK_DEVICE_ONLY

#if !defined(MPK_DEVICE_ONLY)
#if MPK_PART == 0
// defined in the translation units MPK_PART 1..3
template <int MP>
int launch_traj_ct(const TrajArgs& ta, const ActArgs& aa, int ct, bool stream_mode, bool write_through, bool bulk,
                   int quad, int blocks, size_t lds, void* stream, bool split, bool pipe);
extern template int launch_traj_ct<MPK_MP_PROMP>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
extern template int launch_traj_ct<MPK_MP_DMP>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
extern template int launch_traj_ct<MPK_MP_PRODMP>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
#elif MPK_PART > 0
template int launch_traj_ct<MPK_PART - 1>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
#endif
#endif

#if MPK_MAIN
#ifndef MPK_DEVICE_ONLY
int launch_traj_shared(const DevCfg& c, const SharedTables& st, const float* params, const float* init_pos,
                       const float* init_vel, float* pos, float* vel, float* actions, const RolloutDev* rc,
                       const double* c_pos, const double* c_vel, double* q_state, double* qd_state,
                       const int32_t* n_steps, int B, int num_cu, void* stream, const char** kernel_name,
                       const Tuning& tune, const ReplanDev* rp) {
    TrajArgs ta;
    ta.nrt_magic = 0; ta.gstride = 0; ta.wt = 0; ta.flat_img = 0;
    if (rp) ta.rp = *rp;
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
    const bool pipe_fits = table_bytes + 2 * kPipeGroups * kQuadImg * sizeof(float) <= 64 * 1024;
    const long pipe_units = ((long)ta.G + kPipeGroups - 1) / kPipeGroups;
    const bool pipe = closed && c.mp_type != MPK_MP_DMP && pipe_fits && tune.split != 1 &&
                      (tune.pipe == 1 || (tune.pipe != 0 && !variant_forced && pipe_units <= 3L * num_cu));
    const bool split = !pipe && closed && c.mp_type != MPK_MP_DMP && split_shape && tune.split == 1;
    bool stream_mode = !split && (c.mp_type == MPK_MP_DMP || closed || out_bytes > 96.0 * 1024 * 1024);
    if (c.mp_type != MPK_MP_DMP && !closed && ov == 1) stream_mode = false;
    if (ov == 2 && !split) stream_mode = true;       // a forced k_traj_split stays tile-major (its tiles role needs that geometry)
    if (tune.flat == 1 && !closed && c.mp_type != MPK_MP_DMP && !split) stream_mode = true;   // forced k_traj_flat (where it applies)
    if (stream_mode && table_bytes + 4 * kStageFloats * sizeof(float) > 64 * 1024) {
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
            return table_bytes + (4 * nq * kQuadImg) * sizeof(float) <= 64 * 1024;
        };
        const bool serial_variant = stream_mode && (c.mp_type == MPK_MP_DMP || closed);
        const int quad_mode = tune.quad < 0 ? 1 : tune.quad;
        const long units4 = (ta.G + 3) / 4, units2 = (ta.G + 1) / 2;
        // automatic (A/B-measured, profiles/r01_replan_end_to_end.md):
        //   four per wave  while that gives two waves per SIMD but not yet more units than resident waves
        //                  (cfg3 DMP at B = 16384: 35 us vs 44 with two);
        //   two per wave   below that (one wave per SIMD exposes every LDS / MFMA latency: closed loop at B = 8192
        //                  22 -> 17 us) AND above it: at HBM-streaming sizes a four-group wave keeps 16 output streams
        //                  open, two groups write like the episode-major kernel (DMP at B = 262144 792 -> 590 us,
        //                  closed loop at B = 65536 189 -> 166 us);
        //   one per wave   for the closed loop at a few thousand episodes (cfg4 episodes at B = 2048: 0.061 -> 0.052 ms)
        if (serial_variant && quad_mode != 0) {
            if (quad_mode == 2) quad = fits(4) ? 4 : 0;
            else if (quad_mode == 3) quad = fits(2) ? 2 : 0;
            else if (quad_mode == 4) quad = fits(1) ? 1 : 0;
            else if (!closed && fits(4) && units4 >= (long)num_cu * 8 && units4 < max_waves) quad = 4;   // DMP only:
            // the closed loop measured equal or better with two groups at every size (B = 16384: 0.19 vs 0.21 ms / episode)
            else if (fits(2) && units2 >= (long)num_cu * 4) quad = 2;
            else if (closed && fits(1)) quad = 1;
        }
    }
    if (pipe) {
        quad = 0; bulk = false;
        lds = table_bytes;
        const long units = (ta.G + kPipeGroups - 1) / kPipeGroups;
        const long cap = (long)num_cu * 6;                                // 5-wave workgroups: one resident round
        blocks = (int)(units < cap ? units : cap);
        if (blocks >= 8) blocks = blocks / 8 * 8;                         // XCD-contiguous remap needs a multiple of 8
        if (blocks < 1) blocks = 1;
    } else if (quad) {
        lds = table_bytes;
        const long units = (ta.G + quad - 1) / quad;
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
        // mpk_set_option "bulk": 0 disables, 2 forces it below the size threshold too (tests); default: HBM-streaming sizes only
        const int bulk_mode = tune.bulk < 0 ? 1 : tune.bulk;
        // automatic: only when the outputs stream to HBM AND the 4x coarser work units still fill the chip; the
        // latency-bound DMP recurrence prefers occupancy over input staging
        const long chunks = (ta.G + kChunkGroups - 1) / kChunkGroups;
        const bool auto_ok = out_bytes > 96.0 * 1024 * 1024 && chunks >= max_waves / 2 && c.mp_type != MPK_MP_DMP;
        bulk = bulk && bulk_mode != 0 && (bulk_mode == 2 || auto_ok);
        long units = ta.G;
        if (bulk) { lds = lds_bulk; units = (ta.G + kChunkGroups - 1) / kChunkGroups; }
        long waves = units < max_waves ? units : max_waves;
        // whole-trajectory images (k_traj_flat): open loop, promp / prodmp, aligned outputs, T * D a multiple of 4, and
        // two workgroups' images + tables within a CU's LDS.  Automatic once the outputs stream to HBM (A/B on the
        // streaming row, profiles/r03_streaming.md); mpk_set_option "flat": 0 off, 1 force
        const int flat_img = (NTW * TD + 3) / 4 * 4;
        const size_t lds_flat = table_bytes + (size_t)4 * nst * flat_img * sizeof(float);
        const bool flat_ok = !closed && c.mp_type != MPK_MP_DMP && ptr_ok && TD % 4 == 0 && lds_flat <= 80 * 1024;
        if (flat_ok && tune.flat != 0 && (tune.flat == 1 || (out_bytes > 96.0 * 1024 * 1024 && tune.bulk < 0))) {
            ta.flat_img = flat_img;
            bulk = false;
            // (write-through while the outputs fit the memory-side cache: kWtBytes)
            lds = lds_flat + (tune.lds_pad > 0 ? (size_t)tune.lds_pad * 1024 : 0);   // "lds_pad": occupancy experiments
            const long wg = (long)(160 * 1024 / lds) < 3 ? (long)(160 * 1024 / lds) : 3;   // workgroups a CU's LDS holds
            const long resident = (long)num_cu * (wg < 1 ? 1 : wg) * 4;   // 4-wave workgroups, persistent
            waves = ta.G < resident ? ta.G : resident;
        }
        blocks = (int)((waves + 3) / 4);
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;                  // XCD-contiguous remap needs a multiple of 8
    } else {
        const long items = (long)ta.G * NRT;
        long ipw = (items + max_waves - 1) / max_waves;                  // items per wave, balanced
        if (tune.ipw > 0) ipw = tune.ipw;                                // mpk_set_option "ipw" (A/B runs)
        // the kernel divides wave ids by NRT with a 32-bit multiply-high: exact while #waves < 2^32 / NRT
        const long wave_cap = (long)((1ull << 32) / (unsigned long long)NRT) - 8 * NRT;
        if ((items + ipw - 1) / ipw > wave_cap) ipw = (items + wave_cap - 1) / wave_cap;
        const long waves = (items + ipw - 1) / ipw;
        blocks = (int)((waves + 3) / 4);
        {   // #waves % NRT == 0, and a multiple of 8 blocks for the XCD remap once there are that many
            int g8 = 8, r = NRT;
            while (r) { const int t = g8 % r; g8 = r; r = t; }           // gcd(8, NRT)
            const int unit = blocks >= 8 ? NRT / g8 * 8 : NRT;           // lcm(8, NRT) or NRT
            blocks = (blocks + unit - 1) / unit * unit;
        }
        ta.gstride = blocks * 4 / NRT;
        ta.nrt_magic = NRT > 1 ? (unsigned)((1ull << 32) / (unsigned long long)NRT) + 1u : 0u;
    }
    if (blocks < 1) blocks = 1;
    if (!stream_mode && !pipe && ta.gstride <= 0) { set_error("internal: tile-major launch without its group stride"); return MPK_EINVAL; }
    if (!stream_mode && tune.lds_pad > 0) lds = (size_t)tune.lds_pad * 1024;     // A/B runs: caps the workgroups per CU
    if (stream_mode && !pipe && !ta.flat_img && tune.lds_pad > 0) lds += (size_t)tune.lds_pad * 1024;   // episode-major: EXTRA dynamic LDS (occupancy experiments)
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
    switch (c.mp_type) {
        case MPK_MP_PRODMP:
            *kernel_name = pipe ? "k_traj_pipe<prodmp,closed>" : split ? "k_traj_split<prodmp,closed>" : closed ? (quad == 4 ? "k_traj_quad<prodmp,closed>" : quad == 2 ? "k_traj_duo<prodmp,closed>" : quad == 1 ? "k_traj_mono<prodmp,closed>" : "k_traj_stream<prodmp,closed>") : ta.flat_img ? (act ? "k_traj_flat<prodmp,act>" : "k_traj_flat<prodmp>") : stream_mode ? (act ? "k_traj_stream<prodmp,act>" : "k_traj_stream<prodmp>")
                                       : (act ? "k_traj_tiles<prodmp,act>" : "k_traj_tiles<prodmp>");
            return launch_traj_ct<MPK_MP_PRODMP>(ta, aa, ct, stream_mode, write_through, bulk, quad, blocks, lds, stream, split, pipe);
        case MPK_MP_PROMP:
            *kernel_name = pipe ? "k_traj_pipe<promp,closed>" : split ? "k_traj_split<promp,closed>" : closed ? (quad == 4 ? "k_traj_quad<promp,closed>" : quad == 2 ? "k_traj_duo<promp,closed>" : quad == 1 ? "k_traj_mono<promp,closed>" : "k_traj_stream<promp,closed>") : ta.flat_img ? (act ? "k_traj_flat<promp,act>" : "k_traj_flat<promp>") : stream_mode ? (act ? "k_traj_stream<promp,act>" : "k_traj_stream<promp>")
                                       : (act ? "k_traj_tiles<promp,act>" : "k_traj_tiles<promp>");
            return launch_traj_ct<MPK_MP_PROMP>(ta, aa, ct, stream_mode, write_through, bulk, quad, blocks, lds, stream, split, pipe);
        default:
            *kernel_name = quad == 4 ? "k_traj_quad<dmp>" : quad == 2 ? "k_traj_duo<dmp>" : quad == 1 ? "k_traj_mono<dmp>" : "k_traj_stream<dmp>";
            return launch_traj_ct<MPK_MP_DMP>(ta, aa, -1, true, false, bulk, quad, blocks, lds, stream, false, false);
    }
}
#endif  // MPK_DEVICE_ONLY

// ------------------------------------------------------------------------------------------------------------
// k_traj_wide<MP, MT>: shared phase with MORE than 16 contraction columns -- the reference's own "too many basis
// functions" example (examples/examples_movement_primitives.py:67: num_basis = 1000 on a 5-DoF, 200-step ProMP).
// There the path is a real GEMM  C[T x (B D)] = A[T x K] . X[K x (B D)]  (2 MFLOP against 28 KB per episode at K = 1000:
// matrix-core bound, not HBM bound), so it is tiled like one:
//   * a workgroup of 4 waves takes 4 consecutive episode groups (16 (episode, DoF) columns each), one per wave;
//   * the k loop runs in chunks of KC columns: the chunk of the k-major basis table (rows of MT row tiles, all outputs)
//     is copied ONCE per workgroup into LDS with float4 loads (stride == 16 mod 32: the two k rows of a 32-lane
//     fragment read fall on disjoint banks), each wave stages ITS 16 parameter columns beside it -- lane <-> k, i.e. 256
//     contiguous bytes per column and load instruction (raw parameters / boundary conditions, as everywhere: all scales
//     live in the basis rows) -- in a [16][KC + 2] image (conflict-free fragment reads);
//   * per 4 columns of k: ONE B fragment and MT A fragments from LDS feed MT x NOUT v_mfma_f32_16x16x4_f32 on MT x NOUT
//     independent accumulators (the whole horizon of the group stays in registers: nothing is re-read);
//   * epilogue per wave through an LDS image [T][17]: promp's forward difference of the fp32 positions (x aux, the
//     reciprocal fp32 time step, as in the tile kernels), dmp's explicit Euler recurrence on the group's D x epg lanes,
//     coalesced copy-out of each episode's contiguous [T][D] block.
// Accumulation order = ascending k, the order of every other kernel of this file (an MFMA is a k-ordered fmaf chain).
// Two workgroups fit a CU (LDS), so one stages while the other contracts.  Horizons beyond MT row tiles: prodmp walks
// row-tile blocks (its rows are independent); promp / dmp need the whole horizon in one block (T <= 512).
// ------------------------------------------------------------------------------------------------------------
struct WideArgs {
    DevCfg c;
    const float* A;      // [n_out][KP][TS] (k_build_shared)
    const float* aux;    // [TS]
    int TS;
    const float* params;
    const float* init_pos;
    const float* init_vel;
    float* pos;
    float* vel;
    int B, epg, n_units, KC, n_rt, SA;   // episodes per column group, 4-group units, k chunk, row tiles, LDS row stride
    int cgpe;                            // column groups per episode: 1 (D <= 16), else ceil(D / 16) with epg == 1
    int aux_ofs;                         // floats: LDS copy of aux[TS] behind the staging area / epilogue images
};

// raw operand of the contraction for column (episode b, DoF dd), index k  (the sX fill of k_traj_rows)
template <int MP>
__device__ __forceinline__ float wide_x(const DevCfg& c, const float* __restrict__ prm, float ip, float iv, int dd, int k) {
    if (MP == MPK_MP_PRODMP) {
        const int nb = c.nb;
        if (k < nb) return c.disable_weights ? 0.0f : prm[c.off + dd * c.Kloc + k];
        if (k == nb) return c.disable_goal ? 0.0f : prm[c.off + dd * c.Kloc + (c.disable_weights ? 0 : nb)];
        if (k == nb + 1) return ip;
        if (k == nb + 2) return iv;
        return 1.0f;                                   // goal-offset column (MPK_GOAL_OFFSET_ADD)
    }
    if (MP == MPK_MP_PROMP) return k < c.nb ? prm[c.off + dd * c.Kloc + k] : ip;
    return prm[c.off + dd * c.Kloc + k];
}

// 64 accumulator registers or fewer: two workgroups per CU (one stages while the other contracts); 128: one workgroup per CU
// with the whole register file (the register prefetch of the next chunk covers the global latency either way)
template <int MP, int MT>
__global__ void __launch_bounds__(256, ((MP == MPK_MP_PRODMP ? 2 : 1) * MT <= 16 ? 2 : 1)) k_traj_wide(const WideArgs a) {
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 1;
    constexpr int CS = 17;                              // column stride of the epilogue image
    // register images of the NEXT k chunk (requested before the current chunk is contracted).
    // A: wave w takes table rows w, w + 4, ... of the chunk, lane <-> float4 of the row (one coalesced load of up to 1 KB per
    // row and 64-lane span), committed to LDS after the contraction: ANR float4 per lane.
    // X: the B fragments themselves -- lane (column n = lane & 15, k quarter lane >> 4) loads X[k0 + 4 j + (lane >> 4)][n]
    // for j < KC / 4 straight from the column's parameter row (a quad of lanes = 16 contiguous bytes; the KC floats of a
    // column are one or two cache lines that the chunk's loads share): no LDS staging, no per-step LDS read for B.
    constexpr int ANR = 8, XNR = 8;                     // KC <= 32
    // LDS layout of the A chunk: [table row = o * KC + kk][step-in-tile m][row tile r], MTP floats per m.  A lane's fragments
    // of one k for ALL row tiles are contiguous (ds_read_b128: four tiles per read instead of one ds_read_b32 per MFMA);
    // MTP / 4 odd and 16 * MTP a multiple of 64 floats: the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte
    // bank groups (a 16-lane group mixes two k rows: rows are 64-float multiples apart, m * MTP / 4 is a bijection mod 16)
    constexpr int MTQ = (MT + 3) / 4 + (((MT + 3) / 4) % 2 == 0 ? 1 : 0), MTP = 4 * MTQ, SA = 16 * MTP, NR4 = (MT + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevCfg& c = a.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g4 = lane >> 4;
    const int D = c.D, T = c.T, KP = c.KP, KT = c.KT, TS = a.TS, KC = a.KC;
    float* sA = smem;                                   // [NOUT][KC][16][MTP]
    // epilogue images alias the staging area (used after the k loop, behind a barrier): per wave [NIMG][MT*16][CS]
    constexpr int NIMG = MP == MPK_MP_PROMP ? 1 : 2;
    float* sC = smem + (size_t)wave * NIMG * MT * 16 * CS;
    // aux (promp: reciprocal time steps, dmp: scaled-time increments) behind both: read per step in the epilogue, and a
    // global load there waits for every store before it (one counter for loads and stores) -- from LDS it does not
    float* sAux = smem + a.aux_ofs;
    if (MP != MPK_MP_PRODMP) {
        for (int t = tid; t < TS; t += 256) sAux[t] = a.aux[t];
        __syncthreads();
    }
    const int kshift = 31 - __builtin_clz(KC);          // KC is a power of two
    const int nj = KC >> 2;                             // MFMA steps per chunk

#ifdef WIDE_TIME
    unsigned long long tw_mfma = 0, tw_sync = 0, tw_fetch = 0, tw_epi = 0, tw_t0 = __builtin_readcyclecounter(), tw_a, tw_b;
#define TW_A() tw_a = __builtin_readcyclecounter()
#define TW_B(acc) do { tw_b = __builtin_readcyclecounter(); acc += tw_b - tw_a; tw_a = tw_b; } while (0)
#else
#define TW_A()
#define TW_B(acc)
#endif
    for (int unit = blockIdx.x; unit < a.n_units; unit += gridDim.x) {
        const int grp = unit * 4 + wave;
        // D <= 16: the group holds epg whole episodes; D > 16: 16 consecutive DoF (from d0) of ONE episode
        const int b0 = a.cgpe == 1 ? grp * a.epg : grp / a.cgpe;
        const int d0 = a.cgpe == 1 ? 0 : (grp - b0 * a.cgpe) * 16;
        const int ncol = a.cgpe == 1 ? a.epg * D : min(16, D - d0);     // used columns of the group
        // this lane's column: episode, DoF, the sources of its operand column
        const int ce = (int)(((unsigned)m * (65536u / (unsigned)D + 1u)) >> 16), cd = d0 + m - ce * D;
        const int cb = b0 + ce;
        const bool cvalid = m < ncol && cb < a.B;
        const float* cprm = a.params + (size_t)(cvalid ? cb : 0) * c.P;
        const float* cw = cprm + c.off + cd * c.Kloc;                           // the DoF's local block
        const float cip = cvalid && MP != MPK_MP_DMP ? a.init_pos[(size_t)cb * D + cd] : 0.0f;
        const float civ = cvalid && MP == MPK_MP_PRODMP ? a.init_vel[(size_t)cb * D + cd] : 0.0f;
        // columns [0, kplain) are plain parameters at cw[k] (the weights; promp / dmp: all learnable columns)
        const int kplain = (MP == MPK_MP_PRODMP && c.disable_weights) ? 0 : c.nb;
        for (int rt0 = 0; rt0 < a.n_rt; rt0 += MT) {
            const int nrt = min(MT, a.n_rt - rt0);      // row tiles of this block
            const int rowsA = nrt * 16;
            const int r4 = rowsA >> 2;                  // float4 per table row
            const int spans = (r4 + 63) >> 6;           // 64-lane spans per row (1, or 2 for more than 16 row tiles)
            const int nrows = NOUT * KC;                // table rows per chunk; rows per wave x spans <= ANR (launcher)
            f32x4 acc[NOUT][MT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int r = 0; r < MT; ++r) acc[o][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            float4 ra[ANR];
            float xn[XNR], xc[XNR];
            // ---- requests of one k chunk: every load is issued before any of them is used ----
            // Kept lean (the whole non-MFMA part of a chunk is time the wave's SIMD partner -- the other workgroup's wave --
            // must cover with its own MFMAs): table rows through a wave-UNIFORM pointer (scalar address arithmetic) + one
            // lane offset per span, lanes past the row clamped onto its last float4 instead of masked (commit skips them),
            // invalid columns pointed at the start of `params` instead of masked (their products land in columns nobody
            // stores).  Per-wave cycle budget of the num_basis = 1000 launch (build with -DWIDE_TIME, tools/dev/wide_time.py;
            // profiles/r03_wide.md): contraction 33 %, the wait for this lambda's loads one chunk later 36 %, the two
            // barriers + commit 16 %, epilogue 8 % -- two waves per SIMD, so the matrix pipes idle whenever both wait.
            const int q0 = min(lane, r4 - 1), q1 = min(lane + 64, r4 - 1);        // clamped float4 index per span
            const float* const cwl = cvalid ? cw + g4 : a.params;                  // invalid column: any readable floats
            const int xmax = cvalid ? 0x7fffffff : 0;                             // ... at offset 0
            auto fetch = [&](int k0) {
#ifdef WIDE_NO_LOADS
                if (k0 > 0) return;
#endif
#pragma unroll
                for (int p = 0; p < ANR; ++p) {
                    const int row = wave + 4 * (spans == 1 ? p : (p >> 1));       // wave-uniform: o * KC + kk
                    const int kk = row & (KC - 1), o = row >> kshift;
                    const int k = k0 + kk;
                    ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < nrows && k < KP) {                                  // (uniform)
                        const float4* rowp = reinterpret_cast<const float4*>(a.A + ((size_t)o * KP + k) * TS + rt0 * 16);
                        ra[p] = rowp[spans == 1 || !(p & 1) ? q0 : q1];
                    }
                }
                if (k0 + KC <= kplain) {                // the common chunk: nothing but parameters (wave-uniform test)
                    const float* cwk = cwl + min(k0, xmax);
#pragma unroll
                    for (int p = 0; p < XNR; ++p) {
                        xn[p] = 0.0f;
                        if (p < nj) xn[p] = cwk[4 * p];
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < XNR; ++p) {
                        const int k = k0 + 4 * p + g4;
                        xn[p] = 0.0f;
                        if (p < nj && cvalid && k < KT) xn[p] = wide_x<MP>(c, cprm, cip, civ, cd, k);
                    }
                }
            };
            // LDS image of the chunk: float4 = steps 4 q .. 4 q + 3 of the block -> row tile q / 4, steps-in-tile 4 (q % 4) + e;
            // lane + 64 of the second span has the same (q & 3) and (q >> 2) + 16: ONE lane-dependent address, the rest of
            // every address is wave-uniform
            float* const w0 = sA + (size_t)wave * SA + (4 * (lane & 3)) * MTP + (lane >> 2);
            auto commit = [&]() {
                if (lane < r4) {
#pragma unroll
                    for (int p = 0; p < ANR; ++p) {
                        if (spans == 1 || !(p & 1)) {
                            const int row = wave + 4 * (spans == 1 ? p : (p >> 1));
                            if (row < nrows) {
                                float* w = w0 + (size_t)(4 * (spans == 1 ? p : (p >> 1))) * SA;
                                w[0] = ra[p].x; w[MTP] = ra[p].y; w[2 * MTP] = ra[p].z; w[3 * MTP] = ra[p].w;
                            }
                        }
                    }
                }
                if (spans == 2 && lane + 64 < r4) {
#pragma unroll
                    for (int p = 1; p < ANR; p += 2) {
                        const int row = wave + 4 * (p >> 1);
                        if (row < nrows) {
                            float* w = w0 + (size_t)(4 * (p >> 1)) * SA + 16;
                            w[0] = ra[p].x; w[MTP] = ra[p].y; w[2 * MTP] = ra[p].z; w[3 * MTP] = ra[p].w;
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < XNR; ++p) xc[p] = xn[p];
            };
            fetch(0);
            for (int k0 = 0; k0 < KP; k0 += KC) {
                TW_A();
                __syncthreads();                        // the previous chunk (or epilogue image) is consumed
                commit();
                __syncthreads();
                TW_B(tw_sync);
                if (k0 + KC < KP) fetch(k0 + KC);       // in flight under this chunk's contraction
                TW_B(tw_fetch);
                // ---- contraction of the chunk ----
                const float* pa = sA + (size_t)g4 * SA + m * MTP;
                // the full chunk (KC = 32: promp / dmp; KC = 16: prodmp, whose two outputs share the chunk): the A fragments of
                // step p + 1 are requested BEFORE the MFMAs of step p are issued -- left to itself the compiler reads one
                // ds_read_b128 into one register quad, waits, issues its four MFMAs, reads the next (the LDS round trip exposed
                // once per four MFMAs).  Two fragment sets where the accumulators leave room (<= 64 of them), else all reads
                // of a step ahead of its MFMAs.  A wave's contraction now runs at 93 % of the pipe's rate while it lasts; the
                // launch as a whole did not get faster by it (the waits between contractions dominate, see `fetch`).
                auto contract_full = [&](auto nj_tag) {
                    constexpr int NJ = decltype(nj_tag)::value;
                    constexpr int NB = NOUT * MT <= 16 ? 2 : 1;
                    f32x4 af[NB][NOUT][NR4];
                    auto load_step = [&](int buf, int p) {
#pragma unroll
                        for (int o = 0; o < NOUT; ++o)
#pragma unroll
                            for (int c4 = 0; c4 < NR4; ++c4)
                                af[buf][o][c4] = *reinterpret_cast<const f32x4*>(pa + ((size_t)o * KC + 4 * p) * SA + 4 * c4);
                    };
                    load_step(0, 0);
#pragma unroll
                    for (int p = 0; p < NJ; ++p) {
                        const int cur = NB == 2 ? (p & 1) : 0;
                        if (NB == 2 && p + 1 < NJ) load_step(cur ^ 1, p + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        const float bf = xc[p];
#pragma unroll
                        for (int r = 0; r < MT; ++r) {
#pragma unroll
                            for (int o = 0; o < NOUT; ++o)
                                acc[o][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][o][r >> 2][r & 3], bf, acc[o][r], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (NB == 1 && p + 1 < NJ) load_step(0, p + 1);
                    }
                };
                constexpr int NJ_FULL = NOUT == 1 ? XNR : XNR / 2;      // what the launcher's chunk gives (ANR rows per wave)
                // (prodmp with 7 / 8 row tiles sits at the 256-register cap of two workgroups per CU already: the read-ahead
                // spills there, so those two variants keep the plain loop)
                constexpr bool kReadAhead = !(NOUT == 2 && MT > 4 && MT <= 8);
                if (kReadAhead && nj == NJ_FULL) {
                    if constexpr (kReadAhead) contract_full(std::integral_constant<int, NJ_FULL>());
                } else
#pragma unroll
                for (int p = 0; p < XNR; ++p) {
                    if (p < nj) {
                        const float bf = xc[p];
                        // all MT row tiles, unconditionally: MT is the launch's exact row-tile count (or, in the last block
                        // of a long prodmp horizon, more -- those tiles contract stale LDS into accumulators nobody stores);
                        // a guard per tile is a branch per MFMA, and a branch between an LDS read and its MFMA keeps the
                        // compiler from issuing the reads ahead (measured: 17.8 -> 55 TF once the loads pipelined)
                        f32x4 af[NOUT][NR4];
#pragma unroll
                        for (int o = 0; o < NOUT; ++o)
#pragma unroll
                            for (int c4 = 0; c4 < NR4; ++c4)
                                af[o][c4] = *reinterpret_cast<const f32x4*>(pa + ((size_t)o * KC + 4 * p) * SA + 4 * c4);
#pragma unroll
                        for (int r = 0; r < MT; ++r) {
#pragma unroll
                            for (int o = 0; o < NOUT; ++o)
                                acc[o][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[o][r >> 2][r & 3], bf, acc[o][r], 0, 0, 0);
                        }
                    }
                }
                TW_B(tw_mfma);
            }
            __syncthreads();                            // every wave is done with the staging area
            // ---- epilogue: C tiles -> image [t][col] (row = 4 * (lane >> 4) + i of tile r, column = lane & 15) ----
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int r = 0; r < MT; ++r)
                    if (r < nrt) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) sC[((size_t)o * MT * 16 + r * 16 + 4 * g4 + i) * CS + m] = acc[o][r][i];
                    }
            __builtin_amdgcn_wave_barrier();
            const int t_lo = rt0 * 16;
            const int t_n = min(T - t_lo, rowsA);       // valid steps of this block
            float* img0 = sC;
            float* img1 = sC + (size_t)MT * 16 * CS;
            if (MP == MPK_MP_DMP) {
                // explicit Euler, one lane per used column (the operation order of every dmp kernel in this file)
                if (lane < ncol && b0 + lane / D < a.B) {
                    const int e = lane / D, dd = d0 + lane - e * D;
                    const int b = b0 + e;
                    const float* prm = a.params + (size_t)b * c.P;
                    float y = a.init_pos[(size_t)b * D + dd];
                    float z = a.init_vel[(size_t)b * D + dd] * c.tau;
                    const float gl = prm[c.off + dd * c.Kloc + c.nb] * c.gs;
                    const TauDiv td = make_tau_div(c.tau);
                    for (int t = 0; t < T; ++t) {
                        const float f = img0[(size_t)t * CS + lane];
                        img0[(size_t)t * CS + lane] = y;
                        img1[(size_t)t * CS + lane] = div_tau(z, td);
                        if (t < T - 1) {
                            const float ds = sAux[t];
                            const float t1 = gl - y;
                            const float t2 = c.dmp_beta * t1;
                            const float t3 = t2 - z;
                            const float t4 = c.dmp_alpha * t3;
                            const float ac = t4 + f;
                            z = z + ds * ac;
                            y = y + ds * z;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            // copy-out: lane <-> (row of the round, used column of the group); a round covers 64 / ncol rows of every episode
            // of the group, each episode's share one contiguous run of HBM.  Everything lane-dependent is computed once per
            // unit, the loop adds wave-uniform strides, and nothing in it loads from global memory (aux comes from LDS: a
            // global load waits for every store issued before it -- that wait was half of this loop's time)
            {
                const unsigned rn = 65536u / (unsigned)ncol + 1u;
                const int rl = (int)(((unsigned)lane * rn) >> 16), col = lane - rl * ncol;       // lane / ncol, lane % ncol
                const int R = (int)((64u * rn) >> 16);                                            // rows per round
                const int e = a.cgpe == 1 ? (int)(((unsigned)col * (65536u / (unsigned)D + 1u)) >> 16) : 0;
                const int dd = a.cgpe == 1 ? col - e * D : col;
                const bool on = rl < R && b0 + e < a.B;
                // wave-uniform bases (scalar registers) + one 32-bit lane offset shared by both arrays
                float* const pw = a.pos + ((size_t)b0 * T + t_lo) * D + d0;
                float* const vw = a.vel + ((size_t)b0 * T + t_lo) * D + d0;
                int off = (e * T + rl) * D + dd;
                int li = rl * CS + col;
                if (on) {
                    for (int t = rl; t < t_n; t += R, off += R * D, li += R * CS) {
                        float p, v;
                        if (MP == MPK_MP_PROMP) {
                            // vel = forward difference of the fp32 positions, last row repeats; t_lo == 0 here
                            const bool last = t == T - 1;
                            const int la = last ? li - CS : li;
                            const float lo = img0[la], hi = img0[la + CS];
                            p = last ? hi : lo;
                            v = (hi - lo) * sAux[t];
                        } else {
                            p = img0[li];
                            v = img1[li];
                        }
                        pw[off] = p;
                        vw[off] = v;
                    }
                }
            }
            TW_B(tw_epi);
        }
    }
#ifdef WIDE_TIME
    __syncthreads();
    if (lane == 0) {       // debug build: the wave's cycle budget instead of results, in the first floats of `vel`
        float* o = a.vel + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = (float)(__builtin_readcyclecounter() - tw_t0); o[1] = (float)tw_mfma; o[2] = (float)tw_sync;
        o[3] = (float)tw_fetch; o[4] = (float)tw_epi; o[5] = (float)blockIdx.x; o[6] = (float)wave; o[7] = (float)gridDim.x;
    }
#endif
}

#ifndef MPK_DEVICE_ONLY
int launch_traj_wide(const DevCfg& c, const SharedTables& st, const float* params, const float* init_pos,
                     const float* init_vel, float* pos, float* vel, int B, int num_cu, void* stream,
                     const char** kernel_name) {
    if (c.mp_type == MPK_MP_PROMP && c.T < 2) {
        set_error("promp needs at least two time steps for the finite-difference velocity");
        return MPK_EINVAL;
    }
    const int n_rt = (c.T + 15) / 16;
    const int nout = c.mp_type == MPK_MP_PRODMP ? 2 : 1;
    const int nimg = c.mp_type == MPK_MP_PROMP ? 1 : 2;
    const int mt_max = c.mp_type == MPK_MP_PRODMP ? 16 : 32;
    if (c.mp_type != MPK_MP_PRODMP && n_rt > mt_max) return MPK_ENOTIMPL;   // whole horizon in one row-tile block
    // row tiles per block: the smallest instantiated count >= n_rt (the contraction loop runs all MT tiles unconditionally)
    static const int kMT[] = {4, 7, 8, 12, 13, 16, 24, 32};
    int MT = mt_max;
    for (int v : kMT) if (v >= n_rt && v <= mt_max) { MT = v; break; }
    const int rows_max = (n_rt < MT ? n_rt : MT) * 16;                    // rows of a row-tile block that exist
    const int mtq = (MT + 3) / 4 + (((MT + 3) / 4) % 2 == 0 ? 1 : 0);     // the kernel's MTQ / MTP / SA
    const int SA = 16 * 4 * mtq;
    const int spans = (rows_max / 4 + 63) / 64;                            // 64-lane float4 spans per table row
    int KC = 32;                                                           // <= 8 B fragments per lane and chunk
    auto stage_bytes = [&](int kc) { return ((size_t)nout * kc * SA) * sizeof(float); };
    while (KC > 8 && (stage_bytes(KC) > 80 * 1024 || nout * KC / 4 * spans > 8)) KC >>= 1;
    while (KC > 4 && KC / 2 >= c.KP) KC >>= 1;                             // few columns (the D > 16 route): one short chunk
    const size_t epi_bytes = (size_t)4 * nimg * MT * 16 * 17 * sizeof(float);
    const size_t lds_main = stage_bytes(KC) > epi_bytes ? stage_bytes(KC) : epi_bytes;
    const size_t lds = lds_main + (c.mp_type == MPK_MP_PRODMP ? 0 : (size_t)st.TS * sizeof(float));
    const int cgpe = c.D <= 16 ? 1 : (c.D + 15) / 16;
    WideArgs wa{c, st.A, st.aux, st.TS, params, init_pos, init_vel, pos, vel, B, c.D <= 16 ? 16 / c.D : 1, 0, KC, n_rt, SA, cgpe,
                (int)(lds_main / sizeof(float))};
    if (cgpe > 1 && (long long)B * cgpe > 0x7fffffffLL - 8) return MPK_ENOTIMPL;
    const int n_groups = cgpe == 1 ? (B + wa.epg - 1) / wa.epg : B * cgpe;
    wa.n_units = (n_groups + 3) / 4;
    const int per_cu = nout * MT <= 16 ? 2 : 1;                           // see the kernel's launch bounds
    const int blocks = wa.n_units < num_cu * per_cu ? wa.n_units : num_cu * per_cu;
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = allow_full_lds(kern);
            if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, (hipStream_t)stream, wa);
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    };
    auto pick = [&](auto mp_tag) -> int {
        constexpr int MPV = decltype(mp_tag)::value;
        switch (MT) {
            case 4: return go(k_traj_wide<MPV, 4>);
            case 7: return go(k_traj_wide<MPV, 7>);
            case 8: return go(k_traj_wide<MPV, 8>);
            case 12: return go(k_traj_wide<MPV, 12>);
            case 13: return go(k_traj_wide<MPV, 13>);
            case 16: return go(k_traj_wide<MPV, 16>);
            default: break;
        }
        if constexpr (MPV != MPK_MP_PRODMP) {
            if (MT == 24) return go(k_traj_wide<MPV, 24>);
            return go(k_traj_wide<MPV, 32>);
        }
        return go(k_traj_wide<MPV, 16>);
    };
    switch (c.mp_type) {
        case MPK_MP_PRODMP: *kernel_name = "k_traj_wide<prodmp>"; return pick(std::integral_constant<int, MPK_MP_PRODMP>());
        case MPK_MP_PROMP: *kernel_name = "k_traj_wide<promp>"; return pick(std::integral_constant<int, MPK_MP_PROMP>());
        default: *kernel_name = "k_traj_wide<dmp>"; return pick(std::integral_constant<int, MPK_MP_DMP>());
    }
}
#endif  // MPK_DEVICE_ONLY

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
};

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

// the value of the neighbouring lane (lane + 1 / lane - 1 of the 64) as ONE vector instruction (DPP wave shift) instead
// of an LDS round trip (ds_bpermute behind __shfl_*): the ProMP velocity takes two of them per (step, DoF).  The lane
// without a neighbour reads 0; nothing uses it.
__device__ __forceinline__ float lane_above(float x) {      // x of lane + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_below(float x) {      // x of lane - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
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
template <int MP, int KQ, bool TL, bool FL = false>
__global__ void __launch_bounds__(TL ? 1024 : 256) k_traj_phase(const PhaseArgs a) {
    static_assert(MP != MPK_MP_DMP, "dmp has its own kernel");
    static_assert(!TL || MP == MPK_MP_PRODMP, "only prodmp has a row table");
    static_assert(!FL || MP == MPK_MP_PRODMP, "flat rounds: prodmp (promp's difference crosses lanes)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevCfg& c = a.c;
    constexpr int KS = KQ * 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int D = c.D, T = c.T, KT = c.KT;
    double* sCen = reinterpret_cast<double*>(smem);     // [c_pad / 2] RBF centres | bandwidths, shared by the workgroup
    float* sBT = smem + a.c_pad;                        // [t_pad] base times, shared by the workgroup
    float* sTab = sBT + a.t_pad;                        // TL: [n_pc][2*KS + 4] row table, shared by the workgroup
    float* sImg = sTab + a.tab_pad + (size_t)wave * a.wave_floats;  // [2][img_pad] inputs of this / the next chunk
    float* sO0 = sImg + 2 * a.img_pad;                  // [o_pad] pos staging: [sh + lane * D + d]
    float* sO1 = sO0 + a.o_pad;                         // [o_pad] vel staging
    float* sXf = sO1 + a.o_pad;                         // promp: [x_pad] this episode's columns (prodmp: in the input image)
    float* sWgs = smem;                                 // prodmp: weights_goal_scale[nb + 1] (in place of sCen)
    for (int t = threadIdx.x; t < T; t += blockDim.x) sBT[t] = c.base_times[t];
    if (MP != MPK_MP_PRODMP) {
        for (int k = threadIdx.x; k < 2 * c.n_total + 3; k += blockDim.x) sCen[k] = c.tab[k];
    } else {
        const double* S = c.tab + 4 * (size_t)c.n_pc + 2 * (size_t)c.n_pc * (c.nb + 1);
        for (int k = threadIdx.x; k <= c.nb; k += blockDim.x) {
            const bool off = k < c.nb ? c.disable_weights != 0 : c.disable_goal != 0;
            sWgs[k] = off ? 0.0f : (float)S[k];
            if (k == c.nb) sWgs[c.nb + 1] = (float)S[k];     // the goal scale itself, also when the goal is disabled
        }
        if (TL) {
            const float4* src = reinterpret_cast<const float4*>(c.rows32);
            for (int i = threadIdx.x; i < c.n_pc * (2 * KQ + 1); i += blockDim.x)
                reinterpret_cast<float4*>(sTab)[i] = src[i];
        }
    }
    __syncthreads();
    const float* const rows = TL ? sTab : c.rows32;
    constexpr int kRow = 2 * KS + 4;    // [pos half .. y1 (f64) | vel half .. y2 (f64) | dy1 dy2 (f64)]
    const ExactDiv dsdt = make_exact_div(c.scaled_dt);

    // A wave owns chunks of E consecutive episodes.  A chunk's inputs -- E parameter rows, E boundary positions /
    // velocities, E init_times: each one contiguous run -- are fetched with coalesced loads one chunk ahead and
    // collected once per chunk (into the other half of the wave's input image), right after the rows of the chunk's
    // last episode are built: the memory queue is in order, so collecting a load also waits for every store issued
    // before it, and that wait is paid per chunk instead of per episode.
    const int E = a.chunk, P = c.P;
    const int img_floats = a.img_pad;
    constexpr int NLP = 5;                              // E * P <= 320 parameter values per chunk
    const int nchunks = (a.B + E - 1) / E;
    const int cstride = (int)gridDim.x * wpb;
    int ch = (int)blockIdx.x * wpb + wave;
    float lp[NLP], lip = 0.0f, liv = 0.0f, lit = 0.0f;
    auto issue_chunk = [&](int cc) {
        const int b0 = cc * E, ne = min(E, a.B - b0);
        const float* prm = a.params + (size_t)b0 * P;
#pragma unroll
        for (int r = 0; r < NLP; ++r) lp[r] = lane + 64 * r < ne * P ? prm[lane + 64 * r] : 0.0f;
        lip = lane < ne * D ? a.init_pos[(size_t)b0 * D + lane] : 0.0f;
        liv = lane < ne * D ? a.init_vel[(size_t)b0 * D + lane] : 0.0f;
        lit = a.init_time && lane < ne ? a.init_time[b0 + lane] : a.init_time_shared;
    };
    auto park_chunk = [&](float* img) {
#pragma unroll
        for (int r = 0; r < NLP; ++r)
            if (lane + 64 * r < E * P) img[lane + 64 * r] = lp[r];
        if (lane < E * D) { img[E * P + lane] = lip; img[E * P + E * D + lane] = liv; }
        if (lane < E) img[E * P + 2 * E * D + lane] = lit;
    };
    if (ch < nchunks) {
        issue_chunk(ch);
        park_chunk(sImg);
    }
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
            const int K = c.nb + 1;
            float raw[KS], taul = c.tau, delayl = c.delay, itl = 0.0f, yb = 0.0f, ydb = 0.0f;
#pragma unroll
            for (int k = 0; k < KS; ++k) raw[k] = 0.0f;
            if (on) {
                const float* prl = img + le * P;
                if (c.learn_tau) taul = fminf(fmaxf(prl[0], c.tau_lo), c.tau_hi);
                if (c.learn_delay) delayl = fminf(fmaxf(prl[c.learn_tau ? 1 : 0], c.delay_lo), c.delay_hi);
                itl = img[E * P + 2 * E * D + le];
                yb = img[E * P + lane]; ydb = img[E * P + E * D + lane];
                const float* loc = prl + c.off + ld * c.Kloc;
#pragma unroll
                for (int k = 0; k < KS; ++k)
                    if (k < K) {
                        // a disabled block has no parameters (the goal then sits at local index 0)
                        const bool have = k < c.nb ? !c.disable_weights : !c.disable_goal;
                        const int li = k < c.nb ? k : (c.disable_weights ? 0 : c.nb);
                        raw[k] = have ? loc[li] : 0.0f;
                    }
            }
            __builtin_amdgcn_wave_barrier();                 // every read of the image is issued before its first write
            if (on) {
                const float sbl = fmaxf(div_exact(itl - delayl, make_exact_div(taul)), 0.0f);
                const float* rb = rows + (size_t)min((int)rintf(div_exact(sbl, dsdt)), c.n_pc - 1) * kRow;
                double pb = 0.0, vb = 0.0;
                float* xf = imw + le * a.x_pad + ld * KS;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    float wg = 0.0f;
                    if (k < K) {
                        const bool have = k < c.nb ? !c.disable_weights : !c.disable_goal;
                        wg = have ? raw[k] * sWgs[k] : 0.0f;          // (raw is 0 where there is no parameter)
                        if (k == c.nb) {
                            // relative goal: init_pos joins the scaled goal, or (MPK_RELGOAL_BEFORE_SCALE) the raw
                            // parameter -- zero when the goal is disabled -- before the scale
                            if (c.relative_goal) wg = c.relgoal_before_scale ? (raw[k] + yb) * sWgs[c.nb + 1] : wg + yb;
                            if (c.goal_off_on) wg = wg + c.goal_offset;
                        }
                        pb += (double)rb[2 * k] * (double)wg;
                        vb += (double)rb[2 * k + 1] * (double)wg;
                    }
                    xf[k] = wg;
                }
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
                const int idx = min((int)rintf(div_exact(s, dsdt)), c.n_pc - 1);
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
                constexpr int ND = KQ <= 2 ? 2 : 1;
                int d = 0;
                for (; d + ND <= D; d += ND) {
#pragma unroll
                    for (int q = 0; q < ND; ++q) dof(d + q);
                }
                for (; d < D; ++d) dof(d);
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
                const int idxb = min((int)rintf(div_exact(sb, dsdt)), c.n_pc - 1);
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
                    const int idx = min((int)rintf(div_exact(s, dsdt)), c.n_pc - 1);
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
                    const double x = phase_f64(c, time, taud, delay, ec);
#pragma unroll
                    for (int k = 0; k < KS; ++k) h[k] = 0.0f;
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
                constexpr int ND = KQ <= 2 ? 2 : 1;
                int d = 0;
                for (; d + ND <= D; d += ND) {
#pragma unroll
                    for (int i = 0; i < ND; ++i) dof(d + i);
                }
                for (; d < D; ++d) dof(d);
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

template <int KQ>
__global__ void __launch_bounds__(256) k_traj_phase_dmp(const PhaseArgs a) {
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
    const int D = c.D, T = c.T, E = a.chunk, P = c.P;
    const int seg = TT * D;                             // floats of one episode's tile
    double* sCen = reinterpret_cast<double*>(smem);     // [c_pad / 2] RBF centres | bandwidths (| recurrence constants)
    float* sBT = smem + a.c_pad;                        // [t_pad] base times, shared by the workgroup
    float* sX = sBT + a.t_pad + (size_t)wave * a.wave_floats;   // [E][D][KS] columns: weights .., goal, y0, ydot0
    float* sPh = sX + E * a.x_pad;                      // [E][8] tau, delay, init_time (clipped), -, 1 / tau refined (float64), -
    float* sDs = sPh + 8 * E;                           // [E][TT] ds of the tile's steps
    float* sH = sDs + E * TT;                           // [64][KS] the round's RBF rows
    float* sP = sH + 64 * KS;                           // [E][TT * D] forcing -> pos
    float* sV = sP + a.o_pad;                           // [E][TT * D] vel
    for (int t = threadIdx.x; t < T; t += blockDim.x) sBT[t] = c.base_times[t];
    for (int k = threadIdx.x; k < 2 * c.n_total + 3; k += blockDim.x) sCen[k] = c.tab[k];
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
            // ---- A: rows and forcing of the tile
            for (int i0 = 0; i0 < ne * TT; i0 += 64) {
                const int idx = i0 + lane, e = idx >> 4, tl = idx & (TT - 1), t = t0 + tl;
                const bool live = idx < ne * TT && t < T;
                float* row = sH + lane * KS;
                if (live) {
                    const float taue = sPh[8 * e], delaye = sPh[8 * e + 1], ite = sPh[8 * e + 2];
                    const float time = sBT[t] + ite;
                    const PosDiv taud{(double)taue, *reinterpret_cast<const double*>(sPh + 8 * e + 4)};
                    const double x = phase_f64(c, time, taud, delaye, ExpLiteral());
                    // every RBF once, in registers (rbf_cols evaluates them for the sum and again for the values; same bits)
                    float h[KS];
#pragma unroll
                    for (int k = 0; k < KS; ++k) h[k] = 0.0f;
                    rbf_row<KS>(c, sCen, sCen + c.n_total, x, x * (double)c.ws, h, ExpLiteral());
#pragma unroll
                    for (int j = 0; j < KQ; ++j)
                        *reinterpret_cast<f32x4*>(row + 4 * j) = f32x4{h[4 * j], h[4 * j + 1], h[4 * j + 2], h[4 * j + 3]};
                    if (t < T - 1) sDs[idx] = scaled_time(sBT[t + 1] + ite, delaye, taue) - scaled_time(time, delaye, taue);
                }
                __builtin_amdgcn_wave_barrier();
                if (live) {
                    for (int d = 0; d < D; ++d) {
                        float x[KS];
#pragma unroll
                        for (int j = 0; j < KQ; ++j) {
                            const float4 v = *reinterpret_cast<const float4*>(sX + (e * D + d) * KS + 4 * j);
                            x[4 * j + 0] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
                        }
                        x[KS - 3] = 0.0f; x[KS - 2] = 0.0f; x[KS - 1] = 0.0f;      // goal, y0, ydot0 are not weights
                        sP[e * seg + tl * D + d] = row_chain<KQ>(row, x);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            // ---- B: 16 Euler steps of every (episode, DoF) of the chunk (one rounding per operation)
            if (on) {
                float* pp = sP + le * seg + ld;
                float* pv = sV + le * seg + ld;
                const float* pds = sDs + le * TT;
                for (int tl = 0; tl < rows; ++tl) {
                    const float f = pp[tl * D];
                    pp[tl * D] = y;
                    pv[tl * D] = div_tau(z, td);
                    if (t0 + tl < T - 1) {
                        const float ds = pds[tl];
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
            __builtin_amdgcn_wave_barrier();
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
        }
    }
}

#ifndef MPK_DEVICE_ONLY
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
        pa.wave_floats = E * pa.x_pad + 8 * E + E * 16 + 64 * KS + 2 * pa.o_pad;
        pa.vec_ok = ((reinterpret_cast<uintptr_t>(pa.pos) | reinterpret_cast<uintptr_t>(pa.vel)) & 15u) == 0 && (c.T * c.D) % 4 == 0 ? 1 : 0;
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
            const size_t shared0 = (size_t)(pa.t_pad + (c.nb + 2 + 3) / 4 * 4) * sizeof(float);
            const size_t tab_bytes = (size_t)c.n_pc * (2 * KS + 4) * sizeof(float);
            auto resident = [&](int e, bool fl) -> long {           // waves of the whole chip for this layout (as below)
                const int img_in = e * (c.P + 2 * c.D + 1), img_cols = e * (pa.x_pad + (fl ? 12 : 3));
                const size_t wb = (size_t)(2 * (((img_in > img_cols ? img_in : img_cols) + 3) / 4 * 4) + 2 * pa.o_pad) * sizeof(float);
                const bool tab = tune.phase_table != 0 && tab_bytes + 8 * wb <= 160 * 1024 - shared0 && (long)pa.B >= (long)num_cu * 8;
                int w = tab ? (int)((160 * 1024 - shared0 - tab_bytes) / wb) : (int)((64 * 1024 - shared0) / wb);
                w = tab ? (w > 16 ? 16 : w) : (w > 4 ? 4 : (w < 1 ? 1 : w));
                int pc = (int)(160 * 1024 / (wb * w + shared0 + (tab ? tab_bytes : 0)));
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
    pa.c_pad = c.mp_type == MPK_MP_PRODMP ? (c.nb + 2 + 3) / 4 * 4 : (4 * c.n_total + 6 + 3) / 4 * 4;
    const size_t wave_bytes = (size_t)pa.wave_floats * sizeof(float);
    size_t shared_bytes = (size_t)(pa.t_pad + pa.c_pad) * sizeof(float);
    if (wave_bytes + shared_bytes > 160 * 1024) return MPK_ENOTIMPL;
    int wpb = (int)((64 * 1024 - shared_bytes) / wave_bytes);
    wpb = wpb > 4 ? 4 : (wpb < 1 ? 1 : wpb);
    // prodmp: stage the row table in LDS when it leaves room for at least 8 waves ("phase_table" 0: gather from L2)
    bool lds_table = false;
    if (c.mp_type == MPK_MP_PRODMP) {
        const size_t tab_bytes = (size_t)c.n_pc * (2 * KS + 4) * sizeof(float);
        const size_t room = 160 * 1024 - shared_bytes;
        lds_table = tab_bytes + 8 * wave_bytes <= room && (long)pa.B >= (long)num_cu * 8;
        if (tune.phase_table == 0) lds_table = false;
        if (lds_table) {
            pa.tab_pad = c.n_pc * (2 * KS + 4);
            shared_bytes += tab_bytes;
            wpb = (int)((160 * 1024 - shared_bytes) / wave_bytes);
            wpb = wpb > 16 ? 16 : wpb;
        }
    }
    const size_t lds = wave_bytes * wpb + shared_bytes;
    int per_cu = (int)(160 * 1024 / lds);
    per_cu = per_cu > 32 / wpb ? 32 / wpb : per_cu;
    if (!dmp && !modelled) {
        // chunks cost balance (a wave's work is quantised in E episodes): only when every resident wave still gets >= 4
        const long resident = (long)num_cu * per_cu * wpb;
        int E = pa.chunk;
        while (E > 1 && (long)pa.B / E < 4 * resident) E >>= 1;
        if (tune.phase_chunk >= 1 && tune.phase_chunk <= pa.chunk) E = tune.phase_chunk;
        pa.chunk = E;
    }
    const long units = ((long)pa.B + pa.chunk - 1) / pa.chunk;
    long blocks = (units + wpb - 1) / wpb;
    if (blocks > (long)num_cu * per_cu) blocks = (long)num_cu * per_cu;
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = allow_full_lds(kern);
            if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * wpb), lds, (hipStream_t)stream, pa);
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    };
    switch (c.mp_type) {
        case MPK_MP_PRODMP:
            if (lds_table) {
                *kernel_name = flat ? "k_traj_phase<prodmp,lds,flat>" : "k_traj_phase<prodmp,lds>";
                if (flat) return KQ == 2 ? go(k_traj_phase<MPK_MP_PRODMP, 2, true, true>) : go(k_traj_phase<MPK_MP_PRODMP, 4, true, true>);
                return KQ == 2 ? go(k_traj_phase<MPK_MP_PRODMP, 2, true>) : go(k_traj_phase<MPK_MP_PRODMP, 4, true>);
            }
            *kernel_name = flat ? "k_traj_phase<prodmp,flat>" : "k_traj_phase<prodmp>";
            if (flat) return KQ == 2 ? go(k_traj_phase<MPK_MP_PRODMP, 2, false, true>) : go(k_traj_phase<MPK_MP_PRODMP, 4, false, true>);
            return KQ == 2 ? go(k_traj_phase<MPK_MP_PRODMP, 2, false>) : go(k_traj_phase<MPK_MP_PRODMP, 4, false>);
        case MPK_MP_PROMP:
            *kernel_name = "k_traj_phase<promp>";
            if (KQ == 1) return go(k_traj_phase<MPK_MP_PROMP, 1, false>);
            return KQ == 2 ? go(k_traj_phase<MPK_MP_PROMP, 2, false>) : go(k_traj_phase<MPK_MP_PROMP, 4, false>);
        default:
            *kernel_name = "k_traj_phase<dmp>";
            return KQ == 2 ? go(k_traj_phase_dmp<2>) : go(k_traj_phase_dmp<4>);
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
    if (lds > 160 * 1024) { set_error("trajectory too large for the per-episode kernel's LDS budget"); return MPK_EINVAL; }
    RowArgs ra{c, params, init_pos, init_vel, init_time, init_time_shared, pos, vel, range_flag, B};
    int blocks = B < num_cu * 8 ? B : num_cu * 8;
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
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
    if (rc.plant_type != MPK_PLANT_STATIC) {     // a static plant's state is an input only (callers may hold it const)
        Q[e] = q;
        QD[e] = qd;
    }
}

// sin and cos of one float64 angle with a shared three-term Cody-Waite reduction by pi/2 and the classic degree-13 /
// degree-14 kernels on [-pi/4, pi/4] (coefficients of fdlibm's __kernel_sin / __kernel_cos): ~1 ulp for |x| < 1e6, a
// quarter of the instructions of two library calls.  Larger angles (a plant spun far out of range) take the library.
__device__ __forceinline__ void sincos_lean(double x, double* sn, double* cs) {
    if (!(fabs(x) < 1.0e6)) { sincos(x, sn, cs); return; }
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;
    *sn = (q & 2) ? -a : a;
    *cs = ((q + 1) & 2) ? -b : b;
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
    // SimpleReacher reward (RW kernels): see k_reacher_rollout
    const int32_t* step0;
    const double* goal;
    double* rewards;
    int steps_before_reward;
    int wt;                  // write-through stores of the actions (cache-resident batches)
};

// NG = groups per wave: with NG = 4 a wave owns four consecutive groups and lane quarter j runs group j's recurrence,
// so four recurrences advance in parallel (the same idea as k_traj_quad); NG = 1 keeps more waves for small batches.
// RW: additionally SimpleReacherEnv's per-step reward (simple_reacher.py:56-72).  The serial lanes leave the plant
// position and the clipped action of every step of the tile in LDS as float64 (the position image reuses the desired
// pos | vel staging, which the recurrence has already pulled into registers); then all 64 lanes turn (episode, step)
// items into rewards in parallel -- cumulative joint angles, sin / cos, end effector, control cost, each summed left to
// right as numpy does.  Only the recurrence itself stays serial.
template <int NG, bool RW>
__global__ void __launch_bounds__(256) k_pd_rollout_tiles(const PdArgs a) {
    constexpr int SLOT = 3 * kStageStride + (RW ? 2 * kStageStride : 0);   // floats per group slot
    extern __shared__ __attribute__((aligned(16))) float smem[];           // [4 waves][NG][SLOT]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sSt = smem + wave * (NG * SLOT);      // per group: desired pos | desired vel | actions (| u as float64)
    const int D = a.D, T = a.T, B = a.B, SEG = 16 * D, DP = 1 << a.sh, NTW = 16 >> a.sh;
    const int col = lane & 15, bl = col >> a.sh, d = col & (DP - 1);
    const int jq = lane >> 4;                                // the group (of this wave's NG) whose recurrence the lane runs
    const bool lane_serial = jq < NG && d < D;
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
    lod = __builtin_canonicalize(lod); hid = __builtin_canonicalize(hid);   // fmin / fmax need not quiet them per step
    const double dtp = a.rc.dt;
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int units = (a.G + NG - 1) / NG;
    for (int un = vb * 4 + wave; un < units; un += gridDim.x * 4) {
        const int g0 = un * NG;
        const int bs = (g0 + jq) * NTW + bl;                 // the serial lane's episode
        const bool serial = lane_serial && g0 + jq < a.G && bs < B;
        double qs = 0.0, qds = 0.0;
        int nst = T;
        if (serial) {
            const size_t si = (size_t)bs * D + d;
            qs = a.Q[si]; qds = a.QD[si];
            if (a.n_steps) nst = min(a.n_steps[bs], T);
        }
        bool mover[NG];
        const float* gp[NG];
        const float* gv[NG];
        f32x4 lp[NG], lv[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int b0 = (g0 + j) * NTW;
            mover[j] = g0 + j < a.G && sseg < NTW && b0 + sseg < B;
            gp[j] = a.des_pos + (size_t)b0 * T * D + gofs;
            gv[j] = a.des_vel + (size_t)b0 * T * D + gofs;
            lp[j] = f32x4{0, 0, 0, 0}; lv[j] = lp[j];
            if (mover[j] && w4 < min(16, T) * D) {
                lp[j] = *reinterpret_cast<const f32x4*>(gp[j]);
                lv[j] = *reinterpret_cast<const f32x4*>(gv[j]);
            }
        }
        for (int rt = 0; rt < NRT; ++rt) {
            const int rows = min(16, T - rt * 16);
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                if (mover[j] && w4 < rows * D) {
                    *reinterpret_cast<f32x4*>(sSt + j * SLOT + rofs) = lp[j];
                    *reinterpret_cast<f32x4*>(sSt + j * SLOT + kStageStride + rofs) = lv[j];
                }
            }
            if (rt + 1 < NRT) {   // next tile's pieces travel under this tile's recurrence
                const int rows_n = min(16, T - (rt + 1) * 16);
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    if (mover[j] && w4 < rows_n * D) {
                        lp[j] = *reinterpret_cast<const f32x4*>(gp[j] + (size_t)(rt + 1) * SEG);
                        lv[j] = *reinterpret_cast<const f32x4*>(gv[j] + (size_t)(rt + 1) * SEG);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (serial) {
                // the 16 steps of the tile as straight-line code per (controller, plant): a run-time switch inside the
                // step would cost more instructions than the step's arithmetic, and this chain is the critical path
                float* sg = sSt + jq * SLOT;
                const int o0 = bl * SEG + d;
                // branch-free steps (pd_tile_steps, the closed-loop trajectory kernels' chain: a step past the executed
                // ones -- or past T in the last tile -- is computed and discarded by selects; round 2 measured 260 -> 125-180
                // cycles per step for it there); MASKED = false where every serial lane executes the whole tile
                const bool full_tile = rows == 16 && __all(nst >= rt * 16 + 16) != 0;   // over the serial lanes: wave-uniform
                auto tile_steps = [&](auto ctrl_tag, auto plant_tag) {
                    constexpr int CTRL = decltype(ctrl_tag)::value;
                    constexpr bool INTEG = decltype(plant_tag)::value == MPK_PLANT_DOUBLE_INTEGRATOR;
                    double* q64 = reinterpret_cast<double*>(sg) + col;
                    double* u64 = reinterpret_cast<double*>(sg + 3 * kStageStride) + col;
                    if (full_tile)
                        pd_tile_steps<CTRL, false, INTEG, RW>(sg + o0, sg + kStageStride + o0, sg + 2 * kStageStride + o0, D, rt * 16,
                                                              nst, pgd, dgd, lod, hid, dtp, qs, qds, q64, u64);
                    else
                        pd_tile_steps<CTRL, true, INTEG, RW>(sg + o0, sg + kStageStride + o0, sg + 2 * kStageStride + o0, D, rt * 16,
                                                             nst, pgd, dgd, lod, hid, dtp, qs, qds, q64, u64);
                };
                using std::integral_constant;
                const bool dint = a.rc.plant_type == MPK_PLANT_DOUBLE_INTEGRATOR;
                switch (a.rc.controller_type) {
                    case MPK_CTRL_MOTOR:
                        if (dint) tile_steps(integral_constant<int, MPK_CTRL_MOTOR>(), integral_constant<int, MPK_PLANT_DOUBLE_INTEGRATOR>());
                        else tile_steps(integral_constant<int, MPK_CTRL_MOTOR>(), integral_constant<int, MPK_PLANT_STATIC>());
                        break;
                    case MPK_CTRL_POSITION:
                        if (dint) tile_steps(integral_constant<int, MPK_CTRL_POSITION>(), integral_constant<int, MPK_PLANT_DOUBLE_INTEGRATOR>());
                        else tile_steps(integral_constant<int, MPK_CTRL_POSITION>(), integral_constant<int, MPK_PLANT_STATIC>());
                        break;
                    default:
                        if (dint) tile_steps(integral_constant<int, MPK_CTRL_VELOCITY>(), integral_constant<int, MPK_PLANT_DOUBLE_INTEGRATOR>());
                        else tile_steps(integral_constant<int, MPK_CTRL_VELOCITY>(), integral_constant<int, MPK_PLANT_STATIC>());
                        break;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (RW) {
                const int items = NG * NTW * rows;            // (group, episode in group, step in tile)
                for (int it = lane; it < items; it += 64) {
                    const int tl = it % rows, je = it / rows;
                    const int e = je % NTW, j = je / NTW;
                    const int b = (g0 + j) * NTW + e;
                    if (g0 + j < a.G && b < B) {
                        const int t = rt * 16 + tl;
                        const int ns = a.n_steps ? min(a.n_steps[b], T) : T;
                        double r = 0.0;
                        if (t < ns) {
                            const double* qv = reinterpret_cast<const double*>(sSt + j * SLOT) + tl * 16 + e * DP;
                            const double* uv = reinterpret_cast<const double*>(sSt + j * SLOT + 3 * kStageStride) + tl * 16 + e * DP;
                            double ang = 0.0, ex = 0.0, ey = 0.0, ctrl = 0.0;
                            for (int dd = 0; dd < D; ++dd) {
                                ang = dd == 0 ? qv[dd] : ang + qv[dd];      // np.cumsum(joint_angles)
                                double sn, cs;
                                sincos_lean(ang, &sn, &cs);
                                ex = dd == 0 ? cs : ex + cs;                // unit links (base_reacher.py:19,97-104)
                                ey = dd == 0 ? sn : ey + sn;
                                ctrl = dd == 0 ? uv[dd] * uv[dd] : ctrl + uv[dd] * uv[dd];
                            }
                            double rdist = 0.0;
                            if ((a.step0 ? a.step0[b] : 0) + t >= a.steps_before_reward) {
                                const double dx = ex - a.goal[2 * (size_t)b], dy = ey - a.goal[2 * (size_t)b + 1];
                                rdist = 0.0 - sqrt(dx * dx + dy * dy);
                            }
                            r = rdist - ctrl;
                        }
                        a.rewards[(size_t)b * T + t] = r;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (a.actions) {
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    if (mover[j] && w4 < rows * D) {
                        float* dst = a.actions + (size_t)(g0 + j) * NTW * T * D + gofs + (size_t)rt * SEG;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(sSt + j * SLOT + 2 * kStageStride + rofs);
                        if (a.wt) store16<true>(dst, v);      // cache-resident actions: write-through (wave-uniform)
                        else store16<false>(dst, v);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (serial) {
            const size_t si = (size_t)bs * D + d;
            if (a.rc.plant_type != MPK_PLANT_STATIC) { a.Q[si] = qs; a.QD[si] = qds; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_reacher_rollout: k_pd_rollout + SimpleReacherEnv's per-step reward (simple_reacher.py:56-72).  One lane per
// (episode, DoF), 64 / D episodes per wave; the reward couples an episode's DoFs (cumulative joint angles -> end
// effector, base_reacher.py:97-104), which is a segmented scan over the D neighbouring lanes.  float64 without FMA
// contraction; controller, clip and plant are the operations of k_pd_rollout (bit-exact), the scans add in tree order
// (numpy: left to right), so rewards agree with the oracle to rounding, not bit for bit.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double seg_scan(double v, int d, int D) {
    // inclusive prefix sum over the D consecutive lanes of a segment (lane's position d)
    for (int off = 1; off < D; off <<= 1) {
        const double up = __shfl_up(v, off);
        if (d >= off) v += up;
    }
    return v;
}

__device__ __forceinline__ void seg_scan3(double& a, double& b, double& c, int d, int D) {
    // three scans sharing the source-lane arithmetic and the predicate
    for (int off = 1; off < D; off <<= 1) {
        const double ua = __shfl_up(a, off), ub = __shfl_up(b, off), uc = __shfl_up(c, off);
        if (d >= off) { a += ua; b += ub; c += uc; }
    }
}

__global__ void __launch_bounds__(256) k_reacher_rollout(const RolloutDev rc, const int D,
                                                         const float* __restrict__ des_pos,
                                                         const float* __restrict__ des_vel, double* __restrict__ Q,
                                                         double* __restrict__ QD, const int32_t* __restrict__ n_steps,
                                                         const int32_t* __restrict__ step0,
                                                         const double* __restrict__ goal, const int steps_before_reward,
                                                         float* __restrict__ actions, double* __restrict__ rewards,
                                                         const int B, const int T) {
    __shared__ double s_g[4 * kMaxDofArgs];      // gains / bounds: a lane-dependent index into the kernarg arrays would
    if (threadIdx.x < (unsigned)D) {            // push the whole struct to scratch
        const double *pg = rc.pg, *dg = rc.dg, *lo = rc.lo, *hi = rc.hi;
        s_g[threadIdx.x] = pg[threadIdx.x];
        s_g[kMaxDofArgs + threadIdx.x] = dg[threadIdx.x];
        s_g[2 * kMaxDofArgs + threadIdx.x] = lo[threadIdx.x];
        s_g[3 * kMaxDofArgs + threadIdx.x] = hi[threadIdx.x];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int epw = 64 / D;                                       // episodes per wave
    const int el = lane / D, d = lane - el * D;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long bl = wave * epw + el;
    const bool on = el < epw && bl < B;
    const int b = on ? (int)bl : 0;
    const size_t sidx = (size_t)b * D + d;
    double q = on ? Q[sidx] : 0.0, qd = on ? QD[sidx] : 0.0;
    int n = n_steps ? n_steps[b] : T;
    n = !on ? 0 : (n < T ? n : T);
    const int s0 = step0 ? step0[b] : 0;
    const double gx = goal[2 * (size_t)b], gy = goal[2 * (size_t)b + 1];
    const double pg = s_g[d], dg = s_g[kMaxDofArgs + d], lo = s_g[2 * kMaxDofArgs + d], hi = s_g[3 * kMaxDofArgs + d];
    const double dt = rc.dt;
    int nmax = n;                                                 // the wave runs to its longest episode
    for (int m = 32; m >= 1; m >>= 1) nmax = max(nmax, __shfl_xor(nmax, m));
    const size_t base = (size_t)b * T * D + d;
    constexpr int kAhead = 8;                                     // desired states are fetched 8 steps at a time: one
    for (int t0 = 0; t0 < nmax; t0 += kAhead) {                  // memory round trip per 8 serial steps, not per step
        float dpv[kAhead], dvv[kAhead];
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            const bool ld = t0 + i < n;
            dpv[i] = ld ? des_pos[base + (size_t)(t0 + i) * D] : 0.0f;
            dvv[i] = ld ? des_vel[base + (size_t)(t0 + i) * D] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            const int t = t0 + i;
            if (t >= nmax) break;
            const bool live = t < n;
            double u = 0.0;
            if (live) {
                const double dp = (double)dpv[i], dv = (double)dvv[i];
                if (rc.controller_type == MPK_CTRL_MOTOR) u = pg * (dp - q) + dg * (dv - qd);
                else if (rc.controller_type == MPK_CTRL_POSITION) u = dp;
                else u = dv;
                u = fmin(fmax(u, lo), hi);
                qd = qd + dt * u;                  // base_reacher_torque.py:25-26
                q = q + dt * qd;
                if (actions) actions[base + (size_t)t * D] = (float)u;
            }
            const double ang = seg_scan(q, d, D);               // np.cumsum(joint_angles)
            double sn, cs;
            sincos_lean(ang, &sn, &cs);
            double ex = cs, ey = sn, ctrl = u * u;              // unit link lengths (base_reacher.py:19): sums over the links
            seg_scan3(ex, ey, ctrl, d, D);
            if (live && d == D - 1) {
                double rdist = 0.0;
                if (s0 + t >= steps_before_reward) {
                    const double dx = ex - gx, dy = ey - gy;
                    rdist = 0.0 - sqrt(dx * dx + dy * dy);
                }
                rewards[(size_t)b * T + t] = rdist - ctrl;
            }
        }
    }
    if (on) {
        for (int t = n; t < T; ++t) {
            if (actions) actions[base + (size_t)t * D] = 0.0f;
            if (d == D - 1) rewards[(size_t)b * T + t] = 0.0;
        }
        Q[sidx] = q; QD[sidx] = qd;
    }
}

#ifndef MPK_DEVICE_ONLY
int launch_reacher_rollout(const RolloutDev& rc, int D, const float* des_pos,
                           const float* des_vel, double* q, double* qd, const int32_t* n_steps, const int32_t* step0,
                           const double* goal, int steps_before_reward, float* actions, double* rewards, int B, int T,
                           void* stream, const Tuning& tune) {
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int last_rows = T - (T - 1) / 16 * 16;
    const bool tiles_ok = D >= 1 && D <= kMaxD && (T * D) % 4 == 0 && (last_rows * D) % 4 == 0 && aligned16(des_pos) &&
                          aligned16(des_vel) && (!actions || aligned16(actions)) && tune.pd_simple != 1;
    if (tiles_ok) {
        // the tile-streaming rollout with the reward evaluated per tile by all lanes (see k_pd_rollout_tiles, RW)
        PdArgs pa;
        pa.rc = rc; pa.des_pos = des_pos; pa.des_vel = des_vel; pa.Q = q; pa.QD = qd; pa.n_steps = n_steps;
        pa.actions = actions; pa.D = D; pa.B = B; pa.T = T;
        pa.wt = (double)B * T * D * 12.0 <= 96.0 * 1024 * 1024 ? 1 : 0;     // desired (pos, vel) + actions stay cached
        if (tune.write_through >= 0) pa.wt = tune.write_through != 0 ? 1 : 0;
        pa.step0 = step0; pa.goal = goal; pa.rewards = rewards; pa.steps_before_reward = steps_before_reward;
        int sh = 0;
        while ((1 << sh) < D) ++sh;
        pa.sh = sh;
        const int NTW = 16 >> sh;
        pa.G = (B + NTW - 1) / NTW;
        pa.inv_seg4 = 65536u / (unsigned)(4 * D) + 1u;
        const int quad_mode = tune.pd_quad < 0 ? 1 : tune.pd_quad;
        const bool quad = quad_mode == 2 || (quad_mode == 1 && pa.G >= 4 * 256 * 4);   // measured (profiles/r03_rollout.md): 19.3 vs 23.9 us at 4096 groups, 15.6 vs 14.6 at 2048
        const int units = quad ? (pa.G + 3) / 4 : pa.G;
        int blocks = (units + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
        const size_t lds = (size_t)4 * (quad ? 4 : 1) * 5 * kStageStride * sizeof(float);
        auto go = [&](auto kern) -> int {
            if (lds > 64 * 1024) {
                hipError_t e = allow_full_lds(kern);
                if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
            }
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, (hipStream_t)stream, pa);
            MPK_LAUNCH_CHECK();
            return MPK_OK;
        };
        return quad ? go(k_pd_rollout_tiles<4, true>) : go(k_pd_rollout_tiles<1, true>);
    }
    const int epw = 64 / D;
    const long waves = ((long)B + epw - 1) / epw;
    hipLaunchKernelGGL(k_reacher_rollout, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rc, D,
                       des_pos, des_vel, q, qd, n_steps, step0, goal, steps_before_reward, actions, rewards, B, T);
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

#ifndef MPK_DEVICE_ONLY
int launch_pd_rollout(const RolloutDev& rc, int D, const float* des_pos, const float* des_vel, double* q, double* qd,
                      const int32_t* n_steps, float* actions, int B, int T, void* stream, const Tuning& tune) {
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int last_rows = T - (T - 1) / 16 * 16;
    const bool tiles_ok = D >= 1 && D <= kMaxD && (T * D) % 4 == 0 && (last_rows * D) % 4 == 0 && aligned16(des_pos) &&
                          aligned16(des_vel) && (!actions || aligned16(actions)) && tune.pd_simple != 1;
    if (tiles_ok) {
        PdArgs pa;
        pa.rc = rc; pa.des_pos = des_pos; pa.des_vel = des_vel; pa.Q = q; pa.QD = qd; pa.n_steps = n_steps;
        pa.actions = actions; pa.D = D; pa.B = B; pa.T = T;
        pa.wt = (double)B * T * D * 12.0 <= 96.0 * 1024 * 1024 ? 1 : 0;     // desired (pos, vel) + actions stay cached
        if (tune.write_through >= 0) pa.wt = tune.write_through != 0 ? 1 : 0;
        pa.step0 = nullptr; pa.goal = nullptr; pa.rewards = nullptr; pa.steps_before_reward = 0;
        int sh = 0;
        while ((1 << sh) < D) ++sh;
        pa.sh = sh;
        const int NTW = 16 >> sh;
        pa.G = (B + NTW - 1) / NTW;
        pa.inv_seg4 = 65536u / (unsigned)(4 * D) + 1u;
        // four groups per wave once that still leaves every CU several waves ("pd_quad": 0 off, 2 force)
        const int quad_mode = tune.pd_quad < 0 ? 1 : tune.pd_quad;
        const bool quad = quad_mode == 2 || (quad_mode == 1 && pa.G >= 4 * 256 * 4);   // measured (profiles/r03_rollout.md): 19.3 vs 23.9 us at 4096 groups, 15.6 vs 14.6 at 2048
        const int units = quad ? (pa.G + 3) / 4 : pa.G;
        int blocks = (units + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
        const size_t lds = (size_t)4 * (quad ? 4 : 1) * 3 * kStageStride * sizeof(float);
        if (quad) hipLaunchKernelGGL((k_pd_rollout_tiles<4, false>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, pa);
        else hipLaunchKernelGGL((k_pd_rollout_tiles<1, false>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, pa);
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
#endif  // MPK_DEVICE_ONLY

// ------------------------------------------------------------------------------------------------------------
// integer replanning state
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_replan_advance(const ReplanDev rp, const int T, const int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    (void)replan_rule(rp, b, T, true);
}

#ifndef MPK_DEVICE_ONLY
int launch_replan_advance(int32_t* traj_steps, int32_t* plan_steps, int32_t* seg_len, uint8_t* done, int every,
                          int max_planning_times, int horizon, int T, int B, void* stream) {
    ReplanDev rp;
    rp.traj_steps = traj_steps; rp.plan_steps = plan_steps; rp.seg_len = seg_len; rp.done = done;
    rp.every = every; rp.max_planning_times = max_planning_times; rp.horizon = horizon;
    hipLaunchKernelGGL(k_replan_advance, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, rp, T, B);
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

#endif  // MPK_MAIN

}  // namespace mpk

#ifdef MPK_TRACE
// development builds only: fetch and clear the stamps (pairs of tag, shader clock)
extern "C" int mpk_debug_trace(long long* out, int cap) {
    // out: (tag, clock) pairs of the slots stamped since the last call, sorted by clock; returns their number
    long long raw[256];
    if (hipMemcpyFromSymbol(raw, HIP_SYMBOL(mpk::g_trace), sizeof(raw)) != hipSuccess) return -1;
    int n = 0;
    for (int t = 0; t < 256 && n < cap; ++t)
        if (raw[t] != 0) { out[2 * n] = t; out[2 * n + 1] = raw[t]; ++n; }
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && out[2 * j + 1] < out[2 * j - 1]; --j) {
            const long long t0 = out[2 * j], c0 = out[2 * j + 1];
            out[2 * j] = out[2 * j - 2]; out[2 * j + 1] = out[2 * j - 1];
            out[2 * j - 2] = t0; out[2 * j - 1] = c0;
        }
    static const long long zeros[256] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mpk::g_trace), zeros, sizeof(zeros));
    return n;
}
#endif
