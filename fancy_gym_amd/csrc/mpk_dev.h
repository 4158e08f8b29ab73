// Shared device helpers of the gfx950 kernels of libmpk.so: trace stamps, launch check, LDS attribute helper, the float64 /
// fp32 scalar helpers every kernel family uses (reciprocal divisions, lean exp, phase, RBF rows, ProDMP columns) and the
// integer replanning rule.  Compiled with -ffp-contract=off: every fused multiply-add is an explicit fmaf()/MFMA, every
// other a*b+c rounds twice exactly like the reference's separate torch / numpy ops.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "mpk_internal.h"

namespace mpk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Development build only (-DMPK_TRACE): wave 0 of workgroup `MPK_TRACE_BLOCK` stamps the shader clock at labelled points
// of a kernel into a device array that tools/dev/trace_kernel.py prints -- the per-phase timeline of ONE wave.
#ifdef MPK_TRACE
#if !defined(MPK_AMALGAMATED) && !defined(MPK_TRACE_UNIT)
#error "MPK_TRACE builds are single translation unit builds of mpk_kernels.hip, or of ONE unit (MPK_TRACE_UNIT): the trace buffer is one device variable"
#endif
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
}  // namespace mpk
#if defined(MPK_TRACE_UNIT) && !defined(MPK_DEVICE_ONLY)
#include "mpk_trace_reader.h"
#endif
namespace mpk {
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

// ---- the size rules of every launcher, in one place: what is fitted to the caches and the LDS of an MI355X -----------------------
// LDS: 160 KB per CU; a workgroup may take more than kLdsDefault only after allow_full_lds(); the whole-trajectory-image kernels
// leave room for two workgroups per CU (kLdsHalf)
constexpr size_t kLdsPerCu = 160 * 1024;
constexpr size_t kLdsDefault = 64 * 1024;
constexpr size_t kLdsHalf = 80 * 1024;
// Outputs of a launch up to kCachedBytes stay in the L2s + the memory-side cache whatever the store order: the tile-major kernels
// (best load balance, scattered 64-byte runs) keep those launches; above, the episode-major kernels write whole-trajectory runs
// (profiles/r02_streaming.md: the cross-over sits between 64 and 128 MB for every configuration measured)
constexpr double kCachedBytes = 96.0 * 1024 * 1024;
// Write-through (sc1) stores are chosen while a launch's outputs still fit the memory-side cache (256 MB + the L2s): plain
// write-back stores fall off a cliff once the dirty lines exceed it, write-through stores lose once the outputs stream to
// HBM anyway.  Measured per kernel family and size (profiles/r03_streaming_wt.md, us plain vs write-through): k_traj_flat
// +actions 57.1 / 46.1 at 241 MB, 67.5 / 51.9 at 275 MB, 74.3 / 71.3 at 310 MB, 77.7 / 88.6 at 344 MB; cfg3 k_traj_quad<dmp>
// 36.5 / 33.6 at 175 MB, 58.2 / 54.3 at 262 MB, 99 / 104 at 350 MB; closed loop k_traj_duo 80.8 / 70.1 at 262 MB, 173 / 221 at
// 525 MB; per-episode kernels 44.2 / 41.2 at 175 MB, 107 / 117 at 350 MB.
constexpr double kWtBytes = 300.0 * 1024 * 1024;
// k_traj_ring (wave-specialised store engine, in-order batch tickets) takes the open-loop launches that write more than this:
// B = 262144 (2.2 GB): 386 us against k_traj_flat's 417 - 447 on the same boxes; B = 65536 +actions (550 MB): 112 - 115 against
// 115 - 121; trajectory only at 65536 (367 MB): 83 against 80 - 82 -- below that the persistent one-workgroup-per-CU launch has
// too few batches per CU to amortise its ramp (profiles/r04_ring.md).  Second session, one call over nine sizes
// (profiles/r04_open_loop_choice.md; us, flat / ring): +actions 49 152 (424 MB) 92.8 / 91.2, 65 536 (565 MB) 123.4 / 118.1,
// 81 920: 153.1 / 144.2; trajectory only 65 536 (382 MB) 87.0 / 86.0, 81 920 (477 MB) 105.8 / 100.1, 98 304 (573 MB) 124.1 / 118.2;
// 32 768 (283 / 191 MB): 55.9 / 71.2 and 33.6 / 46.9 -- the ring from 440 MB on (was 600 MiB)
// Round 5: the ring's "fixed" 33 - 38 us below ~30 000 episodes was its ticket size (wave 0 takes 3 - 4 tickets of five batches in its first
// atomic: a few dozen workgroups walked off with a small launch); with at least ~16 tickets per workgroup (mpk_traj_launch.hip) the
// ring takes 21 us at 12 288 episodes (was 38; tiles 14.6) and crosses k_traj_flat earlier (profiles/r05_open_loop_choice.md; us, flat /
// ring): trajectory only 57 344 (321 MB) 76.3 / 76.6, 65 536 (367 MB) 85.5 / 82.4, 81 920: 105.2 / 99.6; + actions 40 960 (353 MB)
// 77.2 - 80.4 / 77.0, 49 152 (424 MB) 92.7 / 87.9 -- the ring from 346 MB on (was 440)
constexpr double kRingBytes = 330.0 * 1024 * 1024;
// ... and trajectory-only launches of the shapes k_traj_flat takes: the flat kernel at two workgroups per CU stays ahead of the ring up to
// ~3 GB (round 5; mpk_traj_launch.hip)
constexpr double kRingTrajBytes = 4096.0 * 1024 * 1024;
// ... and they leave the tile-major kernel for it from here on (instead of kCachedBytes)
constexpr double kFlatTrajBytes = 64.0 * 1024 * 1024;
// a ticket of k_traj_ring covers at least this many bytes of batch buffers: one device counter hands out ~88 tickets / us
// (profiles/r04_store_engine_probe_dynamic.md), 32768 tickets of 67 KB saturate it, 10923 of 201 KB do not
constexpr size_t kRingTicketBytes = 192 * 1024;
// closed loop: the ring with consumer waves (k_traj_ring<.., closed>) is automatic once the step's three output arrays exceed this:
// where k_traj_quad's two waves per SIMD stop paying (us, full horizon / replanning step, profiles/r04_ring_closed.md):
//   32 768 episodes (275 MB)  quad 54.8 / 51.8   ring 63.6 / 57.1        36 864 (310 MB)  quad 73.5 / 72.2   ring 69.5 / 65.2
//   34 816 (292 MB)           quad 65.0 / 60.7   ring 66.8 / 58.0        40 960 (344 MB)  quad 92.6, duo 127.5   ring 84.5 / 76.9
constexpr double kRingClosedBytes = 295.0 * 1024 * 1024;

// Kernels that may take more than the default 64 KB of dynamic LDS: the function attribute is raised ONCE per kernel
// instantiation to the CU's whole LDS (160 KB), not per launch with the launch's size -- hipFuncSetAttribute rewrites state of a
// function whose earlier launches may still be in flight (round 3: one silent runtime abort per ~30 000 launches of mixed
// configurations in the fuzz soak went away with this).
#ifndef MPK_DEVICE_ONLY
// keyed on the function's ADDRESS and the device (round 3 kept the flag in a function-local static of a template over the
// kernel's TYPE: every instantiation with the same parameter list shared one flag, so only the first of them ever had the
// attribute raised -- ADVICE r03)
static hipError_t allow_full_lds_addr(const void* fn) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, dev})) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
    if (e == hipSuccess) done.insert({fn, dev});
    return e;
}
template <class K>
static hipError_t allow_full_lds(K kern) { return allow_full_lds_addr(reinterpret_cast<const void*>(kern)); }
#endif

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


// The integer part of BlackBoxWrapper.step's loop (black_box_wrapper.py:174,197,206) for one episode and one plan:
// how many steps this plan executes before the loop breaks (end of the horizon, or the schedule t % every == 0 while
// plan_steps < max_planning_times), and the counters after it.  k_replan_advance and the fused closed-loop kernels both
// call this, `writer` = the one lane per episode that stores the new state.
struct ReplanVals { int seg, cur, plan; bool was_done; };
// the rule itself: loads + arithmetic, nothing written
__device__ __forceinline__ ReplanVals replan_eval(const ReplanDev& rp, int b, int T) {
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
    return ReplanVals{seg, cur, plan, was_done != 0};
}
// ... and the state after the plan.  `valid` false (the validity gate of mpk_replan_step_gated: an invalid plan finishes its episode
// without executing a step, black_box_wrapper.py:169-172): done = 1, nothing else moves, seg_len = 0.  Returns the executed steps.
__device__ __forceinline__ int replan_write(const ReplanDev& rp, int b, const ReplanVals& v, bool valid = true) {
    const int seg = valid ? v.seg : 0;
    rp.seg_len[b] = seg;
    if (!v.was_done) {
        const uint8_t dn = (!valid || (v.cur + seg) >= rp.horizon) ? 1 : 0;
        if (valid) {
            rp.plan_steps[b] = v.plan;
            rp.traj_steps[b] = v.cur + seg;
        }
        rp.done[b] = dn;
        if (rp.done_out) rp.done_out[b] = dn;
    } else if (rp.done_out) {
        rp.done_out[b] = 1;
    }
    return seg;
}
__device__ __forceinline__ int replan_rule(const ReplanDev& rp, int b, int T, bool writer) {
    const ReplanVals v = replan_eval(rp, b, T);
    if (writer) replan_write(rp, b, v);
    return v.seg;
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

// first Taylor coefficient used: 3 = 1/11! (remainder < 7e-15 relative), 6 = 1/8! (remainder r^9 / 9! < 2e-10 for |r| <= ln2 / 2:
// still 300 times below the fp32 rounding that follows every use of these rows) -- A/B build knob (profiles/r04_per_episode_dmp.md)
#ifndef MPK_EXP_FIRST
#define MPK_EXP_FIRST 6
#endif
template <class CF>
__device__ __forceinline__ double exp_nonpos(double x, const CF& cf) {
    x = fmax(x, -700.0);                                        // exp(-700) ~ 1e-304: still normal, rounds to 0.0f
    // (arguments are <= 0 everywhere but in RbfRecur's ratio, which stays far below the overflow threshold)
    const double n = rint(x * cf[0]);
    double r = fma(n, cf[1], x);
    r = fma(n, cf[2], r);
    double p = cf[MPK_EXP_FIRST];
#pragma unroll
    for (int i = MPK_EXP_FIRST + 1; i < 15; ++i) p = fma(p, r, cf[i]);
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

// ---- hand-over between the waves of a workgroup through monotonic LDS counters (k_pd_rollout_tiles<.., helper>, k_phase_fused<.., pipe>).
// A wave's DS operations retire in order and all 64 lanes issue together: whoever reads a counter sees what its writer stored to LDS
// before it.  Relaxed workgroup-scope atomics keep the compiler from caching or waiting (a release fence would make it wait for every
// outstanding global load and store as well); the empty asm statements keep it from moving LDS accesses across them.  Every spin is
// bounded; a wave that gives up says so in the handle's fault word (host memory; codes: k_traj_ring's 1 .. 32, 64 reward helper,
// 128 k_phase_fused pipeline) -- the next entry point on the handle turns it into MPK_EHIP.
constexpr unsigned kFlagSpinLimit = 1u << 21;
__device__ __noinline__ void wave_gave_up(int* fault, int code) {
    if (fault) __hip_atomic_fetch_or(fault, code << 8 | 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ int flag_load(const int* p) {
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void flag_store(int* p, int v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// until *p >= want; false (and the fault word raised) when the spin limit passes
__device__ __forceinline__ bool flag_wait(const int* p, const int want, int* fault, const int code) {
    unsigned spins = 0;
    while (flag_load(p) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kFlagSpinLimit) { wave_gave_up(fault, code); return false; }
    }
    return true;
}

}  // namespace mpk
