// SimpleReacherEnv's per-step reward (envs/classic_control/simple_reacher/simple_reacher.py:56-72) on float64 images in LDS: what the
// rollout kernel with reward (mpk_rollout.hip) and the episode-return kernel (mpk_episode.hip) share.
#pragma once
#include "mpk_tile.h"

namespace mpk {

// sin and cos of one float64 angle with a shared three-term Cody-Waite reduction by pi/2 and the classic degree-13 /
// degree-14 kernels on [-pi/4, pi/4] (coefficients of fdlibm's __kernel_sin / __kernel_cos): ~1 ulp for |x| < 1e6, a
// quarter of the instructions of two library calls.  Larger angles (a plant spun far out of range) take the library.
__device__ __forceinline__ void sincos_core(double x, double* sn, double* cs) {   // |x| < 1e6, no branch
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

__device__ __forceinline__ void sincos_lean(double x, double* sn, double* cs) {
#ifndef MPK_NO_BIG_SINCOS
    if (!(fabs(x) < 1.0e6)) { sincos(x, sn, cs); return; }
#endif
    sincos_core(x, sn, cs);
}

// SimpleReacherEnv's reward of ONE (episode, step) item (simple_reacher.py:56-72; unit links: base_reacher.py:19,97-104) from the
// plant positions qv[0 .. D) after the step and the clipped actions uv[0 .. D): cumulative joint angles, end effector, control
// cost, every sum left to right as numpy adds.  DC > 0: the link count compiled in, the sin / cos evaluations (independent chains
// of ~20 dependent float64 operations each) unrolled side by side -- an A/B knob (MPK_RW_DC), off: measured slower than the
// run-time loop.  Same operations either way: same bits.
#ifndef MPK_RW_CHAINS
#define MPK_RW_CHAINS 3
#endif
#ifndef MPK_RW_DC
#define MPK_RW_DC 0          // 5: the five links of SimpleReacher unrolled, MPK_RW_CHAINS sin / cos chains side by side.  Measured
                             // SLOWER (65 536 episodes x 200 steps: 353 us against 322 for the run-time loop, profiles/r04_reward_pass_ab.md):
                             // the pass is bound by float64 issue (two waves per SIMD both inside it), not by the chains' latency
#endif
#ifndef MPK_RW_LATE
#define MPK_RW_LATE 1        // the control-cost pass of a tile a tile late, inside the next tile's staging (k_pd_rollout_tiles; 0: right after the chain)
#endif
#ifndef MPK_RW_DC_SMALL
#define MPK_RW_DC_SMALL 1    // the unrolled five-link chains in the kernels with one / two groups per wave only (latency-bound launches)
#endif
#ifndef MPK_PD_LOOK
#define MPK_PD_LOOK 2        // tiles of input lookahead of the rollout kernel without reward, one / two groups per wave: 3 measured 1 - 4 % SLOWER
                             // than 2 (4 096: 11.5 vs 11.4 us, 8 192: 17.5 vs 16.8): what a tile's staging costs is its instructions, not a late load
#endif
#ifndef MPK_RW_ALWAYS_TRIG
#define MPK_RW_ALWAYS_TRIG 0 // 1: round 4's reward pass (the sin / cos chains for every item, used or not) -- A/B build knob
#endif
#ifndef MPK_RW_HELPER
// reward kernels: the control-cost pass on two HELPER waves of a six-wave workgroup ("pd_helper" 1).  It lost every A/B of round 5
// (profiles/r05_rollout.md), so release builds do not carry its nine instantiations any more (round 6): built only with -DMPK_ABLATIONS
// (MPK_EXTRA_FLAGS=-DMPK_ABLATIONS MPK_BUILD_OUT=ab/lib_ablations.so python __graft_entry__.py --force); without it "pd_helper" 1 runs
// the pass on the chain waves -- same results, bit for bit
#ifdef MPK_ABLATIONS
#define MPK_RW_HELPER 1
#else
#define MPK_RW_HELPER 0
#endif
#endif
#ifndef MPK_RW_LOOK
#define MPK_RW_LOOK 1        // tiles of input lookahead in the reward kernel (2 = as the kernel without reward: measured slower at every size
                             // even after round 5 took the sin / cos chains out of 199 of 200 steps -- 4 096 episodes 29.7 vs 27.0 us,
                             // 16 384: 81 vs 51.5 (four groups per wave: 256 registers, one wave per SIMD); profiles/r05_rollout.md)
#endif
// qv / uv: the item's first column of the [column][step] float64 images, at its step: DoF dd is 16 doubles further on.
constexpr int kRwCol = 16;     // doubles between neighbouring columns of the float64 images
template <int DC>
__device__ __forceinline__ double reacher_reward_item(const double* qv, const double* uv, const int D, const bool dist_on,
                                                      const double gx, const double gy) {
    double ex = 0.0, ey = 0.0, ctrl = 0.0;
    if constexpr (DC > 0) {
        double ang[DC];
        bool big = false;
#pragma unroll
        for (int dd = 0; dd < DC; ++dd) {
            const double qd_ = qv[dd * kRwCol];
            ang[dd] = dd == 0 ? qd_ : ang[dd > 0 ? dd - 1 : 0] + qd_;           // np.cumsum(joint_angles)
            big = big || !(fabs(ang[dd]) < 1.0e6);
        }
        if (big) {
#pragma unroll 1
            for (int dd = 0; dd < DC; ++dd) {
                double sn, cs;
                sincos_lean(ang[dd], &sn, &cs);
                ex = dd == 0 ? cs : ex + cs;
                ey = dd == 0 ? sn : ey + sn;
            }
        } else {
            // CH chains side by side fill the issue slots of a wave that shares its SIMD with one other; all DC at once cost 70 more
            // registers than two waves per SIMD leave
            constexpr int CH = MPK_RW_CHAINS;
#pragma unroll
            for (int d0 = 0; d0 < DC; d0 += CH) {
                double sn[CH], cs[CH];
#pragma unroll
                for (int dd = d0; dd < (d0 + CH < DC ? d0 + CH : DC); ++dd) sincos_core(ang[dd], &sn[dd - d0], &cs[dd - d0]);
#pragma unroll
                for (int dd = d0; dd < (d0 + CH < DC ? d0 + CH : DC); ++dd) {
                    ex = dd == 0 ? cs[dd - d0] : ex + cs[dd - d0];
                    ey = dd == 0 ? sn[dd - d0] : ey + sn[dd - d0];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int dd = 0; dd < DC; ++dd) {
            const double u_ = uv[dd * kRwCol];
            ctrl = dd == 0 ? u_ * u_ : ctrl + u_ * u_;
        }
    } else {
        double ang = 0.0;
#pragma unroll 1
        for (int dd = 0; dd < D; ++dd) {
            ang = dd == 0 ? qv[dd * kRwCol] : ang + qv[dd * kRwCol];
            double sn, cs;
            sincos_lean(ang, &sn, &cs);
            ex = dd == 0 ? cs : ex + cs;
            ey = dd == 0 ? sn : ey + sn;
            ctrl = dd == 0 ? uv[dd * kRwCol] * uv[dd * kRwCol] : ctrl + uv[dd * kRwCol] * uv[dd * kRwCol];
        }
    }
    double rdist = 0.0;
    if (dist_on) {
        const double dx = ex - gx, dy = ey - gy;
        rdist = 0.0 - sqrt(dx * dx + dy * dy);
    }
    return rdist - ctrl;
}

// The same item where the reference adds no distance term (simple_reacher.py:62-63: `if self._steps >= self.steps_before_reward`,
// 199 of an episode's 200 steps at the reference's setting, :31): 0 - sum(action ** 2), the sum left to right -- the control cost of
// reacher_reward_item operation for operation, so a pass may take either function for an item with dist_on == false: same bits.
// Round 5: the pass ran the D sin / cos chains for every item and dropped 199 of 200 results behind a run-time predicate the compiler
// cannot hoist (review of round 4); now a pass evaluates them only when at least one of its 64 items is past steps_before_reward.
// DC > 0: the DoF count compiled in -- all DC reads of the float64 image issued together, one wait; the run-time loop waits for
// every read in turn (trace, round 5: 820 of a tile's 4 070 cycles at five DoF, 130 per read)
template <int DC>
__device__ __forceinline__ double reacher_ctrl_item(const double* uv, const int D) {
    double ctrl = 0.0;
    if constexpr (DC > 0) {
        double u_[DC];
#pragma unroll
        for (int dd = 0; dd < DC; ++dd) u_[dd] = uv[dd * kRwCol];
#pragma unroll
        for (int dd = 0; dd < DC; ++dd) ctrl = dd == 0 ? u_[dd] * u_[dd] : ctrl + u_[dd] * u_[dd];
    } else {
        for (int dd = 0; dd < D; ++dd) ctrl = dd == 0 ? uv[dd * kRwCol] * uv[dd * kRwCol] : ctrl + uv[dd * kRwCol] * uv[dd * kRwCol];
    }
    return 0.0 - ctrl;
}
__device__ __forceinline__ double reacher_ctrl_item_d(const double* uv, const int D) {      // (D is wave-uniform)
    switch (D) {
        case 2: return reacher_ctrl_item<2>(uv, D);     // SimpleReacher-v0 (envs/__init__.py:41-48: n_links 2)
        case 3: return reacher_ctrl_item<3>(uv, D);
        case 4: return reacher_ctrl_item<4>(uv, D);
        case 5: return reacher_ctrl_item<5>(uv, D);     // LongSimpleReacher-v0 (envs/__init__.py:52-59: n_links 5)
        case 6: return reacher_ctrl_item<6>(uv, D);
        case 7: return reacher_ctrl_item<7>(uv, D);
        case 8: return reacher_ctrl_item<8>(uv, D);
        default: return reacher_ctrl_item<0>(uv, D);
    }
}

}  // namespace mpk
