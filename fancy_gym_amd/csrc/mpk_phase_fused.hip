// k_phase_fused: the one-launch entry points of a step -- mpk_trajectory_actions, mpk_trajectory_rollout, mpk_replan_step(_gated),
// mpk_episode_return(_gated) -- for configurations with a PER-EPISODE phase: learned tau / delay (round 6).
//
// The reference registers three such families: TableTennis-ProDMP (envs/mujoco/table_tennis/mp_wrapper.py:32-57: learn tau + delay,
// 7 DoF x 3 basis, alpha 25, T = 350), its Replan variant (:91-121: 2 basis + goal, `t % 50 == 0`, max_planning_times 3) and
// BeerPong-ProMP (envs/mujoco/beerpong/mp_wrapper.py:9-25: learn tau, 2 basis + 2 zero-start, T = 300).  Until round 5 every fused entry
// point declined them (shared-phase matrix-core kernels only), so a step was k_traj_phase + k_pd_rollout (+ k_replan_advance,
// k_validity, k_condition_gather) with the trajectory re-read from HBM: 180 us for 8 192 episodes of TT-ProDMP where the plan alone
// takes 30 (profiles/r06_before_sweep.md).
//
// Decomposition -- k_traj_phase_dmp's, because the controller / plant loop is the same kind of serial recurrence as DMP's Euler map:
// a wave owns a CHUNK of E consecutive episodes (E D <= 64) and walks the horizon in tiles of 16 steps:
//   A  lane <-> (episode, step of the tile): the step's basis row -- ProDMP: table index with the exact reciprocal divisions, row gather,
//      boundary-condition factors in float64; ProMP: float64 phase + RBF row -- and the D contractions as fmaf chains in ascending k:
//      THE functions and the operation order of k_traj_phase (mpk_phase.h), so plans come out bit for bit as from mpk_trajectory;
//      results land in the chunk's LDS images [episode][step][DoF] (the layout of the outputs in HBM);
//   B  lane <-> (episode, DoF): the 16 steps of the tracking controller + clip (+ double-integrator plant) as the float64 register chain
//      of every closed-loop kernel (pd_tile_steps, mpk_tile.h: numpy's promotion in controller/pd_controller.py:21-29, no FMA) -- all
//      E D recurrences of the chunk at once; the same lanes hold the integer replanning state (replan_rule), gather the boundary
//      condition of the next plan (condition_on_desired) and test the plan against the joint limits (validity gate, below);
//   C  the tile's (pos | vel | actions) runs of 16 D floats per episode leave as float4 stores (images are staged at the 16-byte phase
//      of their destination: T D need not be a multiple of 4 -- 350 x 7 is not); skipped altogether for the verbose < 2 step.
// ProMP's velocity is the forward difference of its positions: the lanes of a tile evaluate the rows of steps t0 + 1 .. t0 + 16 and take
// p[t0] from the tile before (an LDS carry per (episode, DoF)), so a tile needs no seventeenth row and no second pass.
//
// Validity gate (preprocessing_and_validity_callback between plan and rollout, black_box_wrapper.py:155-172; TableTennisEnv.
// check_traj_validity / _get_traj_invalid_penalty, table_tennis_env.py:282-309): the chain lanes read every desired position of their
// column anyway; they test it against the joint limits (exact fp32 thresholds: the float64 comparison of the reference, decided in one
// v_med3_f32) and -- only in a tile that holds a violation -- add max(pos - high, 0) / max(low - pos, 0) in float64.  The rollout runs
// SPECULATIVELY; a plan that turns out invalid is rolled back at the end of its chunk: plant state not written, integer state =
// "finished without a step" (replan_write), condition = row 0, its action rows rewritten as zeros.  Valid plans -- the common case --
// pay one compare per position.
#include "mpk_phase.h"

namespace mpk {

struct FusedArgs {
    DevCfg c;
    const float* params;
    const float* init_pos;
    const float* init_vel;
    float init_time_shared;
    float* pos;                 // [B, T, D] outputs: all three or none (none: the verbose < 2 step, nothing per step is stored)
    float* vel;
    float* actions;
    double* q;                  // closed loop: plant state [B, D] in / out; static plant: the frozen state (c_pos, c_vel), read only
    double* qd;
    const int32_t* n_steps;     // [B] executed steps (NULL = T) when there is no replanning state
    ReplanDev rp;
    double plant_dt;
    int32_t* flag;              // ProDMP: raised when a scaled time leaves the pre-computed table (mpk_check_range)
    // validity gate (gate != 0)
    const float* raw_params;    // [B, P] the action as the caller passed it (tau / delay neither clipped nor frozen)
    uint8_t* valid;             // [B] out
    double* penalty;            // [B] out, optional
    int gate, check_td;
    double tau_b[2], delay_b[2];
    double* ret;                // [B] out, optional: the aggregated reward of the verbose < 2 step (no device reward here: 0)
    int32_t* seg_out;           // [B] out, optional: executed steps
    int B, chunk, x_pad, pitch, wave_floats, t_pad, c_pad, tab_pad, car_pad, wt, vec_ok, td3;
    int nsplit, split_tiles;    // frozen-state actions: a chunk's tiles in nsplit units of split_tiles tiles (one unit per wave trip)
    unsigned inv_d, inv_ch;     // 65536 / D + 1, 65536 / (pitch / 4) + 1: lane / D and job / chunks-per-image as multiply-high
    int* fault;                 // the handle's fault word (PIPE: a wave that gives up waiting for its partner says so)
};

// controller constants and joint limits per DoF (second kernel argument: read with per-lane loads from the kernel-argument segment)
struct FusedLim {
    double pg[kMaxD], dg[kMaxD], lo[kMaxD], hi[kMaxD];
    double glo[kMaxD], ghi[kMaxD];      // joint limits as the reference holds them
    float glo32[kMaxD], ghi32[kMaxD];   // smallest fp32 >= low, largest fp32 <= high: pos in [low, high] <=> pos in [glo32, ghi32]
};
constexpr size_t kFusedLimOffset = (sizeof(FusedArgs) + alignof(FusedLim) - 1) / alignof(FusedLim) * alignof(FusedLim);

template <class T>
__device__ __forceinline__ T kernarg_at(size_t byte_off, int idx) {
    typedef const __attribute__((address_space(4))) char* kptr;
    typedef const __attribute__((address_space(4))) T* tptr;
    kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + byte_off;
    return ((tptr)base)[idx];
}

// Flushing a tile: E images of rows * D floats per array, staged at their destination's 16-byte phase, leave as float4 stores: job
// j = 64 pass + lane -> image e = j / chunks-per-image, 16-byte chunk cq.  What a job needs -- LDS offset, HBM offset relative to
// (array + first episode of the chunk + tile: a SCALAR base), whole / straddling / outside -- is 32-bit arithmetic computed ONCE per
// pass and used for all three arrays (the first version redid it, in 64 bits, per array: 260 of a tile's ~650 instructions).  A chunk
// that straddles its image's first or last float goes element by element -- or, where T D = 2 mod 4 (350 x 7: the TableTennis shapes),
// as the one 8-byte half it then always is.  `wt`: write-through (sc1) stores while the launch's outputs are cache resident (a scalar
// branch around the store instruction, not a second copy of the loop).
__device__ __forceinline__ void st16_at(const bool wt, float* __restrict__ base_uniform, const unsigned off, const f32x4& v) {
    if (wt) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base_uniform) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off), "v"(v), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void st8_at(const bool wt, float* __restrict__ base_uniform, const unsigned off, const f32x2& v) {
    if (wt) asm volatile("global_store_dwordx2 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base_uniform) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void st4_at(const bool wt, float* __restrict__ base_uniform, const unsigned off, const float v) {
    if (wt) asm volatile("global_store_dword %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base_uniform) : "memory");
    else asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v), "s"(base_uniform) : "memory");
}
// bp / bv / ba: (array + ((first episode of the chunk) T + first step of the tile) D) - 4 floats, wave-uniform (the 16 bytes in front keep
// the offset of a chunk that straddles its image's first float non-negative).
// A pass moves the images of EPP consecutive episodes: lane -> (episode of the pass h, 16-byte chunk cq) is fixed for the kernel
// (EPP = 2 where an image has at most 32 chunks: D <= 7), so a pass costs a handful of vector instructions beside its three LDS reads
// and three stores -- episode offsets are scalar.
template <int MASK = 7>      // 1 pos | 2 vel | 4 actions
__device__ __forceinline__ void flush_tile(const float* __restrict__ sP, const int arr_floats, float* __restrict__ bp, float* __restrict__ bv,
                                           float* __restrict__ ba, const int ne, const int pitch, const int nch, const int n, const int b0,
                                           const int td, const int td3, const bool vec, const bool wt, const int lane) {
    const int epp = nch <= 32 ? 2 : 1;
    const int h = epp == 2 ? lane >> 5 : 0, cq0 = epp == 2 ? lane & 31 : lane;
    for (int e0 = 0; e0 < ne; e0 += epp) {
        const int e = e0 + h;
        const int sh = vec ? (int)((((unsigned)(b0 + e) & 3u) * (unsigned)td3) & 3u) : 0;
        const int hi = sh + n;
        for (int cq = cq0; cq < nch; cq += 64) {        // (one trip unless an image has more than 64 chunks: D = 16)
            const int c0 = 4 * cq;
            if (!(e < ne && c0 + 4 > sh && c0 < hi)) continue;
            const float* ip = sP + e * pitch + c0;
            f32x4 vp = {0.f, 0.f, 0.f, 0.f}, vv = vp, va = vp;
            if (MASK & 1) vp = *reinterpret_cast<const f32x4*>(ip);
            if (MASK & 2) vv = *reinterpret_cast<const f32x4*>(ip + arr_floats);
            if (MASK & 4) va = *reinterpret_cast<const f32x4*>(ip + 2 * arr_floats);
            const unsigned go = (unsigned)((e * td + c0 - sh + 4) * 4);
            if (vec && c0 >= sh && c0 + 4 <= hi) {
                if (MASK & 1) st16_at(wt, bp, go, vp);
                if (MASK & 2) st16_at(wt, bv, go, vv);
                if (MASK & 4) st16_at(wt, ba, go, va);
            } else if (vec && td3 == 2) {
                // segment starts and lengths are even: a straddling chunk is exactly its upper half (the image's start) or its lower half (its end)
                const bool head = c0 < sh;
                const unsigned o = go + (head ? 8u : 0u);
                if (MASK & 1) st8_at(wt, bp, o, f32x2{head ? vp[2] : vp[0], head ? vp[3] : vp[1]});
                if (MASK & 2) st8_at(wt, bv, o, f32x2{head ? vv[2] : vv[0], head ? vv[3] : vv[1]});
                if (MASK & 4) st8_at(wt, ba, o, f32x2{head ? va[2] : va[0], head ? va[3] : va[1]});
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (c0 + k >= sh && c0 + k < hi) {
                        if (MASK & 1) st4_at(wt, bp, go + 4u * k, vp[k]);
                        if (MASK & 2) st4_at(wt, bv, go + 4u * k, vv[k]);
                        if (MASK & 4) st4_at(wt, ba, go + 4u * k, va[k]);
                    }
                }
            }
        }
    }
}

#ifndef MPK_PF_WAVES
#define MPK_PF_WAVES 2            // waves per SIMD the register allocation aims at (A/B builds: -DMPK_PF_WAVES=3)
#endif
#ifndef MPK_PF_TL_THREADS
#define MPK_PF_TL_THREADS 512     // largest workgroup of the variants with the row table in LDS (the waves of a workgroup share one copy)
#endif
// MP promp / prodmp; KQ: 4 KQ contraction columns; TL (prodmp): the row table in the workgroup's LDS instead of L2; DC: the DoF count at
// compile time (0: c.D); CT: 0 .. 2 = MPK_CTRL_* against a frozen state (mpk_trajectory_actions), 3 + MPK_CTRL_* = closed loop on the
// double integrator (mpk_trajectory_rollout / mpk_replan_step / mpk_episode_return)
// PIPE (round 6, closed loop of a SMALL launch): the chunk belongs to a WORKGROUP of 1 + kPipeProducers waves instead of one wave -- below
// ~8 000 episodes a launch takes what one wave takes for its tiles (a lone wave pays 5 - 9 cycles per instruction, dependent or not),
// and only the chain B links the tiles.  Wave 0 (consumer) runs the prologue, B for every tile and the chunk's end; producer p runs A
// (and C) of tiles p, p + NP, ...: images in a ring of 2 NP tile slots, hand-over through monotonic LDS counters (mpk_dev.h: flag_load /
// flag_store / flag_wait, no workgroup barrier inside the tile loop): `prod[p]` = tiles producer p has contracted, `chained` = tiles the
// consumer has finished.  A producer publishes a tile, flushes its (pos, vel) and -- one of its own tiles later, when the consumer has
// long passed it -- the tile's actions; it waits for the consumer only to reuse a slot.  The consumer reads the next tile's counter
// before it starts a chain and uses the value after it.  Same device code per phase (the lambdas below): same bits as the one-wave form.
#ifndef MPK_PF_PIPE_NP
#define MPK_PF_PIPE_NP 3         // producers per workgroup: with the consumer one wave per SIMD (A/B: 2 was 3 - 25 % slower, profiles/r06_phase_pipe.md)
#endif
constexpr int kPipeProducers = MPK_PF_PIPE_NP;
#ifndef MPK_PF_PIPE_WAVES
#define MPK_PF_PIPE_WAVES 1       // workgroups of the pipeline form per SIMD the register allocation aims at (A/B builds)
#endif
template <int MP, int KQ, bool TL, int DC, int CT, bool PIPE = false>
__global__ void __launch_bounds__(PIPE ? 64 * (1 + kPipeProducers) : (TL ? MPK_PF_TL_THREADS : 256), PIPE ? MPK_PF_PIPE_WAVES : MPK_PF_WAVES) k_phase_fused(const FusedArgs a, const FusedLim lim) {
    static_assert(MP != MPK_MP_DMP, "dmp with a learned phase keeps its separate launches (no reference configuration has one)");
    static_assert(!TL || MP == MPK_MP_PRODMP, "only prodmp has a row table");
    static_assert(!PIPE || (CT >= 3 && !TL), "the pipeline serves the closed loop of small launches (row table from L2)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = KQ * 4, TT = 16;
    constexpr bool CLOSED = CT >= 3;
    constexpr int CTRL = CLOSED ? CT - 3 : CT;
    const DevCfg& c = a.c;
    (void)lim;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int D = DC > 0 ? DC : c.D, T = c.T, E = a.chunk, P = c.P, pitch = a.pitch;
    double* sCen = reinterpret_cast<double*>(smem);     // promp: [c_pad / 2] RBF centres | bandwidths (| recurrence constants)
    float* sWgs = smem;                                 // prodmp: [KS] column scales | the goal scale
    float* sBT = smem + a.c_pad;                        // [t_pad] base times
    float* sTab = sBT + a.t_pad;                        // TL: [rows][2 KS + 4] row table
    float* sX = sTab + a.tab_pad + (size_t)(PIPE ? 0 : wave) * a.wave_floats;     // [E][D][KS] columns of the chunk (PIPE: of the workgroup)
    float* sPh = sX + E * a.x_pad;                      // [E][16] tau, delay, init_time, 1 / tau | promp: 1 / tau refined (float64) | prodmp: 4 boundary factors (float64)
    float* sCar = sPh + 16 * E;                         // promp: [E][2 D] position | velocity of the step before the tile
    double* sViol = reinterpret_cast<double*>(sCar + a.car_pad);     // [E][2] joint-limit excess above | below (gate)
    float* sP = reinterpret_cast<float*>(sViol + 2 * E);             // [E][pitch] desired positions of the tile
    // (| velocities | actions: E * pitch floats each; PIPE: 2 NP such slots, then the counters and the producers' promp carries)
    constexpr int NP = kPipeProducers, NB = 2 * NP;
    const int img = E * pitch, slot_floats = 3 * img;
    int* const sSync = reinterpret_cast<int*>(sP + NB * slot_floats);      // PIPE: prod[NP] .. | [8] chained | [9] tiles to run | [10 + p] flushed
    float* const sCarP = reinterpret_cast<float*>(sSync + 16);             // PIPE, promp: [NP][car_pad]
    const bool lead = !PIPE || wave == 0;                                  // the wave that owns the chunk's prologue, chain and end

    // ---- tables of the workgroup
    {
        const int tid = (int)threadIdx.x, bd = (int)blockDim.x;
        if (PIPE && tid < 16) sSync[tid] = 0;
        if (MP != MPK_MP_PRODMP) {
            for (int t = tid; t < T; t += bd) sBT[t] = c.base_times[t];
            for (int k = tid; k < 2 * c.n_total + 3; k += bd) sCen[k] = c.tab[k];
        } else {
            const double* S = c.tab + 4 * (size_t)c.n_pc + 2 * (size_t)c.n_pc * (c.nb + 1);
            const double sk = tid <= c.nb ? S[tid] : 0.0;
            const double sg = tid == KS ? S[c.nb] : 0.0;
            if (TL) {
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
            for (int t = tid; t < T; t += bd) sBT[t] = c.base_times[t];
            // sWgs[k], k < KS: the scale of column k -- 0 where the column has no parameter (disabled block, padding) --, sWgs[KS]: the goal
            // scale itself, also when the goal is disabled (relative goals)          (k_traj_phase's table)
            if (tid < KS) {
                const bool off = tid < c.nb ? c.disable_weights != 0 : (tid == c.nb ? c.disable_goal != 0 : true);
                sWgs[tid] = off ? 0.0f : (float)sk;
            }
            if (tid == KS) sWgs[KS] = (float)sg;
        }
    }
    __syncthreads();
    const float* const rows = TL ? sTab : c.rows32;
    constexpr int kRow = 2 * KS + 4;                    // [(Psi_k, dPsi_k) pairs .. | y1 y2 dy1 dy2 (float64)]
    const int row_max = TL ? a.tab_pad / kRow - 1 : c.n_pc - 1;
    const ExactDiv dsdt = make_exact_div(c.scaled_dt);
    ExpRegs ec;
    if (MP == MPK_MP_PROMP) ec.load();

    const int le = (int)(((unsigned)lane * a.inv_d) >> 16), ld = lane - le * D;      // lane <-> (episode, DoF) of the chunk
    const bool vec = a.vec_ok != 0;
    const int nch = pitch >> 2;
    const int NRT = (T + TT - 1) / TT;
    const bool store = a.pos != nullptr;
    // controller constants of the lane's DoF (four vector loads from the kernel-argument segment, waited for once: mpk_traj_quad.h)
    const int ldc = ld < D ? ld : 0;
    double pgd = kernarg_at<double>(kFusedLimOffset + offsetof(FusedLim, pg), ldc);
    double dgd = kernarg_at<double>(kFusedLimOffset + offsetof(FusedLim, dg), ldc);
    const double lod = __builtin_canonicalize(kernarg_at<double>(kFusedLimOffset + offsetof(FusedLim, lo), ldc));
    const double hid = __builtin_canonicalize(kernarg_at<double>(kFusedLimOffset + offsetof(FusedLim, hi), ldc));
    float glo32 = 0.0f, ghi32 = 0.0f;
    double glo = 0.0, ghi = 0.0;
    if (CLOSED && a.gate) {
        glo32 = kernarg_at<float>(kFusedLimOffset + offsetof(FusedLim, glo32), ldc);
        ghi32 = kernarg_at<float>(kFusedLimOffset + offsetof(FusedLim, ghi32), ldc);
        glo = kernarg_at<double>(kFusedLimOffset + offsetof(FusedLim, glo), ldc);
        ghi = kernarg_at<double>(kFusedLimOffset + offsetof(FusedLim, ghi), ldc);
    }
    asm volatile("" : "+v"(pgd), "+v"(dgd), "+v"(glo32), "+v"(ghi32), "+v"(glo), "+v"(ghi));

    const int nchunks = (a.B + E - 1) / E;
    const int cstride = (int)gridDim.x * wpb;
    // the work unit of a wave trip: a chunk -- or, with the frozen plant state (no recurrence links the tiles), split_tiles of its tiles:
    // below ~8 000 episodes a launch takes what ONE wave takes for its tiles, so small launches spread a chunk's tiles over waves
    const int nsplit = CLOSED ? 1 : a.nsplit;
    const int nunits = nchunks * nsplit;
    for (int unit = PIPE ? (int)blockIdx.x : (int)blockIdx.x * wpb + wave; unit < nunits; unit += PIPE ? nunits : cstride) {
        int ch = unit, rt_begin = 0, rt_end = NRT;
        if (!CLOSED && nsplit > 1) {
            ch = unit / nsplit;
            rt_begin = (unit - ch * nsplit) * a.split_tiles;
            rt_end = min(NRT, rt_begin + a.split_tiles);
        }
        const int b0 = ch * E, ne = min(E, a.B - b0);
        const bool on = lead && lane < ne * D;           // (PIPE: the consumer's lanes)
        const int b = b0 + (on ? le : 0);
        __builtin_amdgcn_wave_barrier();                // the chunk before has read its images and columns
        // ---- chunk prologue: columns of every (episode, DoF), per-episode constants, inputs of the recurrences
        float taul = c.tau, delayl = c.delay;
        const float itl = a.init_time_shared;
        if (on) {
            const float* prl = a.params + (size_t)b * P;
            // np.clip(action, low, high): only tau / delay carry finite bounds (black_box_wrapper.py:104-105)
            if (c.learn_tau) taul = fminf(fmaxf(prl[0], c.tau_lo), c.tau_hi);
            if (c.learn_delay) delayl = fminf(fmaxf(prl[c.learn_tau ? 1 : 0], c.delay_lo), c.delay_hi);
            const float yb = a.init_pos[(size_t)b * D + ld];
            const float* loc = prl + c.off + ld * c.Kloc;
            float* xf = sX + le * a.x_pad + ld * KS;
            if (MP == MPK_MP_PRODMP) {
                // wg = scale * [w; g] in fp32 as the reference forms it, and the two boundary residuals of
                //   pos = xi1 (y_b - Psi_b.wg) + xi2 (tau ydot_b - dPsi_b.wg) + Psi.wg        (k_traj_phase's chunk block, operation for operation)
                const float ydb = a.init_vel[(size_t)b * D + ld];
                const int nw = c.disable_weights ? 0 : c.nb, nbk = c.nb;
                float raw[KS - 2], rawg = 0.0f;
#pragma unroll
                for (int k = 0; k < KS - 2; ++k) raw[k] = k < nw ? loc[k] : 0.0f;
                if (!c.disable_goal) rawg = loc[nw];
                const float sbl = fmaxf(div_exact(itl - delayl, make_exact_div(taul)), 0.0f);
                const float* rb = rows + (size_t)min((int)rintf(div_exact(sbl, dsdt)), row_max) * kRow;
                double pb = 0.0, vb = 0.0;
                float wgg = c.disable_goal ? 0.0f : rawg * sWgs[KS];
                if (c.relative_goal) wgg = c.relgoal_before_scale ? (rawg + yb) * sWgs[KS] : wgg + yb;
                if (c.goal_off_on) wgg = wgg + c.goal_offset;
#pragma unroll
                for (int k = 0; k < KS - 2; ++k) {
                    float wg = raw[k] * sWgs[k];
                    wg = k == nbk ? wgg : wg;
                    pb += (double)rb[2 * k] * (double)wg;
                    vb += (double)rb[2 * k + 1] * (double)wg;
                    xf[k] = wg;
                }
                xf[KS - 2] = (float)((double)yb - pb);
                xf[KS - 1] = (float)((double)(taul * ydb) - vb);
                if (ld == 0) {
                    float* sc4 = sPh + 16 * le;
                    sc4[0] = taul; sc4[1] = delayl; sc4[2] = itl; sc4[3] = 1.0f / taul;
                    const double* yb4 = reinterpret_cast<const double*>(rb + 2 * KS - 4);
                    const double y1b = yb4[0], y2b = yb4[1], dy1b = yb4[2], dy2b = yb4[3];
                    const double idet = div_pos(1.0, y1b * dy2b - y2b * dy1b);
                    double* bc4 = reinterpret_cast<double*>(sc4 + 8);
                    bc4[0] = dy2b * idet; bc4[1] = dy1b * idet; bc4[2] = y1b * idet; bc4[3] = y2b * idet;
                }
            } else {
                // raw parameter columns [w_0 .. w_{nb-1}, init_pos (zero-padded family), 0 ..]
#pragma unroll
                for (int k = 0; k < KS; ++k) xf[k] = k < c.nb ? loc[k] : (k < c.KT ? yb : 0.0f);
                if (ld == 0) {
                    float* sc4 = sPh + 16 * le;
                    sc4[0] = taul; sc4[1] = delayl; sc4[2] = itl; sc4[3] = 0.0f;
                    *reinterpret_cast<double*>(sc4 + 4) = make_pos_div((double)taul).y;
                }
            }
        }
        double qs = 0.0, qds = 0.0;
        int nst = T;
        ReplanVals rv{T, 0, 0, false};
        bool t_bad = false;                 // gate: raw tau / delay outside their bounds
        double tpen = 0.0;
        if (on) {
            const size_t ix = (size_t)b * D + ld;
            qs = a.q[ix]; qds = a.qd[ix];
            if (CLOSED) {
                if (a.rp.traj_steps) {
                    rv = replan_eval(a.rp, b, T);
                    if (!a.gate && ld == 0) replan_write(a.rp, b, rv);
                    nst = rv.seg;
                } else if (a.n_steps) {
                    nst = min(a.n_steps[b], T);
                }
                if (a.gate && a.check_td) {
                    const double tau = (double)a.raw_params[(size_t)b * P], delay = (double)a.raw_params[(size_t)b * P + 1];
                    t_bad = !(tau >= a.tau_b[0] && tau <= a.tau_b[1] && delay >= a.delay_b[0] && delay <= a.delay_b[1]);
                    tpen = 3.0 * (fmax(0.0, tau - a.tau_b[1]) + fmax(0.0, a.tau_b[0] - tau)) +
                           3.0 * (fmax(0.0, delay - a.delay_b[1]) + fmax(0.0, a.delay_b[0] - delay));
                }
            }
        }
        asm volatile("" : "+v"(qs), "+v"(qds), "+v"(nst));      // waited for once, here (mpk_traj_quad.h)
        if (CLOSED && a.gate && lane < 2 * E) sViol[lane] = 0.0;
        const int tcond = (CLOSED && a.rp.cond_pos) ? min(max(nst - 1, 0), T - 1) : -1;
        const int she = (vec && on) ? (int)((((unsigned)b & 3u) * (unsigned)a.td3) & 3u) : 0;     // 16-byte phase of the episode's outputs
        const int oq = le * pitch + she + ld;           // (step 0 of the tile, this column) in the chunk's images
        float row0p = 0.0f, row0v = 0.0f;               // gate: the desired state of step 0 (condition of an episode that executes nothing)
        bool p_bad = false;
        double over = 0.0, under = 0.0;
        __builtin_amdgcn_wave_barrier();
        // promp: the carry in front of a tile that starts at step t_begin: p[t_begin] of every (episode, DoF) -- one partial round, lane <->
        // episode (the velocity carry is read by a tile that starts at the last step only: such horizons are neither split nor piped)
        auto carry_at = [&](const int t_begin, float* const car) {
            if (lane < ne) {
                const float* sc4 = sPh + 16 * lane;
                const PosDiv taud{(double)sc4[0], *reinterpret_cast<const double*>(sc4 + 4)};
                float h[KS];
#pragma unroll
                for (int k = 0; k < KS; ++k) h[k] = 0.0f;
                const double x = phase_f64(c, sBT[t_begin] + sc4[2], taud, sc4[1], ec);
                rbf_row<KS>(c, sCen, sCen + c.n_total, x, (double)c.ws, h, ec);
                for (int d = 0; d < D; ++d) {
                    const float* xc = sX + lane * a.x_pad + d * KS;
                    float p = 0.0f;
#pragma unroll
                    for (int k = 0; k < KS; ++k) p = fmaf(h[k], xc[k], p);
                    car[lane * 2 * D + d] = p;
                    car[lane * 2 * D + D + d] = 0.0f;
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        if (MP == MPK_MP_PROMP && !PIPE) carry_at(rt_begin * TT, sCar);
        // (without stores, tiles behind every episode's last executed step and behind the condition step carry nothing anybody reads --
        // unless the gate has to see the whole plan)
        int nrt_live = rt_end;
        if (CLOSED && !store && !a.gate) {
            int nmax = on ? max(nst, tcond + 1) : 0;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) nmax = max(nmax, __shfl_xor(nmax, m));
            nrt_live = min(NRT, (nmax + TT - 1) / TT);
        }
        // ---- the three phases of a tile (device code shared by the one-wave form and the pipeline's roles)
        // A: rows and contractions of the tile's (episode, step) items into the images at iP (| velocities); carA: promp's carry
        auto do_A = [&](const int rt, float* const iP, float* const carA) {
            const int t0 = rt * TT;
            float* const iV = iP + img;
            (void)carA;
            for (int i0 = 0; i0 < ne * TT; i0 += 64) {
                const int idx = i0 + lane, ei = idx >> 4, j = idx & (TT - 1);
                // (lanes past the chunk's last episode compute its last episode's item again: the same values into the same slots)
                const int e = ei < ne ? ei : ne - 1;
                const float* sc4 = sPh + 16 * e;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(sc4);
                const float delay = sc[1], it = sc[2];
                const int shi = vec ? (int)((((unsigned)(b0 + e) & 3u) * (unsigned)a.td3) & 3u) : 0;
                float* const o0 = iP + e * pitch + shi + j * D;
                float* const o1 = iV + e * pitch + shi + j * D;
                const float* const sXe = sX + e * a.x_pad;
                if (MP == MPK_MP_PRODMP) {
                    const int t = min(t0 + j, T - 1);
                    const float inv_tau = sc[3];
                    const ExactDiv dtau{sc[0], inv_tau, (__float_as_uint(sc[0]) & 0x7fffffu) == 0x7fffffu};
                    const double* bc4 = reinterpret_cast<const double*>(sc4 + 8);
                    const double bca = bc4[0], bcb = bc4[1], bcc = bc4[2], bcd = bc4[3];
                    float hq[2 * KS];
                    const float time = sBT[t] + it;
                    const float s = fmaxf(div_exact(time - delay, dtau), 0.0f);
                    if (s > (float)c.len_factor) atomicOr(a.flag, 1);
                    const int ri = min((int)rintf(div_exact(s, dsdt)), row_max);
                    const float4* row = reinterpret_cast<const float4*>(rows + (size_t)ri * kRow);
#pragma unroll
                    for (int jj = 0; jj < (2 * KS - 4) / 4; ++jj) {
                        const float4 q4 = row[jj];
                        hq[4 * jj] = q4.x; hq[4 * jj + 1] = q4.y; hq[4 * jj + 2] = q4.z; hq[4 * jj + 3] = q4.w;
                    }
                    const double* y4 = reinterpret_cast<const double*>(row + (2 * KS - 4) / 4);
                    const double y1 = y4[0], y2 = y4[1], dy1 = y4[2], dy2 = y4[3];
                    hq[2 * KS - 4] = (float)fma(bca, y1, -(bcb * y2));
                    hq[2 * KS - 3] = (float)fma(bca, dy1, -(bcb * dy2));
                    hq[2 * KS - 2] = (float)fma(bcc, y2, -(bcd * y1));
                    hq[2 * KS - 1] = (float)fma(bcc, dy2, -(bcd * dy1));
                    if constexpr (DC > 0) {
                        dofs_unrolled<DC, KQ>(sXe, hq, inv_tau, o0, o1);
                    } else {
                        for (int d = 0; d < D; ++d) {
                            float x[KS];
#pragma unroll
                            for (int jj = 0; jj < KQ; ++jj) {
                                const float4 v = *reinterpret_cast<const float4*>(sXe + d * KS + 4 * jj);
                                x[4 * jj + 0] = v.x; x[4 * jj + 1] = v.y; x[4 * jj + 2] = v.z; x[4 * jj + 3] = v.w;
                            }
                            f32x2 pv = {0.0f, 0.0f};
#pragma unroll
                            for (int k = 0; k < KS; ++k)
                                pv = __builtin_elementwise_fma(f32x2{hq[2 * k], hq[2 * k + 1]}, f32x2{x[k], x[k]}, pv);
                            o0[d] = pv[0];
                            o1[d] = pv[1] * inv_tau;
                        }
                    }
                } else {
                    // promp: the lane evaluates the row of step t + 1; (pos, vel)[t] = (the step before's position, the difference towards this one)
                    const int t = t0 + j;
                    const int ts = min(t + 1, T - 1);
                    const PosDiv taud{(double)sc[0], *reinterpret_cast<const double*>(sc4 + 4)};
                    float h[KS];
#pragma unroll
                    for (int k = 0; k < KS; ++k) h[k] = 0.0f;
                    const double x = phase_f64(c, sBT[ts] + it, taud, delay, ec);
                    rbf_row<KS>(c, sCen, sCen + c.n_total, x, (double)c.ws, h, ec);
                    const int tc = min(t, T - 1);
                    const int th = tc < T - 1 ? tc + 1 : T - 1, tl = tc < T - 1 ? tc : T - 2;
                    const float rdt = 1.0f / ((sBT[th] + it) - (sBT[tl] + it));
                    const bool last_row = t >= T - 1;     // repeats the difference before it
                    float* const car = carA + e * 2 * D;
                    auto dof = [&](const int d) {
                        float xx[KS];
#pragma unroll
                        for (int jj = 0; jj < KQ; ++jj) {
                            const float4 v = *reinterpret_cast<const float4*>(sXe + d * KS + 4 * jj);
                            xx[4 * jj + 0] = v.x; xx[4 * jj + 1] = v.y; xx[4 * jj + 2] = v.z; xx[4 * jj + 3] = v.w;
                        }
                        float p = 0.0f;
#pragma unroll
                        for (int k = 0; k < KS; ++k) p = fmaf(h[k], xx[k], p);
                        const float cp = car[d], cv = car[D + d];
                        const float pb_ = lane_below(p);
                        const float prev = j > 0 ? pb_ : cp;
                        float v = (p - prev) * rdt;
                        const float vb_ = lane_below(v);
                        if (last_row) v = j > 0 ? vb_ : cv;
                        o0[d] = prev;
                        o1[d] = v;
                        if (j == TT - 1 && ei < ne) { car[d] = p; car[D + d] = v; }
                    };
                    if constexpr (DC > 0) {
#pragma unroll
                        for (int d = 0; d < DC; ++d) dof(d);
                    } else {
                        for (int d = 0; d < D; ++d) dof(d);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        // B: 16 steps of every (episode, DoF) recurrence of the chunk on the images at iP (| velocities | actions)
        auto do_B = [&](const int rt, float* const iP) {
            const int t0 = rt * TT, nrows = min(TT, T - t0);
            float* const iV = iP + img;
            float* const iA = iV + img;
            (void)iA;
            if (on) {
                const float* pP = iP + oq;
                const float* pV = iV + oq;
                if (CLOSED && tcond >= t0 && tcond < t0 + TT) {      // condition_on_desired: the desired state at the last executed step
                    const size_t si = (size_t)b * D + ld;
                    a.rp.cond_pos[si] = pP[(tcond - t0) * D];
                    a.rp.cond_vel[si] = pV[(tcond - t0) * D];
                }
                if (CLOSED && a.gate) {
                    if (rt == 0) { row0p = pP[0]; row0v = pV[0]; }
                }
            }
            {
                // the chain; with the validity gate it also tests the positions it pulls into registers against the joint limits (GATE hook of
                // pd_tile_steps: no LDS read or wait of its own -- a separate pass of sixteen reads cost 12 % at 8 192 episodes)
                const bool full_tile = nrows == TT && tile_fully_executed(on, nst, t0);
                int tb = 0;
                double gsum[2] = {over, under};
                if (on && store) {
                    if (CLOSED && a.gate) {
                        if (full_tile)
                            pd_tile_steps<CTRL, false, CLOSED, 0, 0, true, true>(iP + oq, iV + oq, iA + oq, D, t0, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds,
                                                                               nullptr, nullptr, 16, glo32, ghi32, &tb, glo, ghi, gsum);
                        else
                            pd_tile_steps<CTRL, true, CLOSED, 0, 0, true, true>(iP + oq, iV + oq, iA + oq, D, t0, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds,
                                                                              nullptr, nullptr, nrows, glo32, ghi32, &tb, glo, ghi, gsum);
                    } else if (full_tile) {
                        pd_tile_steps<CTRL, false, CLOSED>(iP + oq, iV + oq, iA + oq, D, t0, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds);
                    } else {
                        pd_tile_steps<CTRL, true, CLOSED>(iP + oq, iV + oq, iA + oq, D, t0, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds,
                                                          nullptr, nullptr, nrows);
                    }
                } else if (on) {                        // the verbose < 2 step: no action image either
                    if (CLOSED && a.gate) {
                        if (full_tile)
                            pd_tile_steps<CTRL, false, CLOSED, 0, 0, false, true>(iP + oq, iV + oq, nullptr, D, t0, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds,
                                                                                nullptr, nullptr, 16, glo32, ghi32, &tb, glo, ghi, gsum);
                        else
                            pd_tile_steps<CTRL, true, CLOSED, 0, 0, false, true>(iP + oq, iV + oq, nullptr, D, t0, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds,
                                                                               nullptr, nullptr, nrows, glo32, ghi32, &tb, glo, ghi, gsum);
                    } else if (full_tile) {
                        pd_tile_steps<CTRL, false, CLOSED, 0, 0, false>(iP + oq, iV + oq, nullptr, D, t0, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds);
                    } else {
                        pd_tile_steps<CTRL, true, CLOSED, 0, 0, false>(iP + oq, iV + oq, nullptr, D, t0, nst, pgd, dgd, lod, hid, a.plant_dt, qs, qds,
                                                                       nullptr, nullptr, nrows);
                    }
                }
                if (tb) p_bad = true;
                over = gsum[0]; under = gsum[1];
            }
        };
        // C: the tile's runs of nrows * D floats per episode and array (MASK: 1 pos | 2 vel | 4 actions)
        auto do_C = [&](const int rt, const float* const iP, auto mask_tag) {
            constexpr int MASK = decltype(mask_tag)::value;
            const int t0 = rt * TT, nrows = min(TT, T - t0);
            const size_t toff = ((size_t)b0 * T + t0) * D;
            flush_tile<MASK>(iP, img, a.pos + toff - 4, a.vel + toff - 4, a.actions + toff - 4, ne, pitch, nch, nrows * D, b0, T * D, a.td3, vec,
                             a.wt != 0, lane);
        };
        using std::integral_constant;
        if constexpr (!PIPE) {
            for (int rt = rt_begin; rt < nrt_live; ++rt) {
                do_A(rt, sP, sCar);
                do_B(rt, sP);
                __builtin_amdgcn_wave_barrier();
                if (store) do_C(rt, sP, integral_constant<int, 7>());
                __builtin_amdgcn_wave_barrier();            // the tile's LDS reads are issued before the next tile's writes
            }
        } else {
            // the consumer tells the producers how many tiles the chunk runs (what it knows from the integer state), then the roles part
            if (wave == 0 && lane == 0) sSync[9] = nrt_live;
            __syncthreads();
            const int ntile = sSync[9];
            if (wave == 0) {
                int have = flag_load(sSync + 0);
                for (int rt = 0; rt < ntile; ++rt) {
                    const int p = rt % NP, k = rt / NP;
                    if (have < k + 1 && !flag_wait(sSync + p, k + 1, a.fault, 128)) break;
                    if (rt + 1 < ntile) have = flag_load(sSync + (rt + 1) % NP);   // the next tile's counter: read now, used after the chain
                    do_B(rt, sP + (rt % NB) * slot_floats);
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) flag_store(sSync + 8, rt + 1);
                }
            } else {
                const int p = wave - 1;
                float* const carA = sCarP + p * a.car_pad;
                int prev = -1, k = 0;
                bool alive = true;
                for (int rt = p; rt < ntile && alive; rt += NP, ++k) {
                    float* const iP = sP + (rt % NB) * slot_floats;
                    if (MP == MPK_MP_PROMP) carry_at(rt * TT, carA);
                    do_A(rt, iP, carA);
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) flag_store(sSync + p, k + 1);
                    if (store) do_C(rt, iP, integral_constant<int, 3>());
                    // the tile before this one of the producer's own: the consumer is (long) past it -- its actions leave, its slot is free
                    if (prev >= 0) {
                        alive = flag_wait(sSync + 8, prev + 1, a.fault, 128);
                        if (alive && store) do_C(prev, sP + (prev % NB) * slot_floats, integral_constant<int, 4>());
                    }
                    __builtin_amdgcn_wave_barrier();
                    prev = rt;
                }
                if (prev >= 0 && alive && store && flag_wait(sSync + 8, prev + 1, a.fault, 128))
                    do_C(prev, sP + (prev % NB) * slot_floats, integral_constant<int, 4>());
                if (store && a.gate) {
                    // (the consumer rewrites the action rows of an invalid plan: after this wave's stores have been acknowledged)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) flag_store(sSync + 10 + p, 1);
                }
            }
        }
        // ---- end of the chunk: plant state, integer state, gate verdict (and the roll-back of an invalid plan)
        if (CLOSED && lead) {
            bool invalid = false;
            if (a.gate) {
                const unsigned long long m = __ballot(on && p_bad);
                const unsigned long long mine = (m >> (le * D)) & ((1ull << D) - 1ull);
                invalid = on && (mine != 0ull || t_bad);
                if (m != 0ull) {                        // (wave-uniform) excess of the episode = its columns left to right
                    double so = 0.0, su = 0.0;
                    for (int d = 0; d < D; ++d) {
                        so += __shfl(over, le * D + d);
                        su += __shfl(under, le * D + d);
                    }
                    over = so; under = su;
                }
                if (on && ld == 0) {
                    a.valid[b] = invalid ? 0 : 1;
                    const double n = (double)(T * D);
                    // table_tennis_env.py:282-289: -(3 tau excess + 3 delay excess + mean(max(pos - high, 0)) + mean(max(low - pos, 0)))
                    if (a.penalty) a.penalty[b] = -(tpen + over / n + under / n);
                }
            }
            if (on) {
                const size_t si = (size_t)b * D + ld;
                if (!invalid) { a.q[si] = qs; a.qd[si] = qds; }
                int seg = nst;
                if (a.gate) {
                    if (invalid) {
                        seg = 0;
                        if (a.rp.cond_pos) { a.rp.cond_pos[si] = row0p; a.rp.cond_vel[si] = row0v; }
                    }
                    if (a.rp.traj_steps && ld == 0) replan_write(a.rp, b, rv, !invalid);
                }
                if (ld == 0) {
                    if (a.seg_out) a.seg_out[b] = seg;
                    if (a.ret) a.ret[b] = 0.0;
                }
            }
            if (a.gate && store) {
                // an invalid plan executes nothing: its action rows, written speculatively, become zeros
                const unsigned long long mi = __ballot(invalid && ld == 0);
                if (mi != 0ull) {
                    if (PIPE) {                         // the producers' speculative action rows have landed
                        for (int pp = 0; pp < NP; ++pp)
                            if (!flag_wait(sSync + 10 + pp, 1, a.fault, 128)) break;
                    }
                    for (int e = 0; e < ne; ++e) {
                        if (!((mi >> (e * D)) & 1ull)) continue;
                        float* ap = a.actions + (size_t)(b0 + e) * T * D;
                        for (int i = lane; i < T * D; i += 64) {
                            if (a.wt) store4<true>(ap + i, 0.0f); else store4<false>(ap + i, 0.0f);
                        }
                    }
                }
            }
        }
    }
}

#ifndef MPK_DEVICE_ONLY
// fp32 thresholds of a float64 interval: pos (fp32) lies in [low, high] exactly when it lies in [up(low), down(high)]
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

bool phase_fused_capable(const DevCfg& c) {
    if (c.mp_type != MPK_MP_PROMP && c.mp_type != MPK_MP_PRODMP) return false;
    const int need = c.mp_type == MPK_MP_PRODMP ? c.nb + 3 : c.KT;
    if (need > 8 || c.D > kMaxD || c.D < 1) return false;
    if (c.mp_type == MPK_MP_PRODMP && (!c.rows32 || c.rows32_stride != 20)) return false;
    if (c.mp_type == MPK_MP_PROMP && c.T < 2) return false;
    return true;
}

int launch_phase_fused(const DevCfg& c, const float* params, const float* init_pos, const float* init_vel, float init_time_shared,
                       float* pos, float* vel, float* actions, const RolloutDev& rc, double* q, double* qd, const int32_t* n_steps,
                       const ReplanDev* rp, const GateDev* gate, double* ret, int32_t* seg_out, int32_t* range_flag, int B, int num_cu,
                       void* stream, const char** kernel_name, const Tuning& tune, int* fault) {
    if (!phase_fused_capable(c)) return MPK_ENOTIMPL;
    const bool prodmp = c.mp_type == MPK_MP_PRODMP;
    const int need = prodmp ? c.nb + 3 : c.KT;
    const int KQ = (!prodmp && need <= 4) ? 1 : 2, KS = 4 * KQ;
    const bool closed = rc.plant_type == MPK_PLANT_DOUBLE_INTEGRATOR;
    FusedArgs fa{};
    fa.c = c;
    fa.params = params; fa.init_pos = init_pos; fa.init_vel = init_vel; fa.init_time_shared = init_time_shared;
    fa.pos = pos; fa.vel = vel; fa.actions = actions;
    fa.q = q; fa.qd = qd; fa.n_steps = n_steps;
    if (rp) fa.rp = *rp;
    fa.plant_dt = rc.dt;
    fa.flag = range_flag;
    fa.ret = ret; fa.seg_out = seg_out;
    fa.B = B;
    fa.fault = fault;
    FusedLim fl{};
    for (int d = 0; d < c.D; ++d) { fl.pg[d] = rc.pg[d]; fl.dg[d] = rc.dg[d]; fl.lo[d] = rc.lo[d]; fl.hi[d] = rc.hi[d]; }
    if (gate) {
        if (!closed) { set_error("the validity gate needs the double-integrator plant"); return MPK_EINVAL; }
        fa.gate = 1;
        fa.raw_params = gate->raw_params ? gate->raw_params : params;
        fa.valid = gate->valid; fa.penalty = gate->penalty;
        fa.check_td = gate->check_td;
        fa.tau_b[0] = gate->tau_b[0]; fa.tau_b[1] = gate->tau_b[1];
        fa.delay_b[0] = gate->delay_b[0]; fa.delay_b[1] = gate->delay_b[1];
        for (int d = 0; d < c.D; ++d) {
            fl.glo[d] = gate->lo[d]; fl.ghi[d] = gate->hi[d];
            fl.glo32[d] = f32_at_least(gate->lo[d]); fl.ghi32[d] = f32_at_most(gate->hi[d]);
        }
    }
    // chunks of E episodes: one lane per (episode, DoF) in the recurrence, 16 E items per tile; "phase_chunk" overrides.  Measured
    // (profiles/r06_phase_fused_chunks.md, us at 8 192 / 65 536 episodes, chunks of 4 against 8): TableTennis-ProDMP actions 58.8 / 488
    // against 82.9 / 536, closed loop 74.7 / 588 against 92.9 / 554, verbose < 2 step 50.3 / 367 against 50.9 / 248; BeerPong-ProMP
    // actions 62.9 / 490 against 88.3 / 494, closed loop 78.0 / 636 against 101 / 526 -- the recurrence costs a wave the same whatever
    // its lanes carry, so the closed loop wants eight once a launch has more chunks of eight than the chip holds waves; the frozen-state
    // actions (no dependent chain) and every launch below that prefer the larger number of waves
    const int e_max = 64 / c.D > 8 ? 8 : 64 / c.D;
    int E = e_max >= 8 ? 8 : (e_max >= 4 ? 4 : e_max);
    {
        const long simds = (long)num_cu * 4;
        if (E == 8 && (!closed || ((long)B + 7) / 8 < 4 * simds)) E = 4;
        // small launches: smaller chunks until every SIMD has a wave.  Below ~8 000 episodes a launch takes what ONE wave takes for its 22
        // tiles whatever its lanes carry (us at 1 024 / 2 048 / 4 096 episodes of TableTennis-ProDMP, chunks of 2: closed loop 54.4 / 54.4 /
        // 64.5, chunks of 4: 61.0 / 61.8 / 62.8, chunks of 8: 88 / 88 / 91 -- the second round of items and the extra flush passes of a
        // fuller chunk are serial time on that wave), so the chunk is the smallest that still fills the chip's SIMDs once
        // (the frozen-state actions spread a chunk's TILES over waves instead -- below --, and keep chunks of four)
        while (closed && E > 2 && ((long)B + E - 1) / E < simds && (E / 2) * c.D >= 8) E >>= 1;
    }
    // closed loop of a small launch: the chunk on a workgroup of 1 + kPipeProducers waves (PIPE: the kernel's comment), ONE round of workgroups -- two
    // are resident on a CU (189 registers) --; not for promp horizons of 16 n + 1 steps
    // (a tile that starts at the last step reads the velocity carry); "phase_pipe" 1 / 0 forces / forbids
    const bool pipe_ok = closed && (prodmp || c.T % 16 != 1);
    const int e_top = e_max >= 8 ? 8 : (e_max >= 4 ? 4 : e_max);
    constexpr long kPipeWgsPerCu = MPK_PF_PIPE_WAVES >= 3 ? 3 : 2;     // resident workgroups per CU (registers: the kernel's launch bounds)
    bool pipe = pipe_ok && ((long)B + e_top - 1) / e_top <= kPipeWgsPerCu * (long)num_cu;
    if (tune.phase_pipe >= 0) pipe = tune.phase_pipe == 1 && pipe_ok;
    if (pipe) {
        // chunks of four where two workgroups per CU hold the launch (one round of items per tile, two flush passes), else eight.
        // TableTennis-ProDMP closed loop / verbose < 2, us (profiles/r06_phase_pipe.md): 1 024 episodes 32.0 / 28.1 (one-wave form 52.8 /
        // 39.0), 2 048: 34.9 / 30.1 (52.5 / 38.8), 4 096: 42.7 / 31.4 (61.8 / 38.6); BeerPong-ProMP 1 024: 32.0 / 26.6 (59.2 / 47.7)
        E = e_max >= 4 ? 4 : e_max;
        if (((long)B + E - 1) / E > kPipeWgsPerCu * (long)num_cu && e_top > E) E = e_top;
    }
    if (tune.phase_chunk >= 1 && tune.phase_chunk <= e_max) E = tune.phase_chunk;
    fa.chunk = E;
    fa.x_pad = c.D * KS;
    const bool out = pos != nullptr;
    fa.vec_ok = out && ((reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(vel) | reinterpret_cast<uintptr_t>(actions)) & 15u) == 0 ? 1 : 0;
    fa.td3 = (c.T * c.D) & 3;
    // (staging two or four tiles per flush -- 896- / 1 792-byte store runs -- was measured and lost: profiles/r06_phase_fused_large.md; carrying the
    // option cost the tile loop 3 - 9 %)
    fa.pitch = (16 * c.D + 3 + 3) / 4 * 4;              // 16 steps + up to three floats of shift, whole 16-byte chunks
    fa.t_pad = (c.T + 3) / 4 * 4;
    fa.c_pad = prodmp ? KS + 4 : (4 * c.n_total + 6 + 3) / 4 * 4;
    fa.car_pad = prodmp ? 0 : (E * 2 * c.D + 3) / 4 * 4;
    fa.inv_d = 65536u / (unsigned)c.D + 1u;
    fa.inv_ch = 65536u / (unsigned)(fa.pitch / 4) + 1u;
    fa.wave_floats = E * fa.x_pad + 16 * E + fa.car_pad + 4 * E + 3 * E * fa.pitch;
    if (pipe) fa.wave_floats += (2 * kPipeProducers - 1) * 3 * E * fa.pitch + 16 + kPipeProducers * fa.car_pad;
    const size_t wave_bytes = (size_t)fa.wave_floats * sizeof(float);
    size_t shared_bytes = (size_t)(fa.t_pad + fa.c_pad) * sizeof(float);
    fa.wt = out && (double)B * c.T * c.D * 12.0 <= kWtBytes ? 1 : 0;
    if (tune.write_through >= 0) fa.wt = tune.write_through != 0 ? 1 : 0;
    if (wave_bytes + shared_bytes > kLdsPerCu) return MPK_ENOTIMPL;
    const long chunks = ((long)B + E - 1) / E;
    // prodmp: the row table in LDS when that still leaves eight waves on a CU and the launch fills them ("phase_table" 0: from L2)
    bool lds_table = false;
    int wpb = (int)((kLdsDefault - shared_bytes) / wave_bytes);
    wpb = wpb > 4 ? 4 : (wpb < 1 ? 1 : wpb);
    if (prodmp) {
        int rows_needed = c.n_pc;
        const float tau_lo = c.learn_tau ? c.tau_lo : c.tau, delay_lo = c.learn_delay ? c.delay_lo : c.delay;
        if ((double)c.t_last > 0.0 && tau_lo > 0.f) {
            const double s_max = ((double)c.t_last + (double)init_time_shared - (double)delay_lo) / (double)tau_lo;
            const double r = s_max / (double)c.scaled_dt + 4.0;
            if (r < (double)c.n_pc) rows_needed = r < 4.0 ? 4 : (int)r;
        }
        const size_t tab_bytes = (size_t)rows_needed * (2 * KS + 4) * sizeof(float);
        lds_table = tab_bytes + shared_bytes + 8 * wave_bytes <= kLdsPerCu && chunks >= (long)num_cu * 8;
        if (tune.phase_table == 0 || pipe) lds_table = false;
        if (tune.phase_table == 1 && !pipe && tab_bytes + shared_bytes + wave_bytes <= kLdsPerCu) lds_table = true;
        if (lds_table) {
            fa.tab_pad = rows_needed * (2 * KS + 4);
            shared_bytes += tab_bytes;
            wpb = (int)((kLdsPerCu - shared_bytes) / wave_bytes);
            wpb = wpb > MPK_PF_TL_THREADS / 64 ? MPK_PF_TL_THREADS / 64 : wpb;
        }
    }
    if (tune.tiles_wpb > 0 && wpb > tune.tiles_wpb) wpb = tune.tiles_wpb;
    // frozen-state actions of a small launch: the tiles of a chunk on several waves (each repeats the chunk's prologue) until the chip's
    // SIMDs hold four waves each, eight units per chunk at most; "phase_split" overrides (1 = whole chunks).  TableTennis-ProDMP / BeerPong-
    // ProMP, us, whole chunks -> split (profiles/r06_phase_fused_chunks.md): 1 024 episodes 40.9 -> 12.4 / 49.1 -> 14.6, 2 048: 42.6 -> 19.2 /
    // 49.2 -> 20.3, 4 096: 52.3 -> 30.8 / 54.2 -> 29.2, 8 192: 59.6 -> 61.7 (row table in LDS: stays whole) / 62.7 -> 50.4.
    // promp: a tile that starts AT the last step reads the velocity carry of the tile before it, so such horizons (T = 16 n + 1) stay whole
    fa.nsplit = 1;
    const int nrt = (c.T + 15) / 16;
    fa.split_tiles = nrt;
    if (!closed && (prodmp || c.T % 16 != 1)) {
        const long simds = (long)num_cu * 4;
        long want = chunks < 4 * simds ? (4 * simds + chunks - 1) / chunks : 1;
        want = want > 8 ? 8 : want;
        if (lds_table && want <= 2) want = 1;
        if (tune.phase_split >= 1) want = tune.phase_split;
        want = want > nrt ? nrt : want;
        fa.split_tiles = (int)((nrt + want - 1) / want);
        fa.nsplit = (nrt + fa.split_tiles - 1) / fa.split_tiles;
    }
    const long units = chunks * fa.nsplit;
    if (units < (long)num_cu * wpb) {                  // fewer chunks than one workgroup per CU would take: smaller workgroups
        const int w = (int)((units + num_cu - 1) / num_cu);
        wpb = w < 1 ? 1 : (w < wpb ? w : wpb);
    }
    if (pipe) wpb = 1;
    const size_t lds = wave_bytes * wpb + shared_bytes;
    int per_cu = (int)(kLdsPerCu / lds);
    per_cu = per_cu > 32 / wpb ? 32 / wpb : (per_cu < 1 ? 1 : per_cu);
    if (tune.phase_waves > 0 && per_cu * wpb > tune.phase_waves) per_cu = tune.phase_waves / wpb > 1 ? tune.phase_waves / wpb : 1;
    long blocks = (units + wpb - 1) / wpb;
    if (blocks > (long)num_cu * per_cu) blocks = (long)num_cu * per_cu;
    if (pipe) blocks = chunks;                          // one workgroup per chunk, no loop
    auto go = [&](auto kern) -> int {
        if (lds > kLdsDefault) {
            hipError_t e = allow_full_lds(kern);
            if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(pipe ? 64 * (1 + kPipeProducers) : 64 * wpb), lds, (hipStream_t)stream, fa, fl);
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    };
    const int ct = (closed ? 3 : 0) + rc.controller_type;
    const bool dc7 = c.D == 7 && tune.pd_generic != 1;
    auto by_ct = [&](auto mp_tag, auto kq_tag, auto tl_tag, auto dc_tag) -> int {
        constexpr int MP = decltype(mp_tag)::value, KQ_ = decltype(kq_tag)::value, DC = decltype(dc_tag)::value;
        constexpr bool TL = decltype(tl_tag)::value;
        switch (ct) {
            case 0: return go(k_phase_fused<MP, KQ_, TL, DC, 0>);
            case 1: return go(k_phase_fused<MP, KQ_, TL, DC, 1>);
            case 2: return go(k_phase_fused<MP, KQ_, TL, DC, 2>);
            case 3: if constexpr (!TL) { if (pipe) return go(k_phase_fused<MP, KQ_, false, DC, 3, true>); } return go(k_phase_fused<MP, KQ_, TL, DC, 3>);
            case 4: if constexpr (!TL) { if (pipe) return go(k_phase_fused<MP, KQ_, false, DC, 4, true>); } return go(k_phase_fused<MP, KQ_, TL, DC, 4>);
            default: if constexpr (!TL) { if (pipe) return go(k_phase_fused<MP, KQ_, false, DC, 5, true>); } return go(k_phase_fused<MP, KQ_, TL, DC, 5>);
        }
    };
    using std::integral_constant;
    using std::bool_constant;
    typedef integral_constant<int, MPK_MP_PRODMP> PD;
    typedef integral_constant<int, MPK_MP_PROMP> PM;
    typedef integral_constant<int, 1> I1;
    typedef integral_constant<int, 2> I2;
    typedef integral_constant<int, 0> D0;
    typedef integral_constant<int, 7> D7;
    if (prodmp) {
        *kernel_name = closed ? (out ? (lds_table ? "k_phase_fused<prodmp,lds,closed>" : (pipe ? "k_phase_fused<prodmp,pipe,closed>" : "k_phase_fused<prodmp,closed>"))
                                     : (lds_table ? "k_phase_fused<prodmp,lds,closed,lean>" : (pipe ? "k_phase_fused<prodmp,pipe,closed,lean>" : "k_phase_fused<prodmp,closed,lean>")))
                              : (lds_table ? "k_phase_fused<prodmp,lds,act>" : "k_phase_fused<prodmp,act>");
        if (lds_table) return dc7 ? by_ct(PD(), I2(), bool_constant<true>(), D7()) : by_ct(PD(), I2(), bool_constant<true>(), D0());
        return dc7 ? by_ct(PD(), I2(), bool_constant<false>(), D7()) : by_ct(PD(), I2(), bool_constant<false>(), D0());
    }
    *kernel_name = closed ? (out ? (pipe ? "k_phase_fused<promp,pipe,closed>" : "k_phase_fused<promp,closed>")
                                 : (pipe ? "k_phase_fused<promp,pipe,closed,lean>" : "k_phase_fused<promp,closed,lean>")) : "k_phase_fused<promp,act>";
    if (KQ == 1) return dc7 ? by_ct(PM(), I1(), bool_constant<false>(), D7()) : by_ct(PM(), I1(), bool_constant<false>(), D0());
    return dc7 ? by_ct(PM(), I2(), bool_constant<false>(), D7()) : by_ct(PM(), I2(), bool_constant<false>(), D0());
}
#endif  // MPK_DEVICE_ONLY

}  // namespace mpk
