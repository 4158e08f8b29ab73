// k_pd_rollout / k_pd_rollout_tiles / k_reacher_rollout: tracking controller + plant loop (+ SimpleReacher reward) on existing trajectories
#include "mpk_tile.h"
#include "mpk_reward.h"

namespace mpk {

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
    int* fault;              // the handle's fault word (host memory): a chain / helper wave that gives up waiting says so (wave_gave_up)
};

// NG = groups per wave: with NG = 4 a wave owns four consecutive groups and lane quarter j runs group j's recurrence,
// so four recurrences advance in parallel (the same idea as k_traj_quad); NG = 1 keeps more waves for small batches.
// RW: additionally SimpleReacherEnv's per-step reward (simple_reacher.py:56-72).  The serial lanes leave the plant
// position and the clipped action of every step of the tile in LDS as float64 (the position image reuses the desired
// pos | vel staging, which the recurrence has already pulled into registers); then all 64 lanes turn (episode, step)
// items into rewards in parallel -- cumulative joint angles, sin / cos, end effector, control cost, each summed left to
// right as numpy does.  Only the recurrence itself stays serial.
// CT (round 5): the reward kernels carry their controller as a template parameter (the plant is the double integrator, the launcher
// picks the instantiation): with the 6 (controller, plant) x 2 (full / partial tile) chains of the run-time switch, the reward pass
// and its sin / cos path in ONE kernel the compiler spilled 140 scalar registers into vector lanes inside the tile loop.
// DC (round 5): the DoF count compiled in (0: run time) for the shapes the reference registers -- 2 / 5 links (Simple / LongSimpleReacher,
// envs/__init__.py:38-59), 7 joints (BASELINE cfg2 / cfg4 / cfg5) --: group geometry, staging offsets and the reward pass's reads become
// immediates, as in k_traj_ring / k_traj_flat (round 4).
// Reward kernels with helper waves (HW; round 5 with a workgroup barrier per tile, round 6 with flags): the pass that turns a tile's
// (episode, step) items into rewards needs none of the chain's registers, and at a few thousand episodes the chain wave has its SIMD to
// itself -- every instruction of the pass costs its full issue latency there (820 of a tile's 4 070 cycles, profiles/r05_rollout.md).
// The workgroup gets two more waves: waves 0 - 3 run the chains as before and leave the clipped actions of a tile as a float64 image
// (two buffers, used in turn); helper wave h turns the images of chain waves 2 h and 2 h + 1 into rewards (control cost only) and
// stores them, while the chains run the next tile.  A tile that can hold an item past steps_before_reward (one in thirteen at the
// reference's setting) keeps the whole pass -- end effector and all -- on its chain wave, which alone has the plant positions.  What a
// helper needs per episode (executed steps, step offset, episode) sits in an LDS table the chain wave fills per unit (two tables, by the
// parity of the unit).
// Hand-over WITHOUT a workgroup barrier (round 5's s_barrier per tile tied four latency-bound chain waves to the slowest of them and
// lost at every size: 8 192 episodes 30.9 -> 47.0 us): per chain wave two monotonic counters in LDS, `pub` = tiles whose images are
// complete (written by the chain wave after the tile -- a wave's DS operations retire in order, so whoever reads the counter sees the
// images) and `done` = tiles the helper has consumed (written after its image reads have returned).  A helper polls the `pub` of its two
// chain waves (s_sleep between empty polls); a chain wave reads `done` at the START of a tile and uses the value after the staging -- it
// may overwrite buffer n & 1 once done >= n - 1 -- so it waits only when its helper is a whole tile behind.  No cycle: the helper waits
// for nothing but `pub`.  Every spin is bounded; a wave that gives up raises the handle's fault word (as k_traj_ring's roles do).
constexpr int kRwSlots = 8, kRwSlotInts = 4;                 // per chain wave and parity: eight episode slots of (executed steps, step offset, episode, -)
template <int NG, bool RW, int CT = -1, int DC = 0, bool HW = false>
__global__ void __launch_bounds__(HW ? 384 : 256) k_pd_rollout_tiles(const PdArgs a) {
    static_assert(RW || !HW, "helper waves serve the reward pass");
    static_assert(RW || CT < 0, "only the reward kernels carry the controller as a template parameter");
    constexpr int kShC = DC <= 1 ? 0 : (DC <= 2 ? 1 : (DC <= 4 ? 2 : (DC <= 8 ? 3 : 4)));
    const int sh = DC > 0 ? kShC : a.sh;
    constexpr int SLOT = 3 * kStageStride + (RW ? (HW ? 4 : 2) * kStageStride : 0);   // floats per group slot
    extern __shared__ __attribute__((aligned(16))) float smem[];           // [4 chain waves][NG][SLOT] | [2][4][kRwSlots][kRwSlotInts] slot tables | pub[4] done[4]
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool helper = HW && wave_all >= 4;
    const int wave = helper ? 0 : wave_all;      // (a helper computes with the lane geometry of a chain wave; its own index is wave_all - 4)
    float* sSt = smem + wave * (NG * SLOT);      // per group: desired pos | desired vel | actions (| u as float64, HW: two buffers)
    int* const sTabAll = reinterpret_cast<int*>(smem + 4 * NG * SLOT);
    int* const sPub = sTabAll + 2 * 4 * kRwSlots * kRwSlotInts;            // HW: [4] tiles published per chain wave | [4] tiles consumed
    int* const sDone = sPub + 4;
    const int D = DC > 0 ? DC : a.D, T = a.T, B = a.B, SEG = 16 * D, DP = 1 << sh, NTW = 16 >> sh;
    const int col = lane & 15, bl = col >> sh, d = col & (DP - 1);
    const int jq = lane >> 4;                                // the group (of this wave's NG) whose recurrence the lane runs
    const bool lane_serial = jq < NG && d < D;
    const int seg4 = SEG >> 2;
    const int sseg = DC > 0 ? lane / (4 * DC) : (int)(((unsigned)lane * a.inv_seg4) >> 16);
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
    const int ustride = gridDim.x * 4;
    unsigned tcount = 0;                                             // tiles this wave has passed: parity = float64 action buffer
    if (HW) {
        if (threadIdx.x < 8) sPub[threadIdx.x] = 0;
        __syncthreads();                                             // (once per launch)
    }
    if (helper) {
        // ---------------- helper wave h: the control-cost pass of chain waves 2 h, 2 h + 1, behind them ----------------
        const int h = wave_all - 4;
        const int tl = lane & 15;
        // serve(cw, ..): every tile chain wave cw has published and this wave has not consumed yet; true if there was one
        auto serve = [&](const int cw, int& cons, int& rt, int& it, const int total) -> bool {
            if (cons >= total) return false;
            const int pub = flag_load(sPub + cw);
            bool any = false;
            while (cons < pub) {
                any = true;
                const int* tw = sTabAll + (it & 1) * (4 * kRwSlots * kRwSlotInts) + cw * (kRwSlots * kRwSlotInts);
                const int rows = min(16, T - rt * 16), t = rt * 16 + tl;
                // the chain wave keeps a tile that can hold an item past steps_before_reward: the same conservative rule there
                bool mine = false;
                if (lane < kRwSlots) mine = tw[lane * kRwSlotInts + 2] >= 0 && tw[lane * kRwSlotInts + 1] + rt * 16 + 15 >= a.steps_before_reward;
                if (!(MPK_RW_ALWAYS_TRIG || __any(mine) != 0)) {
                    const float* sW = smem + cw * (NG * SLOT);
                    const int npass = (NG * NTW + 3) >> 2;
#pragma unroll 1
                    for (int p = 0; p < npass; ++p) {
                        const int sl = 4 * p + (lane >> 4);
                        const int* e4 = tw + (sl < kRwSlots ? sl : 0) * kRwSlotInts;
                        const int pns = e4[0], pb = e4[2];
                        const int e = sl & (NTW - 1), j = sl >> (4 - sh);
                        const bool item = sl < NG * NTW && pb >= 0 && tl < rows;
                        const double* uv = reinterpret_cast<const double*>(sW + (j < NG ? j : 0) * SLOT + (3 + 2 * (cons & 1)) * kStageStride) +
                                           e * DP * kRwCol + tl;
                        double r = DC > 0 ? reacher_ctrl_item<DC>(uv, D) : reacher_ctrl_item_d(uv, D);
                        r = t < pns ? r : 0.0;
                        if (item) a.rewards[(size_t)pb * T + t] = r;
                    }
                }
                ++cons;
                if (++rt == NRT) { rt = 0; ++it; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image (and table) reads have returned: the buffer is the chain's again
                if (lane == 0) flag_store(sDone + cw, cons);
            }
            return any;
        };
        auto tiles_of = [&](const int cw) {
            const int first = vb * 4 + cw;
            return first < units ? ((units - first + ustride - 1) / ustride) * NRT : 0;
        };
        const int tot0 = tiles_of(2 * h), tot1 = tiles_of(2 * h + 1);
        int cons0 = 0, rt0 = 0, it0 = 0, cons1 = 0, rt1 = 0, it1 = 0;
        unsigned spins = 0;
        while (cons0 < tot0 || cons1 < tot1) {
            const bool a0 = serve(2 * h, cons0, rt0, it0, tot0);
            const bool a1 = serve(2 * h + 1, cons1, rt1, it1, tot1);
            if (a0 || a1) { spins = 0; continue; }
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kFlagSpinLimit) { wave_gave_up(a.fault, 64); return; }
        }
        return;
    }
    // the helper's `done` counter of this chain wave must reach `want` (rarely waits: the helper's pass is a tenth of a tile)
    auto wait_helper = [&](const int want) {
        unsigned spins = 0;
        while (flag_load(sDone + wave) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kFlagSpinLimit) { wave_gave_up(a.fault, 64); break; }
        }
    };
    for (int it_ = 0, un = vb * 4 + wave; un < units; ++it_, un += ustride) {
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
        // Addresses of the tile pieces: ONE wave-uniform base per array (the unit's first episode) + a 32-bit per-lane byte offset per
        // group that advances by a tile per tile -- the (scalar base, vector offset) form of the loads and stores.  With 64-bit
        // per-lane pointers recomputed per tile the staging of a tile was ~120 instructions, and a wave that has its SIMD to itself
        // (4 096 episodes: one wave per SIMD) pays 5 - 6 cycles for every one of them: 900 of a tile's 2 900 cycles (round 4, second
        // session: trace + disassembly, profiles/r04_vmcnt_in_tile_loops.md)
        const size_t ubase = (size_t)g0 * NTW * T * D;                 // first element of the unit in every [B, T, D] array
        const float* const bpos = a.des_pos + ubase;
        const float* const bvel = a.des_vel + ubase;
        float* const bact = a.actions ? a.actions + ubase : nullptr;
        unsigned goff[NG];                                             // byte offset of this lane's float4 of tile 0, group j
        // the desired (pos, vel) pieces travel TWO tiles ahead of the recurrence, in two register sets used in turn: one tile's
        // chain (16 steps, ~1300 cycles) is shorter than a load's round trip even from the memory-side cache, so with one tile of
        // lookahead every tile waited for its inputs (B = 2048: 182 cycles per step against the chain's 83; profiles/r04_rollout.md)
        f32x4 lpA[NG], lvA[NG], lpB[NG], lvB[NG], lpC[NG], lvC[NG];
        auto fetch = [&](const int rt, f32x4 (&lp)[NG], f32x4 (&lv)[NG]) {
            const bool okt = w4 < min(16, T - rt * 16) * D;
            const unsigned tb = (unsigned)(rt * SEG) * 4u;
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                if (mover[j] && okt) {
                    lp[j] = ld_off(reinterpret_cast<const f32x4*>(bpos), goff[j] + tb);
                    lv[j] = ld_off(reinterpret_cast<const f32x4*>(bvel), goff[j] + tb);
                }
            }
        };
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int b0 = (g0 + j) * NTW;
            mover[j] = g0 + j < a.G && sseg < NTW && b0 + sseg < B;
            goff[j] = (unsigned)(((size_t)j * NTW * T * D + gofs) * sizeof(float));
            lpA[j] = f32x4{0, 0, 0, 0}; lvA[j] = lpA[j]; lpB[j] = lpA[j]; lvB[j] = lpA[j]; lpC[j] = lpA[j]; lvC[j] = lpA[j];
        }
        // RW: the reward pass turns (episode, step) items into rewards, item = pass * 64 + lane -> episode slot je = 4 pass + lane / 16
        // of the unit, step lane % 16 of the tile: a lane meets the SAME episodes in every tile, so what the pass needs per episode
        // (executed steps, step offset, goal) is read once per unit into registers.  Round 4: read per item and tile, these were three
        // dependent trips to the memory-side cache inside every pass, with two waves per SIMD to hide them.
        // (the launcher keeps NG * NTW <= 8 episodes per unit: two passes at most)
        constexpr int kRwPasses = 2;
        const int rw_npass = RW ? (NG * NTW + 3) >> 2 : 0;
        int rw_b[kRwPasses], rw_ns[kRwPasses], rw_s0[kRwPasses], rw_q[kRwPasses];
        double rw_gx[kRwPasses], rw_gy[kRwPasses];
        if (RW) {
#pragma unroll
            for (int p = 0; p < kRwPasses; ++p) {
                const int je = 4 * p + (lane >> 4), e = je & (NTW - 1), j = je >> (4 - sh);
                const int b = (g0 + j) * NTW + e;
                const bool ok = p < rw_npass && j < NG && g0 + j < a.G && b < B;
                rw_b[p] = ok ? b : -1;
                // byte offset of (group slot, the episode's first column, this lane's step) in the [column][step] float64 images
                // (an empty slot reads its lane's step of column 0 of group 0 -- in bounds, discarded)
                rw_q[p] = ok ? (j * SLOT) * 4 + (e * DP * kRwCol + (lane & 15)) * 8 : (lane & 15) * 8;
                rw_ns[p] = ok ? (a.n_steps ? min(a.n_steps[b], T) : T) : 0;
                rw_s0[p] = ok && a.step0 ? a.step0[b] : 0;
                rw_gx[p] = ok ? a.goal[2 * (size_t)b] : 0.0;
                rw_gy[p] = ok ? a.goal[2 * (size_t)b + 1] : 0.0;
            }
        }
        if (HW) wait_helper((int)tcount - 1);   // (the table of this parity served the unit before the last: all of its tiles are consumed)
        if (HW && lane < kRwSlots) {
            // the helpers' per-episode inputs of this unit: slot = (group, episode) as the passes count them
            const int e = lane & (NTW - 1), j = lane >> (4 - sh);
            const int b = (g0 + j) * NTW + e;
            const bool ok = lane < NG * NTW && j < NG && g0 + j < a.G && b < B;
            int* e4 = sTabAll + (it_ & 1) * (4 * kRwSlots * kRwSlotInts) + (wave * kRwSlots + lane) * kRwSlotInts;
            e4[0] = ok ? (a.n_steps ? min(a.n_steps[b], T) : T) : 0;
            e4[1] = ok && a.step0 ? a.step0[b] : 0;
            e4[2] = ok ? b : -1;
        }
        bool prev_own = false;                   // the previous tile's rewards sit in rw_r (a tile with the distance term): this wave's to store
        MPK_STAMP(1);
        fetch(0, lpA, lvA);
        constexpr bool kTwoAhead = !RW || MPK_RW_LOOK == 2;
        // tiles of input lookahead: one with the reward (a tile takes three times as long there; 32 registers less), two with four
        // groups per wave (registers), THREE with one or two groups per wave -- one wave per SIMD or two: nothing else hides a load
        // that takes longer than two tiles (MPK_PD_LOOK: A/B build knob)
        constexpr int kAhead = !kTwoAhead ? 1 : (NG <= 2 && MPK_PD_LOOK == 3 ? 3 : 2);
        if (kAhead >= 2 && NRT > 1) fetch(1, lpB, lvB);
        if (kAhead >= 3 && NRT > 2) fetch(2, lpC, lvC);
        // the unit's serial inputs are waited for HERE (they are older than the fetches above: the wait leaves those in flight).  Left to
        // the compiler the wait sits in front of their first use in every tile's chain as `s_waitcnt vmcnt(0)` -- every tile then began
        // by waiting for the loads it had just issued for two tiles ahead and for the previous tile's stores (round 4, second session)
        asm volatile("" : "+v"(qs), "+v"(qds), "+v"(nst));
        // The actions of tile rt leave at the START of tile rt + 1, after its staging: the wait for a tile's prefetched inputs is
        // `s_waitcnt vmcnt(0)` (the compiler cannot count across the loop), and with the stores issued at a tile's end it was a wait
        // for their acknowledgement too -- 900 - 1 350 cycles per tile against 1 900 of chain (trace, round 4 second session).  Issued
        // here they have the whole chain to retire.  (The action image is read into registers before the chain rewrites it: a wave's
        // LDS operations execute in order.)
        auto store_actions = [&](const int rt) {
            if (!a.actions) return;
            const bool okt = w4 < min(16, T - rt * 16) * D;
            const unsigned tb = (unsigned)(rt * SEG) * 4u;
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                if (mover[j] && okt) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(sSt + j * SLOT + 2 * kStageStride + rofs);
                    const unsigned off = goff[j] + tb;
                    // (write-through in the 64-bit address form of store16: with the scalar base handed to an inline asm the
                    // reward instantiation -- 56 spilled scalar registers -- stored to wild addresses)
                    float* const dst = reinterpret_cast<float*>(reinterpret_cast<char*>(bact) + off);
                    if (a.wt) store16<true>(dst, v);      // cache-resident actions: write-through (wave-uniform)
                    else store16<false>(dst, v);
                }
            }
        };
        // RW: likewise the rewards of tile rt (one float64 per pass and lane, kept in registers) leave at the start of tile rt + 1 --
        // stored at the end of the pass, their acknowledgement was what the next tile's staging waited for (`vmcnt(0)`): with the
        // sin / cos chains gone from 199 of 200 steps (round 5) that wait was most of what the reward still cost
        double rw_r[kRwPasses] = {0.0, 0.0};
        auto store_rewards = [&](const int rt) {
            if (!RW) return;
            const int tl = lane & 15, t = rt * 16 + tl;
            if (tl < min(16, T - rt * 16)) {
#pragma unroll
                for (int p = 0; p < kRwPasses; ++p)
                    if (p < rw_npass && rw_b[p] >= 0) a.rewards[(size_t)rw_b[p] * T + t] = rw_r[p];
            }
        };
        // Round 6: the control-cost pass of a tile WITHOUT the distance term runs a tile LATE, inside the next tile's staging -- its LDS
        // reads are issued at the top of the staging (in front of the chain that rewrites the float64 action image: a wave's LDS
        // operations execute in order), its sum and store at the staging's end, so the reads' latency passes under the staging's own
        // instructions.  Right after the chain the pass was 330 cycles of a tile's 3 350 at 4 096 episodes (one wave per SIMD, trace:
        // profiles/r06_rollout_reward.md) for ~30 instructions: LDS latency + a dependent sum with nothing beside them.
        constexpr bool kLate = RW && !HW && MPK_RW_LATE != 0;
        constexpr int kLateD = DC > 0 ? DC : 1;
        auto late_reads = [&](double (&lu)[kRwPasses][kLateD]) {
            if constexpr (DC > 0) {
#pragma unroll
                for (int p = 0; p < kRwPasses; ++p) {
                    const double* uv = reinterpret_cast<const double*>(reinterpret_cast<const char*>(sSt) + rw_q[p] + 3 * kStageStride * 4);
#pragma unroll
                    for (int dd = 0; dd < DC; ++dd) lu[p][dd] = uv[dd * kRwCol];      // (an empty slot reads in bounds: rw_q)
                }
            }
        };
        auto late_finish = [&](const int rt, const double (&lu)[kRwPasses][kLateD]) {
            const int tl = lane & 15, t = rt * 16 + tl;
            const bool row = tl < min(16, T - rt * 16);
#pragma unroll
            for (int p = 0; p < kRwPasses; ++p) {
                if (p >= rw_npass) break;
                double r;
                if constexpr (DC > 0) {
                    double ctrl = 0.0;
#pragma unroll
                    for (int dd = 0; dd < DC; ++dd) ctrl = dd == 0 ? lu[p][dd] * lu[p][dd] : ctrl + lu[p][dd] * lu[p][dd];
                    r = 0.0 - ctrl;                     // reacher_ctrl_item, operation for operation
                } else {
                    r = reacher_ctrl_item_d(reinterpret_cast<const double*>(reinterpret_cast<const char*>(sSt) + rw_q[p] + 3 * kStageStride * 4), D);
                }
                r = (rw_b[p] >= 0 && t < rw_ns[p]) ? r : 0.0;
                if (row && rw_b[p] >= 0) a.rewards[(size_t)rw_b[p] * T + t] = r;
            }
        };
        auto tile = [&](const int rt, f32x4 (&lp)[NG], f32x4 (&lv)[NG]) {
            const int rows = min(16, T - rt * 16);
            int consumed = 0;
            if (HW) consumed = flag_load(sDone + wave);     // read here, used after the staging: its latency is hidden
            const bool late = kLate && rt > 0 && !prev_own;   // (wave-uniform)
            double lu[kRwPasses][kLateD];
            if (late) late_reads(lu);
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                if (mover[j] && w4 < rows * D) {
                    *reinterpret_cast<f32x4*>(sSt + j * SLOT + rofs) = lp[j];
                    *reinterpret_cast<f32x4*>(sSt + j * SLOT + kStageStride + rofs) = lv[j];
                }
            }
            if (rt + kAhead < NRT) fetch(rt + kAhead, lp, lv);   // into the set this tile has just emptied
            if (rt > 0) { store_actions(rt - 1); if (kLate || HW ? prev_own : true) store_rewards(rt - 1); }
            if (late) late_finish(rt - 1, lu);
            __builtin_amdgcn_wave_barrier();
            if (rt < 16) MPK_STAMP(10 + 3 * rt);
            // RW: can any (episode, step) item of this tile carry the distance term?  (wave-uniform, conservative: the tile's last step
            // against every episode slot's step offset; the pass decides exactly, per item)
            bool tile_dist = RW;
            if (RW && !MPK_RW_ALWAYS_TRIG) {
                bool mine = false;
#pragma unroll
                for (int p = 0; p < kRwPasses; ++p) mine = mine || (rw_b[p] >= 0 && rw_s0[p] + rt * 16 + 15 >= a.steps_before_reward);
                tile_dist = __any(mine) != 0;
            }
            if (HW && consumed < (int)tcount - 1) wait_helper((int)tcount - 1);   // float64 action buffer tcount & 1 is free again
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
                    double* q64 = reinterpret_cast<double*>(sg) + col * kRwCol;           // [column][step] images
                    double* u64 = reinterpret_cast<double*>(sg + (3 + (HW ? 2 * (tcount & 1) : 0)) * kStageStride) + col * kRwCol;
                    auto go = [&](auto keep_tag) {
                        constexpr int KEEP = decltype(keep_tag)::value;
                        if (full_tile)
                            pd_tile_steps<CTRL, false, INTEG, KEEP>(sg + o0, sg + kStageStride + o0, sg + 2 * kStageStride + o0, D, rt * 16,
                                                                    nst, pgd, dgd, lod, hid, dtp, qs, qds, q64, u64);
                        else
                            pd_tile_steps<CTRL, true, INTEG, KEEP>(sg + o0, sg + kStageStride + o0, sg + 2 * kStageStride + o0, D, rt * 16,
                                                                   nst, pgd, dgd, lod, hid, dtp, qs, qds, q64, u64, rows);
                    };
                    if (!RW) go(std::integral_constant<int, 0>());
                    else if (tile_dist) go(std::integral_constant<int, 1>());
                    else go(std::integral_constant<int, 2>());        // no item of the tile needs the plant positions
                };
                using std::integral_constant;
                const bool dint = a.rc.plant_type == MPK_PLANT_DOUBLE_INTEGRATOR;
                if constexpr (CT >= 0) tile_steps(integral_constant<int, CT>(), integral_constant<int, MPK_PLANT_DOUBLE_INTEGRATOR>());
                else switch (a.rc.controller_type) {
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
            if (rt < 16) MPK_STAMP(11 + 3 * rt);
            if (HW || kLate) prev_own = tile_dist;
            if (RW && (!(HW || kLate) || tile_dist)) {
                // item = pass * 64 + lane -> episode slot 4 pass + lane / 16 of the unit, step lane % 16 of the tile; the control cost of
                // every item as straight-line code, the end effector behind ONE wave-uniform branch
                const int tl = lane & 15, t = rt * 16 + tl;
#pragma unroll 1
                for (int p = 0; p < rw_npass; ++p) {                  // (one copy of the pass: the two must not interleave)
                    static_assert(kRwPasses == 2, "selects below");
                    const int pb = p ? rw_b[1] : rw_b[0], pns = p ? rw_ns[1] : rw_ns[0], ps0 = p ? rw_s0[1] : rw_s0[0];
                    const int pq = p ? rw_q[1] : rw_q[0];
                    const bool live = pb >= 0 && t < pns;               // (pns <= T: a step past the horizon is never live)
                    const bool dist_on = live && ps0 + t >= a.steps_before_reward;
                    const char* sb = reinterpret_cast<const char*>(sSt) + pq;
                    const double* qv = reinterpret_cast<const double*>(sb);
                    const double* uv = reinterpret_cast<const double*>(sb + (3 + (HW ? 2 * (tcount & 1) : 0)) * kStageStride * 4);
                    double r = DC > 0 ? reacher_ctrl_item<DC>(uv, D) : reacher_ctrl_item_d(uv, D);
                    // wave-uniform: does ANY of the pass's 64 items carry the distance term?  (MPK_RW_ALWAYS_TRIG: round 4's pass, A/B)
                    if (MPK_RW_ALWAYS_TRIG || (tile_dist && __any(dist_on) != 0)) {
                        // (one or two groups per wave = launches of a few thousand episodes, a wave alone on its SIMD: the five links' sin / cos
                        // chains side by side -- the run-time loop runs them one after the other, 3 150 of the last tile's 3 480 cycles)
                        if ((MPK_RW_DC == 5 || (MPK_RW_DC_SMALL && DC == 5 && NG <= 2)) && D == 5)
                            r = reacher_reward_item<5>(qv, uv, D, dist_on, p ? rw_gx[1] : rw_gx[0], p ? rw_gy[1] : rw_gy[0]);
                        else r = reacher_reward_item<0>(qv, uv, D, dist_on, p ? rw_gx[1] : rw_gx[0], p ? rw_gy[1] : rw_gy[0]);
                    }
                    r = live ? r : 0.0;
                    if (p) rw_r[1] = r; else rw_r[0] = r;
                }
                __builtin_amdgcn_wave_barrier();
            }
            __builtin_amdgcn_wave_barrier();
            if (rt < 16) MPK_STAMP(12 + 3 * rt);
            if (HW) {                                  // the tile's float64 action images are the helper's now
                ++tcount;
                if (lane == 0) flag_store(sPub + wave, (int)tcount);
            }
        };
        if (kAhead == 3) {
            for (int rt = 0; rt < NRT; rt += 3) {
                tile(rt, lpA, lvA);
                if (rt + 1 < NRT) tile(rt + 1, lpB, lvB);
                if (rt + 2 < NRT) tile(rt + 2, lpC, lvC);
            }
        } else if (kAhead == 1) {
#pragma unroll 1
            for (int rt = 0; rt < NRT; ++rt) tile(rt, lpA, lvA);
        } else {
            for (int rt = 0; rt < NRT; rt += 2) {
                tile(rt, lpA, lvA);
                if (rt + 1 < NRT) tile(rt + 1, lpB, lvB);
            }
        }
        store_actions(NRT - 1);
        if (kLate || HW ? prev_own : true) store_rewards(NRT - 1);
        if (kLate && !prev_own) {
            double lu[kRwPasses][kLateD];
            late_reads(lu);
            late_finish(NRT - 1, lu);
        }
        __builtin_amdgcn_wave_barrier();
        if (serial) {
            const size_t si = (size_t)bs * D + d;
            if (a.rc.plant_type != MPK_PLANT_STATIC) { a.Q[si] = qs; a.QD[si] = qds; }
        }
        MPK_STAMP(90);
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_pd_rollout_pipe (round 6): the rollout on existing trajectories of a SMALL launch as a producer / consumer workgroup -- the form
// k_phase_fused<.., pipe> gave the learned-phase step.  At a few thousand episodes a wave of k_pd_rollout_tiles has its SIMD to itself and
// pays 5 - 9 cycles for every instruction it issues: staging 950 - 1 400 cycles, chain 1 900 - 2 000, reward pass 100 - 330 per tile
// (profiles/r06_rollout_reward.md) -- and only the chain links the tiles.  Here wave 0 (consumer) runs the chains of a unit's NG
// groups, producer p = 1 .. NP stages tile p - 1, p - 1 + NP, ... (coalesced float4 loads -> its LDS slot), and, once the consumer has chained it,
// stores the tile's actions and turns its float64 images into rewards (control cost; the end effector where an item is past
// steps_before_reward) -- everything but the chain is off the critical path: the reward costs the consumer its sixteen image writes per
// tile (+ 13 %), and the launch the end-effector passes of the LAST tile, which run after the consumer has finished (~2 us of tail).
// Hand-over: monotonic LDS counters as in k_phase_fused<.., pipe> (mpk_dev.h flag_*), one tile slot per producer (while the consumer
// chains a tile, the other producers finish the tiles before it and stage the tiles after it); every spin bounded, fault word on
// timeout.  Same device functions (pd_tile_steps, reacher_*_item): same bits.  Measured: - 10 % against k_pd_rollout_tiles below 8 192
// episodes, with and without the reward (profiles/r06_rollout_reward.md, section 4).
#ifndef MPK_ROLL_PIPE_NP
#define MPK_ROLL_PIPE_NP 3       // producers per workgroup (A/B builds): with two, a producer's finish (action stores + two reward passes: ~2 300 cycles
                                 // for a lone wave) + restaging left the consumer waiting ~600 cycles on every other tile (trace: profiles/r06_rollout_reward.md)
#endif
constexpr int kRollPipeNP = MPK_ROLL_PIPE_NP;
template <int NG, bool RW, int CT, int DC>
__global__ void __launch_bounds__(64 * (1 + kRollPipeNP)) k_pd_rollout_pipe(const PdArgs a) {
    static_assert(RW || CT < 0, "only the reward kernels carry the controller as a template parameter");
    constexpr int NP = kRollPipeNP, NB = NP;        // one tile slot per producer: stage -> (consumer chains) -> finish, the other producer's tile in between
    constexpr int kShC = DC <= 1 ? 0 : (DC <= 2 ? 1 : (DC <= 4 ? 2 : (DC <= 8 ? 3 : 4)));
    const int sh = DC > 0 ? kShC : a.sh;
    constexpr int SLOT = 3 * kStageStride + (RW ? 2 * kStageStride : 0);       // floats per group: pos | vel | act (| u as float64)
    extern __shared__ __attribute__((aligned(16))) float smem[];                // [NB][NG][SLOT] | prod[NP] .. [8] chained
    int* const sSync = reinterpret_cast<int*>(smem + NB * NG * SLOT);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = DC > 0 ? DC : a.D, T = a.T, B = a.B, SEG = 16 * D, DP = 1 << sh, NTW = 16 >> sh;
    const int NRT = (T + 15) >> 4;
    const int units = (a.G + NG - 1) / NG;
    if (threadIdx.x < 16) sSync[threadIdx.x] = 0;
    __syncthreads();
    if (wave == 0) {
        // ---------------- consumer: the NG recurrences of a unit, one group per lane quarter ----------------
        const int col = lane & 15, bl = col >> sh, d = col & (DP - 1), jq = lane >> 4;
        const bool lane_serial = jq < NG && d < D;
        double pgd = 0.0, dgd = 0.0, lod = 0.0, hid = 0.0;
#pragma unroll
        for (int dd = 0; dd < kMaxD; ++dd)
            if (dd == d) { pgd = a.rc.pg[dd]; dgd = a.rc.dg[dd]; lod = a.rc.lo[dd]; hid = a.rc.hi[dd]; }
        lod = __builtin_canonicalize(lod); hid = __builtin_canonicalize(hid);
        const double dtp = a.rc.dt;
        int g = 0;                                       // tiles this workgroup has passed (all units)
        int have = 0;
        MPK_STAMP(1);
        for (int un = blockIdx.x; un < units; un += gridDim.x) {
            const int g0 = un * NG;
            const int bs = (g0 + jq) * NTW + bl;
            const bool serial = lane_serial && g0 + jq < a.G && bs < B;
            double qs = 0.0, qds = 0.0;
            int nst = T, s0 = 0;
            if (serial) {
                const size_t si = (size_t)bs * D + d;
                qs = a.Q[si]; qds = a.QD[si];
                if (a.n_steps) nst = min(a.n_steps[bs], T);
                if (RW && a.step0) s0 = a.step0[bs];
            }
            asm volatile("" : "+v"(qs), "+v"(qds), "+v"(nst), "+v"(s0));
            for (int rt = 0; rt < NRT; ++rt, ++g) {
                const int p = g % NP, k = g / NP;
                if (rt < 16) MPK_STAMP(10 + 3 * rt);
                if (have < k + 1 && !flag_wait(sSync + p, k + 1, a.fault, 256)) return;
                have = flag_load(sSync + (g + 1) % NP);        // the next tile's counter: read now, used after the chain
                if (rt < 16) MPK_STAMP(11 + 3 * rt);
                float* sg = smem + ((g % NB) * NG + (jq < NG ? jq : 0)) * SLOT;
                const int rows = min(16, T - rt * 16);
                // RW: can any (episode, step) item of this tile carry the distance term?  (the producers apply the same rule)
                bool tile_dist = RW;
                if (RW && !MPK_RW_ALWAYS_TRIG) tile_dist = __any(serial && s0 + rt * 16 + 15 >= a.steps_before_reward) != 0;
                if (serial) {
                    const int o0 = bl * SEG + d;
                    const bool full_tile = rows == 16 && __all(nst >= rt * 16 + 16) != 0;
                    auto tile_steps = [&](auto ctrl_tag, auto plant_tag) {
                        constexpr int CTRL = decltype(ctrl_tag)::value;
                        constexpr bool INTEG = decltype(plant_tag)::value == MPK_PLANT_DOUBLE_INTEGRATOR;
                        double* q64 = reinterpret_cast<double*>(sg) + col * kRwCol;
                        double* u64 = reinterpret_cast<double*>(sg + 3 * kStageStride) + col * kRwCol;
                        auto go = [&](auto keep_tag) {
                            constexpr int KEEP = decltype(keep_tag)::value;
                            if (full_tile)
                                pd_tile_steps<CTRL, false, INTEG, KEEP>(sg + o0, sg + kStageStride + o0, sg + 2 * kStageStride + o0, D, rt * 16, nst,
                                                                        pgd, dgd, lod, hid, dtp, qs, qds, q64, u64);
                            else
                                pd_tile_steps<CTRL, true, INTEG, KEEP>(sg + o0, sg + kStageStride + o0, sg + 2 * kStageStride + o0, D, rt * 16, nst,
                                                                       pgd, dgd, lod, hid, dtp, qs, qds, q64, u64, rows);
                        };
                        if (!RW) go(std::integral_constant<int, 0>());
                        else if (tile_dist) go(std::integral_constant<int, 1>());
                        else go(std::integral_constant<int, 2>());
                    };
                    using std::integral_constant;
                    const bool dint = a.rc.plant_type == MPK_PLANT_DOUBLE_INTEGRATOR;
                    if constexpr (CT >= 0) tile_steps(integral_constant<int, CT>(), integral_constant<int, MPK_PLANT_DOUBLE_INTEGRATOR>());
                    else switch (a.rc.controller_type) {
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
                if (rt < 16) MPK_STAMP(12 + 3 * rt);
                if (lane == 0) flag_store(sSync + 8, g + 1);
            }
            MPK_STAMP(90);
            if (serial && a.rc.plant_type != MPK_PLANT_STATIC) {
                const size_t si = (size_t)bs * D + d;
                a.Q[si] = qs; a.QD[si] = qds;
            }
        }
        return;
    }
    // ---------------- producers: staging, action stores and the reward pass of tiles p, p + NP, ... ----------------
    const int p = wave - 1;
    const int seg4 = SEG >> 2;
    const int sseg = DC > 0 ? lane / (4 * DC) : (int)(((unsigned)lane * a.inv_seg4) >> 16);
    const int w4 = (lane - sseg * seg4) * 4;
    const unsigned rofs = (unsigned)(sseg * SEG + w4);
    const size_t gofs = (size_t)sseg * T * D + w4;
    constexpr int kRwPasses = 2;
    int g = 0;
    for (int un = blockIdx.x; un < units; un += gridDim.x) {
        const int g0 = un * NG;
        const size_t ubase = (size_t)g0 * NTW * T * D;
        const float* const bpos = a.des_pos + ubase;
        const float* const bvel = a.des_vel + ubase;
        float* const bact = a.actions ? a.actions + ubase : nullptr;
        bool mover[NG];
        unsigned goff[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int b0 = (g0 + j) * NTW;
            mover[j] = g0 + j < a.G && sseg < NTW && b0 + sseg < B;
            goff[j] = (unsigned)(((size_t)j * NTW * T * D + gofs) * sizeof(float));
        }
        // RW: what the pass needs per episode, once per unit (k_pd_rollout_tiles: the same slots, item = pass * 64 + lane)
        const int rw_npass = RW ? (NG * NTW + 3) >> 2 : 0;
        int rw_b[kRwPasses], rw_ns[kRwPasses], rw_s0[kRwPasses], rw_q[kRwPasses];
        double rw_gx[kRwPasses], rw_gy[kRwPasses];
        if (RW) {
#pragma unroll
            for (int q = 0; q < kRwPasses; ++q) {
                const int je = 4 * q + (lane >> 4), e = je & (NTW - 1), j = je >> (4 - sh);
                const int b = (g0 + j) * NTW + e;
                const bool ok = q < rw_npass && j < NG && g0 + j < a.G && b < B;
                rw_b[q] = ok ? b : -1;
                rw_q[q] = ok ? (j * SLOT) * 4 + (e * DP * kRwCol + (lane & 15)) * 8 : (lane & 15) * 8;
                rw_ns[q] = ok ? (a.n_steps ? min(a.n_steps[b], T) : T) : 0;
                rw_s0[q] = ok && a.step0 ? a.step0[b] : 0;
                rw_gx[q] = ok ? a.goal[2 * (size_t)b] : 0.0;
                rw_gy[q] = ok ? a.goal[2 * (size_t)b + 1] : 0.0;
            }
        }
        // the tile's actions and rewards leave once the consumer has chained it
        auto finish = [&](const int rt, const int gg) {
            const float* sS = smem + ((gg % NB) * NG) * SLOT;
            const int rows = min(16, T - rt * 16);
            if (a.actions) {
                const bool okt = w4 < rows * D;
                const unsigned tb = (unsigned)(rt * SEG) * 4u;
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    if (mover[j] && okt) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(sS + j * SLOT + 2 * kStageStride + rofs);
                        float* const dst = reinterpret_cast<float*>(reinterpret_cast<char*>(bact) + goff[j] + tb);
                        if (a.wt) store16<true>(dst, v);
                        else store16<false>(dst, v);
                    }
                }
            }
            if (RW) {
                const int tl = lane & 15, t = rt * 16 + tl;
                bool mine = false;
#pragma unroll
                for (int q = 0; q < kRwPasses; ++q) mine = mine || (rw_b[q] >= 0 && rw_s0[q] + rt * 16 + 15 >= a.steps_before_reward);
                const bool tile_dist = MPK_RW_ALWAYS_TRIG || __any(mine) != 0;
#pragma unroll 1
                for (int q = 0; q < rw_npass; ++q) {
                    const int pb = q ? rw_b[1] : rw_b[0], pns = q ? rw_ns[1] : rw_ns[0], ps0 = q ? rw_s0[1] : rw_s0[0];
                    const int pq = q ? rw_q[1] : rw_q[0];
                    const bool live = pb >= 0 && t < pns;
                    const bool dist_on = live && ps0 + t >= a.steps_before_reward;
                    const char* sb = reinterpret_cast<const char*>(sS) + pq;
                    const double* qv = reinterpret_cast<const double*>(sb);
                    const double* uv = reinterpret_cast<const double*>(sb + 3 * kStageStride * 4);
                    double r = DC > 0 ? reacher_ctrl_item<DC>(uv, D) : reacher_ctrl_item_d(uv, D);
                    if (MPK_RW_ALWAYS_TRIG || (tile_dist && __any(dist_on) != 0)) {
                        if (DC == 5) r = reacher_reward_item<5>(qv, uv, D, dist_on, q ? rw_gx[1] : rw_gx[0], q ? rw_gy[1] : rw_gy[0]);
                        else r = reacher_reward_item<0>(qv, uv, D, dist_on, q ? rw_gx[1] : rw_gx[0], q ? rw_gy[1] : rw_gy[0]);
                    }
                    r = live ? r : 0.0;
                    if (tl < rows && pb >= 0) {
                        // (write-through like the actions while the launch is cache resident: plain stores leave dirty lines that the END of the
                        // kernel has to write back -- measured on the tail of the launch, not on any wave)
                        float* const rp = reinterpret_cast<float*>(a.rewards + (size_t)pb * T + t);
                        const f32x2 rb = __builtin_bit_cast(f32x2, r);
                        if (a.wt) store8<true>(rp, rb); else store8<false>(rp, rb);
                    }
                }
            }
        };
        // the producer's tiles of this unit: rt = r0, r0 + NP, ...; the inputs of the NEXT one are requested before the wait for the consumer
        // (registers: their round trip to the memory-side cache -- ~1 500 cycles -- passes under the consumer's chain)
        f32x4 lp[NG], lv[NG];
        auto fetch = [&](const int rt) {
            const bool okt = w4 < min(16, T - rt * 16) * D;
            const unsigned tb = (unsigned)(rt * SEG) * 4u;
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                lp[j] = f32x4{0, 0, 0, 0}; lv[j] = lp[j];
                if (mover[j] && okt) {
                    lp[j] = ld_off(reinterpret_cast<const f32x4*>(bpos), goff[j] + tb);
                    lv[j] = ld_off(reinterpret_cast<const f32x4*>(bvel), goff[j] + tb);
                }
            }
        };
        const int r0 = ((p - g) % NP + NP) % NP;          // first tile of this unit with (g + rt) % NP == p
        if (r0 < NRT) fetch(r0);
        for (int rt = r0; rt < NRT; rt += NP) {
            const int gg = g + rt, k = gg / NP;
            float* sS = smem + ((gg % NB) * NG) * SLOT;
            {
                const bool okt = w4 < min(16, T - rt * 16) * D;
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    if (mover[j] && okt) {
                        *reinterpret_cast<f32x4*>(sS + j * SLOT + rofs) = lp[j];
                        *reinterpret_cast<f32x4*>(sS + j * SLOT + kStageStride + rofs) = lv[j];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) flag_store(sSync + p, k + 1);
            if (rt + NP < NRT) fetch(rt + NP);
            if (!flag_wait(sSync + 8, gg + 1, a.fault, 256)) return;     // the consumer has chained it: actions and rewards leave, the slot is free
            finish(rt, gg);
            __builtin_amdgcn_wave_barrier();
        }
        g += NRT;
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
            // the end effector only where some episode of the wave is past steps_before_reward (simple_reacher.py:62-63; wave-uniform)
            const bool dist_on = live && s0 + t >= steps_before_reward;
            if (MPK_RW_ALWAYS_TRIG || __any(dist_on) != 0) {
                const double ang = seg_scan(q, d, D);               // np.cumsum(joint_angles)
                double sn, cs;
                sincos_lean(ang, &sn, &cs);
                double ex = cs, ey = sn, ctrl = u * u;              // unit link lengths (base_reacher.py:19): sums over the links
                seg_scan3(ex, ey, ctrl, d, D);
                if (live && d == D - 1) {
                    double rdist = 0.0;
                    if (dist_on) {
                        const double dx = ex - gx, dy = ey - gy;
                        rdist = 0.0 - sqrt(dx * dx + dy * dy);
                    }
                    rewards[(size_t)b * T + t] = rdist - ctrl;
                }
            } else {
                const double ctrl = seg_scan(u * u, d, D);
                if (live && d == D - 1) rewards[(size_t)b * T + t] = 0.0 - ctrl;
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
                           void* stream, const Tuning& tune, int* fault) {
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int last_rows = T - (T - 1) / 16 * 16;
    // (D = 1: sixteen episodes per group, more than the reward pass's two register sets -- the generic kernel)
    const bool tiles_ok = D >= 2 && D <= kMaxD && (T * D) % 4 == 0 && (last_rows * D) % 4 == 0 && aligned16(des_pos) &&
                          aligned16(des_vel) && (!actions || aligned16(actions)) && tune.pd_simple != 1;
    if (tiles_ok) {
        // the tile-streaming rollout with the reward evaluated per tile by all lanes (see k_pd_rollout_tiles, RW)
        PdArgs pa;
        pa.rc = rc; pa.des_pos = des_pos; pa.des_vel = des_vel; pa.Q = q; pa.QD = qd; pa.n_steps = n_steps;
        pa.actions = actions; pa.D = D; pa.B = B; pa.T = T;
        pa.wt = (double)B * T * D * 12.0 <= kCachedBytes ? 1 : 0;     // desired (pos, vel) + actions stay cached
        if (tune.write_through >= 0) pa.wt = tune.write_through != 0 ? 1 : 0;
        pa.step0 = step0; pa.goal = goal; pa.rewards = rewards; pa.steps_before_reward = steps_before_reward;
        pa.fault = fault;
        int sh = 0;
        while ((1 << sh) < D) ++sh;
        pa.sh = sh;
        const int NTW = 16 >> sh;
        pa.G = (B + NTW - 1) / NTW;
        pa.inv_seg4 = 65536u / (unsigned)(4 * D) + 1u;
        // groups per wave by the waves per SIMD they leave, as launch_pd_rollout ("pd_quad": 0 one, 2 four, 3 two)
        const int quad_mode = tune.pd_quad < 0 ? 1 : tune.pd_quad;
        const long simds = 1024;
        int ng = 1;
        if (quad_mode == 2) ng = 4;
        else if (quad_mode == 3) ng = 2;
        // (second session: the thresholds of launch_pd_rollout -- with the reward 10 240 / 12 288 / 14 336 episodes 85.0 / 85.8 / 100.3 ->
        // 78.6 / 78.9 / 79.4 us with four groups per wave, 3 072: 48.0 -> 42.6 with two; 8 192: two 52.5, four 68)
        // round 6: beyond ~10 groups per SIMD two groups per wave again (one tile of input lookahead: the larger number of waves hides more
        // of the loads) -- LongSimpleReacher + reward, us, four / two: 16 384 episodes 50.2 / 53.9, 24 576: 86.2 / 78.3, 32 768: 118 / 108,
        // 65 536: 229 / 213, 262 144: 871 / 840 (profiles/r06_rollout_reward.md)
        else if (quad_mode == 1) ng = pa.G >= 10 * simds ? 2 : (pa.G >= 5 * simds ? 4 : (2 * pa.G >= 3 * simds ? 2 : 1));
        while (ng > 1 && ng * NTW > 8) ng >>= 1;       // the reward pass holds the inputs of two passes (eight episodes) in registers
        // small launches: the producer / consumer workgroup (k_pd_rollout_pipe: four groups per consumer, so at most two episodes per group),
        // while two workgroups per CU (60 KB of LDS each) hold the launch; "pd_pipe" 1 / 0 forces / forbids
        {
            const long punits = ((long)pa.G + 3) / 4;
            bool pipe = NTW <= 2 && tune.pd_simple != 1 && tune.pd_quad < 0 && punits <= 2L * 256;
            if (tune.pd_pipe >= 0) pipe = tune.pd_pipe == 1 && NTW <= 2;
            if (pipe) {
                const size_t plds = ((size_t)kRollPipeNP * 4 * 5 * kStageStride + 16) * sizeof(float);
                const int pblocks = (int)(punits < 2L * 256 ? punits : 2L * 256);
                auto gop = [&](auto kern) -> int {
                    if (plds > kLdsDefault) {
                        hipError_t e = allow_full_lds(kern);
                        if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
                    }
                    hipLaunchKernelGGL(kern, dim3(pblocks), dim3(64 * (1 + kRollPipeNP)), plds, (hipStream_t)stream, pa);
                    MPK_LAUNCH_CHECK();
                    return MPK_OK;
                };
                if (rc.controller_type == MPK_CTRL_MOTOR && D == 5 && tune.pd_generic != 1) return gop(k_pd_rollout_pipe<4, true, MPK_CTRL_MOTOR, 5>);
                return gop(k_pd_rollout_pipe<4, true, -1, 0>);
            }
        }
        const int units = (pa.G + ng - 1) / ng;
        int blocks = (units + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        if (tune.phase_waves > 0 && blocks > 64 * tune.phase_waves) blocks = 64 * tune.phase_waves;   // (A/B runs: waves per CU)
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
        // helper waves for the control-cost pass (k_pd_rollout_tiles<.., HW>): built, tested, measured SLOWER, therefore only on request
        // ("pd_helper" 1).  LongSimpleReacher + reward, us without / with helpers: 2 048 episodes 19.8 / 20.7, 4 096: 23.5 / 25.0,
        // 8 192: 30.9 / 47.0, 16 384: 50.1 / 76.0, 65 536: 217 / 321 (profiles/r05_rollout.md) -- one workgroup barrier per tile ties
        // four latency-bound chain waves (and the helpers that share their SIMDs) to the slowest of them, and the second float64 action
        // buffer takes a resident workgroup per CU away; the pass it moves off the chain waves is 820 of 4 070 cycles per tile
        const bool hw = MPK_RW_HELPER && tune.pd_helper == 1;
        const size_t lds = hw ? ((size_t)4 * ng * 7 * kStageStride + 2 * 4 * kRwSlots * kRwSlotInts + 8) * sizeof(float)
                              : (size_t)4 * ng * 5 * kStageStride * sizeof(float);
        auto go = [&](auto kern) -> int {
            if (lds > kLdsDefault) {
                hipError_t e = allow_full_lds(kern);
                if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
            }
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(hw ? 384 : 256), lds, (hipStream_t)stream, pa);
            MPK_LAUNCH_CHECK();
            return MPK_OK;
        };
        // the motor controller on 2 / 5 links (the reference's SimpleReacher environments: simple_reacher/mp_wrapper.py:11-16) with
        // controller and link count compiled in; everything else on the run-time form of the same code
        auto by_ng = [&](auto ct_tag, auto dc_tag) -> int {
            constexpr int CT = decltype(ct_tag)::value, DC = decltype(dc_tag)::value;
            if constexpr (MPK_RW_HELPER != 0) {
                if (hw) return ng == 4 ? go(k_pd_rollout_tiles<4, true, CT, DC, true>) : (ng == 2 ? go(k_pd_rollout_tiles<2, true, CT, DC, true>) : go(k_pd_rollout_tiles<1, true, CT, DC, true>));
            }
            return ng == 4 ? go(k_pd_rollout_tiles<4, true, CT, DC>) : (ng == 2 ? go(k_pd_rollout_tiles<2, true, CT, DC>) : go(k_pd_rollout_tiles<1, true, CT, DC>));
        };
        using std::integral_constant;
        if (rc.controller_type == MPK_CTRL_MOTOR && D == 5 && tune.pd_generic != 1) return by_ng(integral_constant<int, MPK_CTRL_MOTOR>(), integral_constant<int, 5>());
        if (rc.controller_type == MPK_CTRL_MOTOR && D == 2 && tune.pd_generic != 1) return by_ng(integral_constant<int, MPK_CTRL_MOTOR>(), integral_constant<int, 2>());
        return by_ng(integral_constant<int, -1>(), integral_constant<int, 0>());
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
                      const int32_t* n_steps, float* actions, int B, int T, void* stream, const Tuning& tune, int* fault) {
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int last_rows = T - (T - 1) / 16 * 16;
    const bool tiles_ok = D >= 1 && D <= kMaxD && (T * D) % 4 == 0 && (last_rows * D) % 4 == 0 && aligned16(des_pos) &&
                          aligned16(des_vel) && (!actions || aligned16(actions)) && tune.pd_simple != 1;
    if (tiles_ok) {
        PdArgs pa;
        pa.rc = rc; pa.des_pos = des_pos; pa.des_vel = des_vel; pa.Q = q; pa.QD = qd; pa.n_steps = n_steps;
        pa.actions = actions; pa.D = D; pa.B = B; pa.T = T;
        pa.wt = (double)B * T * D * 12.0 <= kCachedBytes ? 1 : 0;     // desired (pos, vel) + actions stay cached
        if (tune.write_through >= 0) pa.wt = tune.write_through != 0 ? 1 : 0;
        pa.step0 = nullptr; pa.goal = nullptr; pa.rewards = nullptr; pa.steps_before_reward = 0; pa.fault = fault;
        int sh = 0;
        while ((1 << sh) < D) ++sh;
        pa.sh = sh;
        const int NTW = 16 >> sh;
        pa.G = (B + NTW - 1) / NTW;
        pa.inv_seg4 = 65536u / (unsigned)(4 * D) + 1u;
        // groups per wave (lane quarter j runs group j's recurrence: NG x fewer serial instructions per episode) chosen by the
        // waves per SIMD it leaves (1024 SIMDs): four from 5 groups per SIMD on, two from 1.5, else one ("pd_quad": 0 one, 2 four,
        // 3 two).  Measured (profiles/r04_rollout.md; second session, after the tile loop stopped waiting for its stores, cfg2 shape:
        // 3 072 episodes 12.7 -> 11.3 us with two, 10 240 / 12 288 / 14 336: 24.1 / 26.5 / 30.6 -> 21.6 / 23.2 / 24.9 us with four;
        // 8 192: two 16.8, four 17.1)
        const int quad_mode = tune.pd_quad < 0 ? 1 : tune.pd_quad;
        const long simds = 1024;
        int ng = 1;
        if (quad_mode == 2) ng = 4;
        else if (quad_mode == 3) ng = 2;
        else if (quad_mode == 1) ng = pa.G >= 5 * simds ? 4 : (2 * pa.G >= 3 * simds ? 2 : 1);
        // small launches: the producer / consumer workgroup (k_pd_rollout_pipe), while four workgroups per CU hold the launch
        {
            const long punits = ((long)pa.G + 3) / 4;
            bool pipe = tune.pd_quad < 0 && punits <= 4L * 256;
            if (tune.pd_pipe >= 0) pipe = tune.pd_pipe == 1;
            if (pipe) {
                const size_t plds = ((size_t)kRollPipeNP * 4 * 3 * kStageStride + 16) * sizeof(float);
                const int pblocks = (int)(punits < 4L * 256 ? punits : 4L * 256);
                const dim3 pb(64 * (1 + kRollPipeNP));
                if (D == 7 && tune.pd_generic != 1) hipLaunchKernelGGL((k_pd_rollout_pipe<4, false, -1, 7>), dim3(pblocks), pb, plds, (hipStream_t)stream, pa);
                else if (D == 5 && tune.pd_generic != 1) hipLaunchKernelGGL((k_pd_rollout_pipe<4, false, -1, 5>), dim3(pblocks), pb, plds, (hipStream_t)stream, pa);
                else hipLaunchKernelGGL((k_pd_rollout_pipe<4, false, -1, 0>), dim3(pblocks), pb, plds, (hipStream_t)stream, pa);
                MPK_LAUNCH_CHECK();
                return MPK_OK;
            }
        }
        const int units = (pa.G + ng - 1) / ng;
        int blocks = (units + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        // Beyond the caches ONE workgroup of four waves per CU instead of eight (round 5): the persistent waves walk the units in order,
        // and fewer read / write streams per CU stream faster -- cfg2 shape 131 072 / 262 144 / 524 288 episodes 235 / 497 / 991 ->
        // 225 / 448 / 870 us, LongSimpleReacher shape 65 536 / 131 072: 186 / 361 -> 172 / 328; neutral at 400 MB, slower below
        // (16 384 episodes of the reacher shape: 37 -> 43 us).  Uneven numbers of workgroups per CU lose (192, 384 blocks), two waves per
        // CU lose half.  The reward variant needs its occupancy (794 -> 897 us at 262 144) and keeps eight.  "phase_waves": A/B runs.
        if (tune.phase_waves > 0) { if (blocks > 64 * tune.phase_waves) blocks = 64 * tune.phase_waves; }
        else if ((double)B * T * D * 12.0 >= 512.0 * 1024 * 1024 && blocks > 256) blocks = 256;
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
        const size_t lds = (size_t)4 * ng * 3 * kStageStride * sizeof(float);
        auto by_ng = [&](auto dc_tag) {
            constexpr int DC = decltype(dc_tag)::value;
            if (ng == 4) hipLaunchKernelGGL((k_pd_rollout_tiles<4, false, -1, DC>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, pa);
            else if (ng == 2) hipLaunchKernelGGL((k_pd_rollout_tiles<2, false, -1, DC>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, pa);
            else hipLaunchKernelGGL((k_pd_rollout_tiles<1, false, -1, DC>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, pa);
        };
        if (D == 7 && tune.pd_generic != 1) by_ng(std::integral_constant<int, 7>());
        else if (D == 5 && tune.pd_generic != 1) by_ng(std::integral_constant<int, 5>());
        else by_ng(std::integral_constant<int, 0>());
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

}  // namespace mpk
