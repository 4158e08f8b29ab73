// k_episode_return: ONE launch per plan of a `verbose < 2` episode -- plan, controller, plant, reward and reward aggregation, with
// nothing per step written to memory (round 5).
//
// The reference returns trajectories, step actions and step rewards only when verbose >= 2 (black_box_wrapper.py:160,184,208-213); at
// the default verbose = 1 a step is (obs, reward_aggregation(rewards[:t + 1]), terminated, truncated, {trajectory_length, ..})
// (:215-217).  mpk_replan_step / mpk_trajectory_rollout materialise pos, vel, actions [B, T, D] (8.4 KB per episode at cfg2's shape)
// whatever the caller keeps; with a plant that lives on the GPU none of it needs to leave the CU: 224 bytes in, the
// aggregated reward, the executed steps, the plant state and the replanning state out.  What bounds the launch then is the serial
// chain (float64, ~16 instructions per step and wave), not HBM.
//
// Structure: k_traj_quad's closed-loop form (mpk_traj_quad.h) -- a wave owns NQ consecutive episode groups, per row tile the NQ C
// tiles on the matrix cores into LDS images, lane quarter q runs group q's recurrence (pd_tile_steps: the chain of every closed-loop
// kernel, same bits) -- without the stores; the serial lanes leave the clipped action (and, for the tiles that need the end effector,
// the plant position) as float64 [column][step] images, and all 64 lanes turn (episode, step) items into SimpleReacher rewards
// (mpk_reward.h: the pass of k_pd_rollout_tiles) that each lane ACCUMULATES over the tiles.  Aggregation order (both this kernel and
// mpk_reward_aggregate, which serves the verbose = 2 path, so the two agree bit for bit): per (episode, step slot t mod 16) the sum
// over the row tiles in time order, then the sixteen slots left to right; mean = that sum / executed steps; last = the last executed
// step's reward.  (numpy's np.sum adds pairwise: the same value to a few ulp.)
#include "mpk_reward.h"

namespace mpk {

// Workgroups of four or EIGHT waves (blockDim.x): the waves of a workgroup share one LDS copy of the basis tables, and with the reward
// a wave's images take 16.5 KB at four groups per wave -- two four-wave workgroups with a 23 KB table each (ProMP, 200 steps) do not fit
// a CU's 160 KB, one eight-wave workgroup does: two waves per SIMD instead of one.
template <int MP, int CT, int NQ, int RWD>
__global__ void __launch_bounds__(512) k_episode_return(const TrajArgs a, const ActArgs act, const EpArgs ep) {
    static_assert(CT >= 3 && MP != MPK_MP_DMP, "closed loop; promp / prodmp rows (DMP arrives as its response rows)");
    extern __shared__ __attribute__((aligned(16))) float sDyn[];     // tables | aux | 4 waves x NQ images | 4 waves x slot table
    constexpr int KM = 4;
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = c.KP, TS = a.TS, D = c.D, B = a.B, T = c.T, km = ep.km;
    float* sA = sDyn;
    float* sAux = sDyn + NOUT * KP * TS;
    float* sImg = sAux + TS;                                        // (TS is a multiple of 16 floats)
    constexpr int IMG = RWD ? kEpImg : kEpImgPlain;                  // (without a reward: desired pos | vel only)
    const int WPB = blockDim.x >> 6;
    float* sW = sImg + wave * (NQ * IMG);
    int* sSlot = reinterpret_cast<int*>(sImg + WPB * NQ * IMG) + wave * (kEpSlots * kEpSlotInts);
    const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
    const int NTW = L.NTW, NRT = (T + 15) >> 4, DP = 1 << a.sh;
    const int nslots = NQ * NTW, npass = RWD ? (nslots + 3) >> 2 : 0;
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int ustride = gridDim.x * WPB;
    const int NU = (a.G + NQ - 1) / NQ;
    int u = vb * WPB + wave;

    struct SerialIn { double qs, qds, gx, gy; int nst, s0; bool on; };
    auto load_serial = [&](int uu) {
        SerialIn si{0.0, 0.0, 0.0, 0.0, T, 0, false};
        const int gq = uu * NQ + L.q, bq = gq * NTW + L.bl;
        si.on = L.dvalid && L.q < NQ && gq < a.G && bq < B;
        if (si.on) {
            const size_t ix = (size_t)bq * D + L.d;
            si.qs = a.q_state[ix]; si.qds = a.qd_state[ix];
            if (a.rp.traj_steps) {
                si.s0 = a.rp.traj_steps[bq];                       // (read before the rule below advances it: same lane, program order)
                if (!a.gate_valid) si.nst = replan_rule(a.rp, bq, T, L.d == 0);      // (gated: after the verdict, below)
            } else {
                if (a.n_steps) si.nst = min(a.n_steps[bq], T);
                si.s0 = ep.step0 ? ep.step0[bq] : 0;
            }
            if (RWD && L.d == 0) { si.gx = ep.goal[2 * (size_t)bq]; si.gy = ep.goal[2 * (size_t)bq + 1]; }
        }
        return si;
    };
    GroupIn<KM> nx[NQ];
    SerialIn sn{0.0, 0.0, 0.0, 0.0, T, 0, false};
    if (u < NU) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int g = u * NQ + j;
            nx[j] = load_group<MP, false, KM>(a, L, g < a.G ? g : a.G - 1);
        }
        sn = load_serial(u);
    }
    {   // basis tables + aux row (contiguous in sDyn as in the table slot: [NOUT][KP][TS] rows, then a.aux) -> LDS
        const float4* src = reinterpret_cast<const float4*>(a.A);
        const float4* sx = reinterpret_cast<const float4*>(a.aux);
        float4* dst = reinterpret_cast<float4*>(sA);
        const int nA4 = (NOUT * KP * TS) >> 2, nX4 = TS >> 2;
        for (int i = threadIdx.x; i < nA4 + nX4; i += blockDim.x) dst[i] = i < nA4 ? src[i] : sx[i - nA4];
    }
    __syncthreads();
    if (u >= NU) return;
    const float* ap = sA + L.q * TS + L.col;
    const Gains gq_ = kernarg_gains(L.dvalid ? L.d : 0);
    double pgd = gq_.pg, dgd = gq_.dg;
    const double lod = __builtin_canonicalize(gq_.lo), hid = __builtin_canonicalize(gq_.hi);
    asm volatile("" : "+v"(pgd), "+v"(dgd));                       // waited for once, here (mpk_traj_quad.h)
    (void)act;
    GateLim glim{0.0, 0.0, 0.0f, 0.0f};
    if (a.gate_valid) {
        glim = kernarg_gate(L.dvalid ? L.d : 0);
        asm volatile("" : "+v"(glim.lo), "+v"(glim.hi), "+v"(glim.lo32), "+v"(glim.hi32));
    }

    float xb[NQ][KM];
    while (u < NU) {
        const int g0 = u * NQ;
#pragma unroll
        for (int j = 0; j < NQ; ++j) finish_group<KM>(L, nx[j], xb[j]);
        const SerialIn sc = sn;
        const int un = u + ustride;
        if (un < NU) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int g = un * NQ + j;
                nx[j] = load_group<MP, false, KM>(a, L, g < a.G ? g : a.G - 1);
            }
            sn = load_serial(un);
        }
        const int gq = g0 + L.q, bq = gq * NTW + L.bl;
        const bool serial = sc.on;
        const int oq = L.bl * a.pitch + L.d;                        // (row 0, this column) in group q's desired images
        float* sQ = sW + L.q * IMG;
        double qs = sc.qs, qds = sc.qds;
        int nst_ = sc.nst, s0_ = sc.s0;
        // validity gate: this kernel stores nothing per step, so the plan is judged WHILE it runs -- the position C tiles of the main loop
        // go through the scan (gate_scan_tile, mpk_tile.h), the verdict falls after the last tile, and an invalid plan is simply not
        // committed: plant state, integer state, aggregated reward as if nothing had been executed (k_traj_quad, which stores actions,
        // takes a separate first pass instead)
        const bool gated = a.gate_valid != nullptr;
        ReplanVals rv{sc.nst, 0, 0, false};
        double tpen = 0.0;
        bool t_bad = false;
        GateScan<NQ> gs;
        gs.init(glim);
        if (gated && serial) {
            if (a.rp.traj_steps) { rv = replan_eval(a.rp, bq, T); nst_ = rv.seg; }
            if (a.gate_check_td) {
                const double tau = (double)a.gate_raw[(size_t)bq * c.P], delay = (double)a.gate_raw[(size_t)bq * c.P + 1];
                t_bad = !(tau >= a.gate_tb[0] && tau <= a.gate_tb[1] && delay >= a.gate_db[0] && delay <= a.gate_db[1]);
                tpen = 3.0 * (fmax(0.0, tau - a.gate_tb[1]) + fmax(0.0, a.gate_tb[0] - tau)) +
                       3.0 * (fmax(0.0, delay - a.gate_db[1]) + fmax(0.0, a.gate_db[0] - delay));
            }
        }
        asm volatile("" : "+v"(qs), "+v"(qds), "+v"(nst_), "+v"(s0_));
        const int nst = nst_;
        const int tcond = a.rp.cond_pos ? min(max(nst - 1, 0), T - 1) : -1;
        // the unit's episode slots (group q, episode bl -> slot q NTW + bl): what the reward pass needs per episode, written by the
        // episode's d == 0 lane; first step at which ANY episode of the unit carries the reward's distance term (wave-uniform)
        int tdist = 1 << 30;
        if (RWD) {
            if (L.d == 0 && L.q < NQ) {
                int* sl = sSlot + (L.q * NTW + L.bl) * kEpSlotInts;
                sl[0] = serial ? nst : 0; sl[1] = s0_; sl[2] = serial ? bq : -1; sl[3] = 0;
                *reinterpret_cast<double*>(sl + 4) = sc.gx; *reinterpret_cast<double*>(sl + 6) = sc.gy;
                if (serial && nst > 0) tdist = max(ep.steps_before_reward - s0_, 0);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) tdist = min(tdist, __shfl_xor(tdist, m));
        }
        if (ep.seg_out && serial && L.d == 0 && !gated) ep.seg_out[bq] = nst;
        double racc[kEpMaxPass];
#pragma unroll
        for (int p = 0; p < kEpMaxPass; ++p) racc[p] = 0.0;
        float afn[NOUT][KM];
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int m = 0; m < KM; ++m) afn[o][m] = m < km ? ap[(o * KP + 4 * m) * TS] : 0.0f;
        __builtin_amdgcn_wave_barrier();
        // (tiles behind every episode's last executed step and behind the condition step carry nothing anybody reads)
        int nmax = serial ? max(nst, tcond + 1) : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) nmax = max(nmax, __shfl_xor(nmax, m));
        const int nrt_live = gated ? NRT : min(NRT, (nmax + 15) >> 4);      // (the gate has to see the whole plan)
        float row0p = 0.0f, row0v = 0.0f;               // gate: the desired state of step 0 (condition of an episode that executes nothing)
        for (int rt = 0; rt < nrt_live; ++rt) {
            const int rows = min(16, T - rt * 16);
            float af[NOUT][KM];
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int m = 0; m < KM; ++m) af[o][m] = afn[o][m];
            if (rt + 1 < nrt_live) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
#pragma unroll
                    for (int m = 0; m < KM; ++m) afn[o][m] = m < km ? ap[(o * KP + 4 * m) * TS + (rt + 1) * 16] : 0.0f;
            }
            // 1. NQ C tiles on the matrix cores -> desired (pos, vel) images
            bool tile_viol = false;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (g0 + j < a.G) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < KM; ++m) {
                        if (m < km) {                              // (wave-uniform)
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][m], xb[j][m], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][m], xb[j][m], acc1, 0, 0, 0);
                            if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[NOUT > 2 ? 2 : 0][m], xb[j][m], acc2, 0, 0, 0);
                        }
                    }
                    if (L.dvalid) {
                        float dtd[4] = {1.f, 1.f, 1.f, 1.f};
                        if (MP == MPK_MP_PROMP) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) dtd[r] = sAux[rt * 16 + 4 * L.q + r];
                        }
                        tile_epilogue<MP, -1>(acc0, acc1, acc2, dtd, 0.0, 0.0, Gains{0.0, 0.0, 0.0, 0.0}, sW + j * IMG, L.wofs, D);
                    }
                    if (gated) tile_viol = gate_scan_tile<NQ>(gs, j, acc0, rt, L.q, L.dvalid && (g0 + j) * NTW + L.bl < B, glim, T) || tile_viol;
                }
            }
            if (gated && rt < 64 && __any(tile_viol) != 0) gs.tiles |= 1ull << rt;
            __builtin_amdgcn_wave_barrier();
            // 2. the recurrences, one group per lane quarter; float64 [column][step] images for the reward pass
            const bool tile_dist = RWD && rt * 16 + 15 >= tdist;
            if (gated && rt == 0 && serial) { row0p = sQ[oq]; row0v = sQ[kStageStride + oq]; }
            if (serial && rt * 16 < max(nst, tcond + 1)) {
                if (tcond >= rt * 16 && tcond < rt * 16 + 16) {     // condition_on_desired: the desired state at the last executed step
                    const size_t si = (size_t)bq * D + L.d;
                    a.rp.cond_pos[si] = sQ[oq + (tcond - rt * 16) * D];
                    a.rp.cond_vel[si] = sQ[kStageStride + oq + (tcond - rt * 16) * D];
                }
                double* q64 = reinterpret_cast<double*>(sQ) + L.col * kRwCol;
                double* u64 = reinterpret_cast<double*>(sQ + 2 * kStageStride) + L.col * kRwCol;
                const bool full_tile = rows == 16 && __all(!serial || nst >= rt * 16 + 16) != 0;
                auto go = [&](auto keep_tag) {
                    constexpr int KEEP = decltype(keep_tag)::value;
                    if (full_tile)
                        pd_tile_steps<CT - 3, false, true, KEEP, 0, false>(sQ + oq, sQ + kStageStride + oq, nullptr, D, rt * 16, nst, pgd, dgd,
                                                                          lod, hid, a.plant_dt, qs, qds, q64, u64);
                    else
                        pd_tile_steps<CT - 3, true, true, KEEP, 0, false>(sQ + oq, sQ + kStageStride + oq, nullptr, D, rt * 16, nst, pgd, dgd,
                                                                         lod, hid, a.plant_dt, qs, qds, q64, u64, rows);
                };
                if (!RWD) go(std::integral_constant<int, 0>());
                else if (tile_dist) go(std::integral_constant<int, 1>());
                else go(std::integral_constant<int, 2>());
            }
            __builtin_amdgcn_wave_barrier();
            // 3. rewards of the tile's (episode, step) items, accumulated per lane
            if (RWD) {
                const int tl = lane & 15, t = rt * 16 + tl;
                static_assert(kEpMaxPass == 2, "the accumulator selects below");
#pragma unroll 1
                for (int p = 0; p < npass; ++p) {                  // (one copy of the pass: two side by side cost 256 registers)
                    {
                        const int s = 4 * p + (lane >> 4);
                        const int ss = s < nslots ? s : 0;
                        const int* sl = sSlot + ss * kEpSlotInts;
                        const int pns = sl[0], ps0 = sl[1], pb = sl[2];
                        const bool live = s < nslots && pb >= 0 && t < pns;
                        const bool dist_on = live && ps0 + t >= ep.steps_before_reward;
                        const int j = ss >> (4 - a.sh), e = ss & (NTW - 1);
                        const float* img = sW + j * IMG;
                        const double* qv = reinterpret_cast<const double*>(img) + e * DP * kRwCol + tl;
                        const double* uv = reinterpret_cast<const double*>(img + 2 * kStageStride) + e * DP * kRwCol + tl;
                        double r = reacher_ctrl_item_d(uv, D);
                        if (tile_dist && __any(dist_on) != 0)
                            r = reacher_reward_item<0>(qv, uv, D, dist_on, *reinterpret_cast<const double*>(sl + 4), *reinterpret_cast<const double*>(sl + 6));
                        r = live ? r : 0.0;
                        const double old = p ? racc[1] : racc[0];
                        const double upd = ep.agg == 2 ? ((live && t == pns - 1) ? r : old) : old + r;
                        if (p) racc[1] = upd; else racc[0] = upd;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        bool invalid = false;
        if (gated) {
            float xz[NQ][KM];
#pragma unroll
            for (int j = 0; j < NQ; ++j)
#pragma unroll
                for (int m = 0; m < KM; ++m) xz[j][m] = g0 + j < a.G ? xb[j][m] : 0.0f;
            double over, under;
            const bool p_bad = gate_verdict<KM, NQ>(a, L, gs, ap, TS, km, xz, g0, glim, over, under);
            invalid = serial && (p_bad || t_bad);
            if (serial) {
                const size_t si = (size_t)bq * D + L.d;
                if (invalid && a.rp.cond_pos) { a.rp.cond_pos[si] = row0p; a.rp.cond_vel[si] = row0v; }
                if (L.d == 0) {
                    a.gate_valid[bq] = invalid ? 0 : 1;
                    const double n = (double)(T * D);
                    if (a.gate_penalty) a.gate_penalty[bq] = -(tpen + over / n + under / n);
                    if (a.rp.traj_steps) replan_write(a.rp, bq, rv, !invalid);
                    if (ep.seg_out) ep.seg_out[bq] = invalid ? 0 : nst;
                }
            }
        }
        if (serial && !invalid) {
            const size_t si = (size_t)bq * D + L.d;
            a.q_state[si] = qs; a.qd_state[si] = qds;
        }
        if (RWD && gated) {                             // (an invalid plan executed nothing: its aggregated reward is 0)
            if (invalid && L.d == 0 && L.q < NQ) sSlot[(L.q * NTW + L.bl) * kEpSlotInts + 3] = 1;
            __builtin_amdgcn_wave_barrier();
        }
        // the aggregated reward: the sixteen step slots of an episode left to right (see the header of this file)
        if (ep.ret) {
            if (RWD) {
                const int base = lane & 48;
#pragma unroll 1
                for (int p = 0; p < npass; ++p) {
                    {
                        const double mine = p ? racc[1] : racc[0];
                        double acc = __shfl(mine, base);
#pragma unroll
                        for (int i = 1; i < 16; ++i) acc = acc + __shfl(mine, base + i);
                        const int s = 4 * p + (lane >> 4);
                        if ((lane & 15) == 0 && s < nslots) {
                            const int* sl = sSlot + s * kEpSlotInts;
                            const int pns = sl[0], pb = sl[2];
                            if (pb >= 0) ep.ret[pb] = sl[3] != 0 ? 0.0 : (ep.agg == 1 ? (pns > 0 ? acc / (double)pns : 0.0) : acc);
                        }
                    }
                }
            } else if (serial && L.d == 0) {
                ep.ret[bq] = 0.0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        u = un;
    }
}

#ifndef MPK_DEVICE_ONLY
template <int MP>
int launch_episode_kernel(const TrajArgs& ta, const ActArgs& aa, const EpArgs& ea, int ct, int nq, int rwd, int blocks, size_t lds,
                          void* stream) {
    const int wpb = ea.wpb;
    if constexpr (MP == MPK_MP_DMP) {
        (void)wpb; (void)ta; (void)aa; (void)ea; (void)ct; (void)nq; (void)rwd; (void)blocks; (void)lds; (void)stream;
        set_error("internal: the episode kernel takes promp / prodmp rows");
        return MPK_EINVAL;
    } else {
        auto go = [&](auto kern) -> int {
            if (lds > kLdsDefault) {
                hipError_t e = allow_full_lds(kern);
                if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return MPK_EHIP; }
            }
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * wpb), lds, (hipStream_t)stream, ta, aa, ea);
            MPK_LAUNCH_CHECK();
            return MPK_OK;
        };
        auto by_ct = [&](auto nq_tag, auto rw_tag) -> int {
            constexpr int NQ = decltype(nq_tag)::value, RW = decltype(rw_tag)::value;
            switch (ct) {
                case 3 + MPK_CTRL_MOTOR: return go(k_episode_return<MP, 3 + MPK_CTRL_MOTOR, NQ, RW>);
                case 3 + MPK_CTRL_VELOCITY: return go(k_episode_return<MP, 3 + MPK_CTRL_VELOCITY, NQ, RW>);
                default: return go(k_episode_return<MP, 3 + MPK_CTRL_POSITION, NQ, RW>);
            }
        };
        using std::integral_constant;
        if (nq == 4) return rwd ? by_ct(integral_constant<int, 4>(), integral_constant<int, 1>()) : by_ct(integral_constant<int, 4>(), integral_constant<int, 0>());
        if (nq == 2) return rwd ? by_ct(integral_constant<int, 2>(), integral_constant<int, 1>()) : by_ct(integral_constant<int, 2>(), integral_constant<int, 0>());
        return rwd ? by_ct(integral_constant<int, 1>(), integral_constant<int, 1>()) : by_ct(integral_constant<int, 1>(), integral_constant<int, 0>());
    }
}
#ifdef MPK_MP_UNIT
template int launch_episode_kernel<MPK_MP_UNIT>(const TrajArgs&, const ActArgs&, const EpArgs&, int, int, int, int, size_t, void*);
#else
template int launch_episode_kernel<MPK_MP_PROMP>(const TrajArgs&, const ActArgs&, const EpArgs&, int, int, int, int, size_t, void*);
template int launch_episode_kernel<MPK_MP_DMP>(const TrajArgs&, const ActArgs&, const EpArgs&, int, int, int, int, size_t, void*);
template int launch_episode_kernel<MPK_MP_PRODMP>(const TrajArgs&, const ActArgs&, const EpArgs&, int, int, int, int, size_t, void*);
#endif
#endif  // MPK_DEVICE_ONLY

}  // namespace mpk
