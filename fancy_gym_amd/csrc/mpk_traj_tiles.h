// k_traj_tiles / k_traj_split: the tile-major shared-phase trajectory kernels
#pragma once
#include "mpk_tile.h"

namespace mpk {

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
    const int wid = vb * a.wpb + wave;
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
        // (the epilogue with the DoF count as a compile-time constant -- immediates instead of address arithmetic, what moved
        // k_traj_ring -- measured 8.67 vs 8.66 us here: this launch is not instruction-issue bound.  profiles/r04_headline_ab.md)
        if (L.dvalid)
            tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, cp, cv, kg, sSt, L.wofs + ep_shift(a, g * L.NTW + L.bl), D);
        __builtin_amdgcn_wave_barrier();
        MPK_STAMP(10);
#if MPK_TILES_COLLECT_FIRST
        // (A/B build knob: the next group's fragments collected BEFORE this group's stores, so that the wait for the prefetched loads
        // does not cover these stores -- the queue retires in order)
        finish_group<KM>(L, nxt, xb);
        asm volatile("" : "+v"(xb[0]));
        tile_store<NST, KM, WT>(a, L, sSt, lane, g * L.NTW, rt, rows);
        __builtin_amdgcn_wave_barrier();
        MPK_STAMP(11);
#else
        tile_store<NST, KM, WT>(a, L, sSt, lane, g * L.NTW, rt, rows);
        __builtin_amdgcn_wave_barrier();
        MPK_STAMP(11);
        // 5. finish the prefetched fragments for the next iteration
        finish_group<KM>(L, nxt, xb);
#endif
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
#ifndef MPK_TILES_COLLECT_FIRST
#define MPK_TILES_COLLECT_FIRST 0
#endif
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

}  // namespace mpk
