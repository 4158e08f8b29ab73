// k_traj_quad / duo / mono: the serial-recurrence variants, several groups per wave
#pragma once
#include "mpk_tile.h"

namespace mpk {

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

// waves per SIMD the register allocation aims at, per groups-per-wave variant (A/B builds: -DMPK_QUAD_WPE4=.. etc.)
#ifndef MPK_QUAD_WPE4
#define MPK_QUAD_WPE4 2      // four groups: 271 registers unconstrained = ONE wave per SIMD; 256 without a spill = two (closed loop at 16 384: 37.9 -> 31.0 us)
#endif
#ifndef MPK_QUAD_WPE2
#define MPK_QUAD_WPE2 1      // (3 together with MPK_QUAD_LAUNDER_ALL: 160 registers, measured and not kept -- see the stores below)
#endif
#ifndef MPK_QUAD_WPE1
#define MPK_QUAD_WPE1 1
#endif
#ifndef MPK_QUAD_LAUNDER_ALL
#define MPK_QUAD_LAUNDER_ALL 0
#endif

template <int MP, int CT, int KM, int NQ>
__global__ void __launch_bounds__(256, (NQ == 4 ? MPK_QUAD_WPE4 : NQ == 2 ? MPK_QUAD_WPE2 : MPK_QUAD_WPE1)) k_traj_quad(const TrajArgs a, const ActArgs act) {
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
    const int vb = (gridDim.x & 7) == 0 && !a.inorder ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
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
                // (with the validity gate the integer rule waits for the verdict: evaluated at the unit's start, written after gate_pass)
                if (a.rp.traj_steps) { if (!a.gate_valid) si.nst = replan_rule(a.rp, bq, T, L.d == 0); }
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
        // the gains arrive by vector loads: make the compiler wait for them HERE.  Left to it, the wait sits in front of their first
        // use in every row tile's chain, and -- the loop being a loop -- it is `s_waitcnt vmcnt(0)`: every tile's recurrence began by
        // waiting for the acknowledgement of the previous tile's STORES (round 4, second session: the disassembly of the tile loop)
        asm volatile("" : "+v"(pgd), "+v"(dgd));
    }
    (void)act;
    GateLim glim{0.0, 0.0, 0.0f, 0.0f};
    if (CLOSED && a.gate_valid) {
        glim = kernarg_gate(L.dvalid ? L.d : 0);
        asm volatile("" : "+v"(glim.lo), "+v"(glim.hi), "+v"(glim.lo32), "+v"(glim.hi32));
    }

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
        int nst_ = sc.nst;
        bool inv = false;               // gate: this lane's plan is invalid (its chain runs all the same: below)
        if (CLOSED && a.gate_valid) {
            // validity gate: judge the unit's plans first (gate_pass, mpk_tile.h); an invalid plan executes nothing
            ReplanVals rv{sc.nst, 0, 0, false};
            double tpen = 0.0;
            bool t_bad = false;
            if (serial) {
                if (a.rp.traj_steps) rv = replan_eval(a.rp, bq, T);
                if (a.gate_check_td) {
                    const double tau = (double)a.gate_raw[(size_t)bq * P], delay = (double)a.gate_raw[(size_t)bq * P + 1];
                    t_bad = !(tau >= a.gate_tb[0] && tau <= a.gate_tb[1] && delay >= a.gate_db[0] && delay <= a.gate_db[1]);
                    tpen = 3.0 * (fmax(0.0, tau - a.gate_tb[1]) + fmax(0.0, a.gate_tb[0] - tau)) +
                           3.0 * (fmax(0.0, delay - a.gate_db[1]) + fmax(0.0, a.gate_db[0] - delay));
                }
            }
            double over, under;
            const bool p_bad = gate_pass<KM, NQ>(a, L, ap, TS, KM, xb, g0, glim, over, under);
            const bool invalid = serial && (p_bad || t_bad);
            // An invalid plan executes nothing -- but its lanes run the chain like their neighbours' and DROP the result (plant state not
            // written, actions clipped to [0, 0]): with nst = 0 beside lanes that execute the whole plan, every tile of the wave was a
            // masked tile (ballot search + three selects per step, 1.75 x a full tile) -- cfg5 at 8 192 episodes, two invalid plans:
            // 53.6 -> 79.8 us for the LAUNCH (profiles/r06_finished_episodes.md; k_phase_fused and k_traj_pipe run speculatively anyway)
            inv = invalid;
            nst_ = rv.seg;
            if (serial && L.d == 0) {
                a.gate_valid[bq] = invalid ? 0 : 1;
                const double n = (double)(T * D);
                if (a.gate_penalty) a.gate_penalty[bq] = -(tpen + over / n + under / n);
                if (a.rp.traj_steps) replan_write(a.rp, bq, rv, !invalid);
            }
        }
        float ey = sc.ey, ez = sc.ez, eg = sc.eg;
        // (the same for the unit's serial inputs, fetched one unit ago: waited for once per unit, not in every tile's chain)
        if (CLOSED) asm volatile("" : "+v"(qs), "+v"(qds), "+v"(nst_));
        else asm volatile("" : "+v"(ey), "+v"(ez), "+v"(eg));
        const int nst = nst_;
        const int tcond = (CLOSED && a.rp.cond_pos) ? (inv ? 0 : min(max(nst - 1, 0), T - 1)) : -1;      // (invalid: the desired state of step 0)
        // (an invalid lane's actions are clipped to [0, 0]: zeros in the action image without a write of their own)
        const double lod_u = inv ? 0.0 : lod, hid_u = inv ? 0.0 : hid;
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
            // 1. four C tiles on the matrix cores -> four staging images   ("ring_dbg" 1 / 2: ablations for measurements -- no
            //    production / no stores; outputs are then unwritten or wrong)
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (g0 + j < a.G && !(a.ring_dbg & 1)) {
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
            if (serial && !(a.ring_dbg & 1) && (!CLOSED || rt * 16 < max(nst, tcond + 1))) {
                if (CLOSED) {
                    if (tcond >= rt * 16 && tcond < rt * 16 + 16) {   // condition_on_desired: the desired state at the
                        const size_t si = (size_t)bq * D + L.d;          // last executed step
                        a.rp.cond_pos[si] = sQ[oq + (tcond - rt * 16) * D];
                        a.rp.cond_vel[si] = sQ[kStageStride + oq + (tcond - rt * 16) * D];
                    }
                    if (full_tile)
                        pd_tile_steps<(CLOSED ? CT - 3 : 0), false>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D,
                                                                    rt * 16, nst, pgd, dgd, lod_u, hid_u, a.plant_dt, qs, qds);
                    else
                        pd_tile_steps<(CLOSED ? CT - 3 : 0), true>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D,
                                                                   rt * 16, nst, pgd, dgd, lod_u, hid_u, a.plant_dt, qs, qds, nullptr, nullptr, rows);
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
                if (g0 + j < a.G && !(a.ring_dbg & 2))
                {
                    // four groups per wave: the group's first episode goes through an opaque scalar, so that what the store derives
                    // from it (addresses and range predicates of 4 groups x 3 arrays) is recomputed per tile instead of living in
                    // ~50 registers across the recurrence: closed loop 271 -> 175 registers, dmp 176 -> 134 = two / three waves per
                    // SIMD (cfg3 at 16 384: 34.0 -> 32.1 us).  With two groups or one the recomputation costs more than the third
                    // wave per SIMD buys below 16 384 episodes (dmp at 4 096: 18.6 -> 20.0 us) and the extra resident waves open more
                    // output streams at HBM-streaming sizes (dmp at 65 536: 184 -> 199 us): profiles/r04_closed_loop.md
                    int b0 = (g0 + j) * NTW;
                    if (NQ == 4 || MPK_QUAD_LAUNDER_ALL) asm volatile("" : "+s"(b0));
                    if (a.wt) tile_store<NST, KM, true>(a, L, sW + j * kQuadImg, lane, b0, rt, rows);
                    else tile_store<NST, KM, false>(a, L, sW + j * kQuadImg, lane, b0, rt, rows);
                }
            __builtin_amdgcn_wave_barrier();
            MPK_STAMP(50 + rt);
        }
        if (CLOSED) {
            if (serial && !inv) {
                const size_t si = (size_t)bq * D + L.d;
                a.q_state[si] = qs; a.qd_state[si] = qds;
            }
        }
        MPK_STAMP(90);
        u = un;
    }
}

}  // namespace mpk
