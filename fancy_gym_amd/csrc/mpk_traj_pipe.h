// k_traj_pipe: the closed-loop step as a wave-specialised producer / consumer pipeline
#pragma once
#include "mpk_tile.h"

namespace mpk {

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
// Validity gate (round 6; a.gate_valid): the consumer's chain tests the desired positions it reads anyway (pd_tile_steps' GATE hook, as
// k_phase_fused), the rollout runs speculatively over the WHOLE plan, the verdict falls after the last tile; an invalid plan is taken
// back: plant state not written, replan_write(valid = false), condition = row 0, and -- after one more barrier -- the producer of its
// group rewrites its action rows as zeros (the same wave that stored them: ordered).  Until then gated launches of a few thousand
// episodes ran on k_traj_mono<.., gate> (cfg5 at 1 024 episodes: 37 us against 25 ungated).
constexpr int kPipeGroups = 4;
#ifndef MPK_PIPE_PRE
#define MPK_PIPE_PRE 0       // 1: the consumer pulls and converts a whole tile before its chain (pd_tile_steps): 14 instead of 17
                             // instructions per step, no LDS wait inside the tile -- and the same 2 050 cycles per tile (a lone
                             // wave issues one instruction per 6 cycles, a dependent one per 8.9: profiles/r04_closed_loop.md)
#endif

#ifndef MPK_PIPE_LEAN_WAVES
#define MPK_PIPE_LEAN_WAVES 0    // 6: the register-lean instantiation forced to six waves per SIMD (80 registers + 16 B of scratch) so that a
                                 // FOURTH workgroup per CU is resident.  Measured slower everywhere (6 144: 13.1 -> 13.8 us, 8 192: 20.2 -> 23.9):
                                 // A/B build knob, off (profiles/r04_serial_quantization.md)
#endif
#if MPK_PIPE_LEAN_WAVES
#define MPK_PIPE_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(LEAN ? MPK_PIPE_LEAN_WAVES : 4, LEAN ? MPK_PIPE_LEAN_WAVES : 10)))
#else
#define MPK_PIPE_WAVES_ATTR
#endif
template <int MP, int CT, int KM, bool LEAN, bool GATE = false>      // GATE: its own instantiations -- the gate's chain costs the consumer 154 registers where the plain kernel lives on 81 - 103
__global__ void __launch_bounds__(320) MPK_PIPE_WAVES_ATTR k_traj_pipe(const TrajArgs a, const ActArgs act) {
    static_assert(CT >= 3 && MP != MPK_MP_DMP, "closed loop, promp / prodmp");
    __shared__ __attribute__((aligned(16))) float smem[2 * kPipeGroups * kQuadImg];   // [buffer][group] pos | vel | act
    __shared__ int sBad[kPipeGroups * 16];                                            // gate: [group][episode of the group] invalid
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
        MPK_STAMP(2);
        // ---------------- consumer: four recurrences, one per lane quarter ----------------
        const Gains gq = kernarg_gains(L.dvalid ? L.d : 0);
        const double pgd = gq.pg, dgd = gq.dg, lod = __builtin_canonicalize(gq.lo), hid = __builtin_canonicalize(gq.hi);
        constexpr bool gated = GATE;
        GateLim glim{0.0, 0.0, 0.0f, 0.0f};
        if (gated) glim = kernarg_gate(L.dvalid ? L.d : 0);
        for (int u = vb; u < NU; u += (int)gridDim.x) {
            const int gsel = u * kPipeGroups + L.q, bq = gsel * NTW + L.bl;
            const bool serial = L.dvalid && gsel < a.G && bq < B;
            double qs = 0.0, qds = 0.0;
            int nst = 0;
            ReplanVals rv{T, 0, 0, false};
            bool t_bad = false, p_bad = false;
            double tpen = 0.0, over = 0.0, under = 0.0;
            float row0p = 0.0f, row0v = 0.0f;
            if (serial) {
                const size_t ix = (size_t)bq * D + L.d;
                qs = a.q_state[ix]; qds = a.qd_state[ix];
                nst = T;
                if (a.rp.traj_steps) {
                    if (gated) { rv = replan_eval(a.rp, bq, T); nst = rv.seg; }     // (written after the verdict)
                    else nst = replan_rule(a.rp, bq, T, L.d == 0);
                } else if (a.n_steps) nst = min(a.n_steps[bq], T);
                if (gated && a.gate_check_td) {
                    const double tau = (double)a.gate_raw[(size_t)bq * c.P], delay = (double)a.gate_raw[(size_t)bq * c.P + 1];
                    t_bad = !(tau >= a.gate_tb[0] && tau <= a.gate_tb[1] && delay >= a.gate_db[0] && delay <= a.gate_db[1]);
                    tpen = 3.0 * (fmax(0.0, tau - a.gate_tb[1]) + fmax(0.0, a.gate_tb[0] - tau)) +
                           3.0 * (fmax(0.0, delay - a.gate_db[1]) + fmax(0.0, a.gate_db[0] - delay));
                }
            }
            const int tcond = (serial && a.rp.cond_pos) ? min(max(nst - 1, 0), T - 1) : -1;
            const int oq = L.bl * a.pitch + L.d + (int)ep_shift(a, bq);      // (row 0, this column) in group q's image
            __syncthreads();                                                // tile 0 is in image 0
            MPK_STAMP(3);
            for (int rt = 0; rt < NRT; ++rt) {
                float* sQ = smem + ((rt & 1) * kPipeGroups + L.q) * kQuadImg;
                const bool full_tile = tile_fully_executed(serial, nst, rt * 16);
                MPK_STAMP(10 + rt);
                if (gated && rt == 0 && serial) { row0p = sQ[oq]; row0v = sQ[kStageStride + oq]; }
                if (serial && (gated || rt * 16 < max(nst, tcond + 1))) {   // (the gate has to see the whole plan)
                    if (tcond >= rt * 16 && tcond < rt * 16 + 16) {   // condition_on_desired: the desired state at the
                        const size_t si = (size_t)bq * D + L.d;          // last executed step
                        a.rp.cond_pos[si] = sQ[oq + (tcond - rt * 16) * D];
                        a.rp.cond_vel[si] = sQ[kStageStride + oq + (tcond - rt * 16) * D];
                    }
                    if constexpr (gated) {
                        int tb = 0;
                        double gsum[2] = {over, under};
                        if (full_tile)
                            pd_tile_steps<CT - 3, false, true, 0, 0, true, true>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D, rt * 16, nst,
                                                                               pgd, dgd, lod, hid, a.plant_dt, qs, qds, nullptr, nullptr, 16, glim.lo32, glim.hi32,
                                                                               &tb, glim.lo, glim.hi, gsum);
                        else
                            pd_tile_steps<CT - 3, true, true, 0, 0, true, true>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D, rt * 16, nst,
                                                                              pgd, dgd, lod, hid, a.plant_dt, qs, qds, nullptr, nullptr, min(16, T - rt * 16),
                                                                              glim.lo32, glim.hi32, &tb, glim.lo, glim.hi, gsum);
                        if (tb) p_bad = true;
                        over = gsum[0]; under = gsum[1];
                    } else if (full_tile)
                        pd_tile_steps<CT - 3, false, true, false, MPK_PIPE_PRE>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D, rt * 16, nst,
                                                     pgd, dgd, lod, hid, a.plant_dt, qs, qds);
                    else
                        pd_tile_steps<CT - 3, true, true, false, MPK_PIPE_PRE>(sQ + oq, sQ + kStageStride + oq, sQ + 2 * kStageStride + oq, D, rt * 16, nst,
                                                    pgd, dgd, lod, hid, a.plant_dt, qs, qds, nullptr, nullptr, min(16, T - rt * 16));
                }
                MPK_STAMP(30 + rt);
                __syncthreads();                                            // tile rt's actions are final; tile rt + 1 is in
            }
            MPK_STAMP(60);
            bool invalid = false;
            if constexpr (gated) {
                // the episode's D lanes sit side by side in lane quarter q: its verdict, and its excess sums left to right
                const int base = L.q * 16 + (L.bl << a.sh);
                const unsigned long long m = __ballot(serial && p_bad);
                invalid = serial && ((((m >> base) & ((1ull << D) - 1ull)) != 0ull) || t_bad);
                if (m != 0ull) {
                    double so = 0.0, su = 0.0;
                    for (int d = 0; d < D; ++d) { so += __shfl(over, base + d); su += __shfl(under, base + d); }
                    over = so; under = su;
                }
                if (serial) {
                    const size_t si = (size_t)bq * D + L.d;
                    if (invalid && a.rp.cond_pos) { a.rp.cond_pos[si] = row0p; a.rp.cond_vel[si] = row0v; }
                    if (L.d == 0) {
                        a.gate_valid[bq] = invalid ? 0 : 1;
                        const double n = (double)(T * D);
                        if (a.gate_penalty) a.gate_penalty[bq] = -(tpen + over / n + under / n);
                        if (a.rp.traj_steps) replan_write(a.rp, bq, rv, !invalid);
                    }
                }
                if (L.dvalid && L.d == 0) sBad[L.q * 16 + L.bl] = invalid ? 1 : 0;
                __syncthreads();                                            // the verdicts are the producers' (zeroed action rows)
            }
            if (serial && !invalid) {
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
                // LEAN: the group's first episode through an opaque scalar -- the stores' addresses and range predicates are recomputed
                // per tile instead of living across the unit: 100 -> 81 registers = FIVE waves per SIMD, i.e. a third (and fourth)
                // 5-wave workgroup per CU.  With the 100-register version the workgroups of a launch with three units per CU did
                // not all fit (104-register granules: 4 waves per SIMD, the fifth wave of a workgroup doubles up on one SIMD):
                // B = 6 144: 18.7 -> 13.7 us, replanning step 15.6 -> 11.9.  With <= two units per CU the recomputation only costs
                // (4 096: 11.3 -> 11.6 us, replanning step 9.6 -> 10.1): the launcher picks (profiles/r04_closed_loop.md)
                int b0 = g * NTW;
                if (LEAN) asm volatile("" : "+s"(b0));
                if (a.wt) tile_store_sel<MASK, KM, true>(a, L, sJ, lane, b0, rt, rows);
                else tile_store_sel<MASK, KM, false>(a, L, sJ, lane, b0, rt, rows);
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
                MPK_STAMP_AT(110 + rt, 64);
                __syncthreads();                                            // tile rt's actions are final
                MPK_STAMP_AT(130 + rt, 64);
                if (have) store_arrays(std::integral_constant<int, 4>(), rt);
                MPK_STAMP_AT(150 + rt, 64);
            }
            if constexpr (GATE) {
                __syncthreads();                                            // the consumer's verdicts
                if (have) {
                    for (int e = 0; e < NTW; ++e) {
                        const int be = g * NTW + e;
                        if (be >= B || !sBad[j * 16 + e]) continue;             // (wave-uniform)
                        float* zp = a.actions + (size_t)be * T * D;
                        for (int i = lane; i < T * D; i += 64) {
                            if (a.wt) store4<true>(zp + i, 0.0f); else store4<false>(zp + i, 0.0f);
                        }
                    }
                }
            }
        }
    }
}

}  // namespace mpk
