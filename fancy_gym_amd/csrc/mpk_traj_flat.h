// k_traj_flat: episode-major with whole-trajectory LDS images (HBM-streaming open-loop launches)
#pragma once
#include "mpk_tile.h"
#include "mpk_traj_stream.h"

namespace mpk {

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

}  // namespace mpk
