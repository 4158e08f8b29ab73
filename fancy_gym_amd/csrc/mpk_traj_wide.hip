// k_traj_wide: shared phase, more than 16 contraction columns or more than 16 DoF -- a k-chunked GEMM on the matrix cores
#include "mpk_tile.h"

namespace mpk {

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
// promp / dmp keep the whole horizon in one row-tile block of at most 32 tiles
bool traj_wide_fits(const DevCfg& c) { return c.mp_type == MPK_MP_PRODMP || (c.T + 15) / 16 <= 32; }

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
    if (!traj_wide_fits(c)) return MPK_ENOTIMPL;                           // whole horizon in one row-tile block
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
    while (KC > 8 && (stage_bytes(KC) > kLdsHalf || nout * KC / 4 * spans > 8)) KC >>= 1;
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
        if (lds > kLdsDefault) {
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

}  // namespace mpk
