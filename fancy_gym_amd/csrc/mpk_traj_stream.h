// k_traj_stream: the episode-major shared-phase trajectory kernel (one group per wave)
#pragma once
#include "mpk_tile.h"

namespace mpk {

// ---- episode-major ---------------------------------------------------------------------------------------------
// all row tiles of one episode group, in order (shared by the two input-staging variants of k_traj_stream)
template <int MP, int CT, int KM>
__device__ __forceinline__ void stream_group(const TrajArgs& a, const LaneMap<KM>& L, const float* ap,
                                             const float* sAux, const double* sg, float* sSt, int lane, int b0,
                                             const float (&xb)[KM], double cp, double cv, float ey, float ez,
                                             float eg, bool eul, double& qs, double& qds, int nst, bool serial) {
    constexpr bool ACT = CT >= 0;
    constexpr bool CLOSED = CT >= 3;
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : (MP == MPK_MP_PROMP ? 3 : 1);
    constexpr int NST = 2 + (ACT ? 1 : 0);
    const DevCfg& c = a.c;
    const int KP = 4 * KM, TS = a.TS, D = c.D, T = c.T;
    const int NRT = (T + 15) >> 4;
    const unsigned shw = ep_shift(a, b0 + L.bl);          // this column's episode image offset (same for every tile)
    const unsigned wofs = L.wofs + shw;
    const int o0 = L.bl * a.pitch + L.d + (int)shw;       // (row 0, this column) for the serial recurrences
    // step whose desired state is gathered for the next plan's boundary condition (k_condition_gather's clamp); -1 = off
    const int tcond = (CLOSED && a.rp.cond_pos) ? min(max(nst - 1, 0), T - 1) : -1;
    for (int rt = 0; rt < NRT; ++rt) {
        const int rows = min(16, T - rt * 16);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < KM; ++m) {
            const float* am = ap + (4 * m) * TS + rt * 16;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[0], xb[m], acc0, 0, 0, 0);
            if (NOUT > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[KP * TS], xb[m], acc1, 0, 0, 0);
            if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(am[2 * KP * TS], xb[m], acc2, 0, 0, 0);
        }
        if (MP != MPK_MP_DMP) {
            float dtd[4] = {1.f, 1.f, 1.f, 1.f};
            if (MP == MPK_MP_PROMP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dtd[r] = sAux[rt * 16 + 4 * L.q + r];
            }
            if (L.dvalid) {
                Gains gn{0.0, 0.0, 0.0, 0.0};
                if (CT >= 0 && CT < 3) gn = parked_gains(sg);
                tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, cp, cv, gn, sSt, wofs, D);
            }
            if (CLOSED) {
                // the step loop of black_box_wrapper.py:175-203 on the reference's torque double integrator
                // (base_reacher_torque.py:25-26), serial in t on the lanes (q == 0); float64, no FMA
                __builtin_amdgcn_wave_barrier();
                const bool full_tile = tile_fully_executed(serial, nst, rt * 16);
                // row tiles past the executed steps (and past the gathered step) have nothing serial to do: a replanning
                // plan that executes 25 of its 100 steps runs the recurrence on 2 of 7 tiles
                if (serial && rt * 16 < max(nst, tcond + 1)) {
                    // canonical once: fmin / fmax otherwise quiet their bound operands again at every step
                    const double pgd = sg[0], dgd = sg[16], lod = __builtin_canonicalize(sg[32]),
                                 hid = __builtin_canonicalize(sg[48]), dtp = a.plant_dt;
                    if (tcond >= rt * 16 && tcond < rt * 16 + 16) {   // condition_on_desired: the desired state at the
                        const size_t si = (size_t)(b0 + L.bl) * D + L.d;     // last executed step
                        a.rp.cond_pos[si] = sSt[o0 + (tcond - rt * 16) * D];
                        a.rp.cond_vel[si] = sSt[kStageStride + o0 + (tcond - rt * 16) * D];
                    }
                    if (full_tile)
                        pd_tile_steps<CT - 3, false>(sSt + o0, sSt + kStageStride + o0, sSt + 2 * kStageStride + o0, D,
                                                     rt * 16, nst, pgd, dgd, lod, hid, dtp, qs, qds);
                    else
                        pd_tile_steps<CT - 3, true>(sSt + o0, sSt + kStageStride + o0, sSt + 2 * kStageStride + o0, D,
                                                    rt * 16, nst, pgd, dgd, lod, hid, dtp, qs, qds);
                }
            }
        } else {
            // DMP: forcing tile -> LDS, then explicit Euler in scaled time on lanes (q == 0), serial in t;
            // one rounding per op (no FMA), first sample = initial condition
            float* sF = sSt + 2 * kStageStride;
            if (L.dvalid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sF[wofs + r * D] = acc0[r];
            }
            __builtin_amdgcn_wave_barrier();
            if (eul)
                dmp_tile_steps(sF + o0, sSt + o0, sSt + kStageStride + o0, sAux + rt * 16, D, rt * 16, T, c.dmp_alpha,
                               c.dmp_beta, eg, make_tau_div(c.tau), ey, ez);
            // (vel = z / tau is written by the recurrence lanes themselves)
        }
        __builtin_amdgcn_wave_barrier();
        if (a.wt) tile_store<NST, KM, true>(a, L, sSt, lane, b0, rt, rows);      // cache-resident outputs (wave-uniform)
        else tile_store<NST, KM, false>(a, L, sSt, lane, b0, rt, rows);
        __builtin_amdgcn_wave_barrier();
    }
}

constexpr int kChunkGroups = 4;   // episode groups whose inputs one bulk read brings in (BULK variant)

// BULK = false: the raw inputs of the next episode group are gathered per lane straight from HBM (as tile-major).
// BULK = true : a wave owns CHUNKS of kChunkGroups consecutive groups; the chunk's params / init_pos / init_vel
//               (/ c_pos / c_vel) blocks are contiguous in HBM and are read with a handful of coalesced float4 loads
//               one chunk ahead, parked in registers, and committed to a double-buffered wave-private LDS image from
//               which the B fragments are gathered.  Rationale: at HBM-streaming batch sizes the
//               scattered 336-byte parameter reads interleaved with the write stream cost ~35 % of the bandwidth.
template <int MP, int CT, int KM, bool BULK>
__global__ void __launch_bounds__(256) k_traj_stream(const TrajArgs a, const ActArgs act) {
    __shared__ __attribute__((aligned(16))) float smem[4 * kStageFloats];
    extern __shared__ __attribute__((aligned(16))) float sTab[];   // [NOUT][KP][TS] rows + [TS] aux (+ chunk images)
    constexpr bool ACT = CT >= 0 && CT < 3;   // open loop: frozen state (c_pos, c_vel) is an input
    constexpr bool CLOSED = CT >= 3;          // closed loop: plant state (q, qd) is read, integrated and written back

    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : (MP == MPK_MP_PROMP ? 3 : 1);
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = 4 * KM, TS = a.TS, D = c.D, B = a.B, P = c.P;
    float* sSt = smem + wave * kStageFloats;
    float* sA = sTab;
    float* sAux = sTab + NOUT * KP * TS;
    stage_tables(a.A, a.aux, sA, sAux, (NOUT * KP * TS) >> 2, TS >> 2, threadIdx.x);   // once per workgroup
    __syncthreads();
    const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
    // XCD-contiguous virtual block id (workgroup b runs on XCD b % 8): neighbouring episode groups share an L2
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3) : blockIdx.x;
    const int wstride = gridDim.x * 4;
    const int w0 = vb * 4 + wave;
    const float* ap = sA + L.q * TS + L.col;
    const double* sg = reinterpret_cast<const double*>(sSt + 3 * kStageStride) + (L.dvalid ? L.d : 0);

    if (!BULK) {
        int g = w0;
        if (g >= a.G) return;
        if (CT >= 0) park_gains(act, lane, L.d, sSt);
        float xb[KM];
        GroupIn<KM> cur = load_group<MP, ACT, KM>(a, L, g);
        finish_group<KM>(L, cur, xb);
        double cp = cur.cp, cv = cur.cv;
        while (g < a.G) {
            const int b0 = g * L.NTW;
            const int gn = g + wstride;
            const GroupIn<KM> nxt = load_group<MP, ACT, KM>(a, L, gn < a.G ? gn : g);
            float ey = 0.f, ez = 0.f, eg = 0.f;
            const bool eul = MP == MPK_MP_DMP && L.dvalid && L.q == 0 && b0 + L.bl < B;
            if (MP == MPK_MP_DMP) {
                if (eul) {
                    const int b = b0 + L.bl;
                    ey = a.init_pos[(size_t)b * D + L.d];
                    ez = a.init_vel[(size_t)b * D + L.d] * c.tau;
                    eg = a.params[(size_t)b * P + c.off + L.d * c.Kloc + c.nb] * c.gs;
                }
            }
            const bool serial = CLOSED && L.dvalid && L.q == 0 && b0 + L.bl < B;
            double qs = 0.0, qds = 0.0;
            int nst = c.T;
            if (CLOSED) {
                if (serial) {
                    const size_t si = (size_t)(b0 + L.bl) * D + L.d;
                    qs = a.q_state[si]; qds = a.qd_state[si];
                    if (a.rp.traj_steps) nst = replan_rule(a.rp, b0 + L.bl, c.T, L.d == 0);
                    else if (a.n_steps) nst = min(a.n_steps[b0 + L.bl], c.T);
                }
            }
            stream_group<MP, CT, KM>(a, L, ap, sAux, sg, sSt, lane, b0, xb, cp, cv, ey, ez, eg, eul, qs, qds, nst, serial);
            if (CLOSED) {
                if (serial) {
                    const size_t si = (size_t)(b0 + L.bl) * D + L.d;
                    a.q_state[si] = qs; a.qd_state[si] = qds;
                }
            }
            finish_group<KM>(L, nxt, xb);
            cp = nxt.cp; cv = nxt.cv;
            g = gn;
        }
    } else {
        constexpr int CH = kChunkGroups;
        const int NTW = L.NTW, EPC = CH * NTW;                 // episodes per chunk
        const int NCH = (B + EPC - 1) / EPC;
        int ch = w0;
        if (ch >= NCH) return;
        if (CT >= 0) park_gains(act, lane, L.d, sSt);
        // chunk image (floats): [params EPC*P | init_pos EPC*D | init_vel EPC*D | c_pos 2*EPC*D | c_vel 2*EPC*D]
        const int offIP = EPC * P, offIV = offIP + EPC * D, offCP = offIV + EPC * D, offCV = offCP + 2 * EPC * D;
        const int img = offCV + 2 * EPC * D;
        float* sImg = sAux + TS + wave * (2 * img);
        const int nP4 = (EPC * P) >> 2, nI4 = (EPC * D) >> 2, nC4 = (EPC * D) >> 1;    // float4 per block
        f32x4 rp0 = {0, 0, 0, 0}, rp1 = rp0, rip = rp0, riv = rp0, rcp = rp0, rcv = rp0;
        auto issue = [&](int chn) {       // coalesced float4 reads of a FULL chunk (ragged chunks are read below)
            const size_t e0 = (size_t)chn * EPC;
            const f32x4* p4 = reinterpret_cast<const f32x4*>(a.params + e0 * P);
            const f32x4* i4 = reinterpret_cast<const f32x4*>(a.init_pos + e0 * D);
            const f32x4* v4 = reinterpret_cast<const f32x4*>(a.init_vel + e0 * D);
            if (lane < nP4) rp0 = p4[lane];
            if (lane + 64 < nP4) rp1 = p4[lane + 64];
            if (lane < nI4) { rip = i4[lane]; riv = v4[lane]; }
            if (ACT) {
                if (lane < nC4) {
                    rcp = reinterpret_cast<const f32x4*>(a.c_pos + e0 * D)[lane];
                    rcv = reinterpret_cast<const f32x4*>(a.c_vel + e0 * D)[lane];
                }
            }
        };
        auto commit = [&](float* buf) {
            f32x4* b4 = reinterpret_cast<f32x4*>(buf);
            if (lane < nP4) b4[lane] = rp0;
            if (lane + 64 < nP4) b4[lane + 64] = rp1;
            if (lane < nI4) { b4[(offIP >> 2) + lane] = rip; b4[(offIV >> 2) + lane] = riv; }
            if (ACT) {
                if (lane < nC4) { b4[(offCP >> 2) + lane] = rcp; b4[(offCV >> 2) + lane] = rcv; }
            }
        };
        auto read_ragged = [&](int chn, float* buf) {   // last, incomplete chunk: element-wise, bounds-checked
            const size_t e0 = (size_t)chn * EPC;
            const int ne = B - (int)e0;
            for (int e = lane; e < ne * P; e += 64) buf[e] = a.params[e0 * P + e];
            for (int e = lane; e < ne * D; e += 64) {
                buf[offIP + e] = a.init_pos[e0 * D + e];
                buf[offIV + e] = a.init_vel[e0 * D + e];
                if (ACT) {
                    reinterpret_cast<double*>(buf + offCP)[e] = a.c_pos[e0 * D + e];
                    reinterpret_cast<double*>(buf + offCV)[e] = a.c_vel[e0 * D + e];
                }
            }
        };
        auto full = [&](int chn) { return (chn + 1) * EPC <= B; };
        int cur = 0;
        if (full(ch)) { issue(ch); commit(sImg); } else read_ragged(ch, sImg);
        __builtin_amdgcn_wave_barrier();
        while (ch < NCH) {
            const int chn = ch + wstride;
            const bool have_next = chn < NCH, next_full = have_next && full(chn);
            if (next_full) issue(chn);                      // in flight under this chunk's CH groups
            const float* buf = sImg + cur * img;
            for (int j = 0; j < CH; ++j) {
                const int g = ch * CH + j;
                if (g >= a.G) break;
                const int b0 = g * NTW;
                const float* pj = buf + j * NTW * P;
                const unsigned io = (unsigned)(j * NTW * D) + L.ioff;
                float xb[KM];
                const float ip = buf[offIP + io], iv = buf[offIV + io];
#pragma unroll
                for (int m = 0; m < KM; ++m) {
                    const float raw = pj[L.poff[m]];
                    xb[m] = L.isp[m] ? raw : (L.isip[m] ? ip : (L.isiv[m] ? iv : L.cst[m]));
                }
                double cp = 0.0, cv = 0.0;
                if (ACT) {
                    cp = reinterpret_cast<const double*>(buf + offCP)[io];
                    cv = reinterpret_cast<const double*>(buf + offCV)[io];
                }
                float ey = 0.f, ez = 0.f, eg = 0.f;
                const bool eul = MP == MPK_MP_DMP && L.dvalid && L.q == 0 && b0 + L.bl < B;
                if (MP == MPK_MP_DMP) {
                    if (eul) {
                        ey = ip;
                        ez = iv * c.tau;
                        eg = pj[L.bl * P + c.off + L.d * c.Kloc + c.nb] * c.gs;
                    }
                }
                const bool serial = CLOSED && L.dvalid && L.q == 0 && b0 + L.bl < B;
                double qs = 0.0, qds = 0.0;
                int nst = c.T;
                if (CLOSED) {
                    if (serial) {
                        const size_t si = (size_t)(b0 + L.bl) * D + L.d;
                        qs = a.q_state[si]; qds = a.qd_state[si];
                        if (a.rp.traj_steps) nst = replan_rule(a.rp, b0 + L.bl, c.T, L.d == 0);
                        else if (a.n_steps) nst = min(a.n_steps[b0 + L.bl], c.T);
                    }
                }
                stream_group<MP, CT, KM>(a, L, ap, sAux, sg, sSt, lane, b0, xb, cp, cv, ey, ez, eg, eul, qs, qds, nst,
                                         serial);
                if (CLOSED) {
                    if (serial) {
                        const size_t si = (size_t)(b0 + L.bl) * D + L.d;
                        a.q_state[si] = qs; a.qd_state[si] = qds;
                    }
                }
            }
            if (have_next) {
                float* nb = sImg + (cur ^ 1) * img;
                if (next_full) commit(nb); else read_ragged(chn, nb);
                __builtin_amdgcn_wave_barrier();
            }
            cur ^= 1;
            ch = chn;
        }
    }
}

}  // namespace mpk
