// The 16 x 16 tile machinery shared by every shared-phase trajectory kernel family (and the rollout kernels): kernel
// arguments, lane maps, B-fragment gathers, controller / Euler step chains, the C-tile epilogue and the coalesced tile stores.
#pragma once
#include "mpk_dev.h"

namespace mpk {

// ------------------------------------------------------------------------------------------------------------
// The [T x K] . [K x D] contraction on the matrix cores (v_mfma_f32_16x16x4_f32) + fused epilogues.
//
// Tile = 16 time steps x 16 (episode, DoF) columns, K = 4*KM <= 16.  A fragments = basis rows (with weights_scale /
// goal_scale / tau / relative goal folded in at build time); B fragments = RAW parameters / boundary conditions
// gathered from HBM/L2 in fragment layout (wave-uniform base pointers + lane-constant 32-bit offsets, straight-line
// code, prefetched one episode group ahead).  The C tile is transposed through a wave-private LDS buffer so that
// every output array of a tile leaves as ONE coalesced float4 store instruction.  Wave-level indices live in SGPRs.
//
// Two work decompositions of the same tile code (tools/store_probe.hip, profiles/r01_store_patterns.md):
//   k_traj_tiles   tile-major: a wave owns ONE row tile (A fragments stay in registers) and walks episode groups.
//                  Maximum parallelism for small batches whose outputs stay cache resident.
//   k_traj_stream  episode-major: a wave owns an episode group and walks its row tiles in order, A fragments come
//                  from a per-workgroup LDS copy of the basis tables.  Every wave writes long contiguous runs,
//                  which is what the HBM write path needs at large batch (4.9 vs 3.0 TB/s for the same bytes).
//                  DMP always runs here (the Euler recurrence is serial in t).
// CT: fused controller: -1 none; MPK_CTRL_* (0..2) = open loop against a frozen state (c_pos, c_vel);
//     3 + MPK_CTRL_* = CLOSED loop with the double-integrator plant integrated in the kernel (episode-major only).
// ------------------------------------------------------------------------------------------------------------
struct TrajArgs {
    DevCfg c;
    const float* A;
    const float* aux;
    int TS;
    const float* params;
    const float* init_pos;
    const float* init_vel;
    float* pos;
    float* vel;
    float* actions;
    const double* c_pos;
    const double* c_vel;
    int B, sh, G, vec_ok;
    // LDS staging geometry: `pitch` floats per episode (16*D, +4 when the image is shifted), `cps` float4 chunks per
    // episode segment, inv_cps = 65536 / cps + 1.  shifted: T*D is not a multiple of 4, so episode b starts
    // ((b & 3) * (T*D & 3)) & 3 floats past a 16-byte boundary; its tile image is staged with the same offset so
    // that 16-byte-aligned LDS chunks map onto 16-byte-aligned HBM chunks (partial chunks at both ends go scalar)
    int pitch, cps, shifted, td3;
    unsigned inv_cps;
    // tile-major work assignment: wave w owns row tile w % NRT of groups w / NRT, + gstride, ...; nrt_magic =
    // 2^32 / NRT + 1 (NRT > 1) turns the division into a multiply-high
    unsigned nrt_magic;
    int gstride;
    // episode-major kernels: write-through (sc1) stores while the outputs are cache resident (a serial-recurrence launch
    // of a few thousand episodes: closed-loop step at B = 4096 22.7 -> 18.7 us); plain stores once they stream to HBM
    // (write-through costs 25 % there).  The tile-major kernel has the policy as a template parameter.
    int wt;
    int flat_img;          // k_traj_flat / k_traj_ring: floats per whole-trajectory array image (NTW * T * D); 0 = another kernel runs
    // k_traj_ring (ring_np > 0): producer waves, store-engine waves, episode groups per batch, batch buffers in the LDS ring
    int ring_np, ring_ns, ring_m, ring_nbuf;
    int ring_nc;           // k_traj_ring, closed loop: consumer waves (the recurrences of a batch, one group per lane quarter)
    int ring_aw;           // ... 1: one action-writer wave per consumer (the consumers never issue a global store)
    unsigned* ring_ctr;    // k_traj_ring: device-wide ticket counter (zeroed before the launch); nullptr = static batch ranges
    int ring_tb;           // batches per ticket
    int ring_parts;        // waves that share one group's row tiles (long horizons: the image of ONE group fills a batch buffer)
    int burst;             // k_traj_burst: short-lived workgroups, one batch of ring_m groups each, ring_np waves per group
    int lean;              // k_traj_pipe: the register-lean instantiation (more than two work units per CU)
    int inorder;           // k_traj_quad: 1 = one unit per wave, workgroup b takes units 4 b .. 4 b + 3 (short-lived workgroups in address order)
    int wpb;               // tile-major kernel: waves per workgroup (4; "tiles_wpb" 1 / 2 for A/B runs)
    int ring_dbg;          // ablations (mpk_set_option "ring_dbg"): 1 producers publish without contracting, 2 the engine skips its stores
    int* fault;            // k_traj_ring: the handle's fault word (host memory, mapped): a role that gives up waiting ORs its code in
    unsigned ser_blocks;   // k_traj_split: workgroups [0, ser_blocks) run the serial role
    // closed-loop rollout fused into the episode-major kernel (CT >= 3)
    double* q_state;       // [B, D] plant position, in/out
    double* qd_state;      // [B, D] plant velocity, in/out
    const int32_t* n_steps;  // [B] executed steps of this plan (NULL = T)
    double plant_dt;
    ReplanDev rp;            // closed loop only: integer replanning state advanced in the kernel (replaces n_steps)
    // validity gate of the lane-quarter closed-loop kernels (k_traj_quad / duo / mono, k_episode_return; round 6): gate_valid != nullptr
    // switches it on -- a first pass over the unit's row tiles judges every plan BEFORE its recurrence starts (gate_pass below)
    uint8_t* gate_valid;     // [B] out
    double* gate_penalty;    // [B] out, optional
    const float* gate_raw;   // [B, P] the action as passed: tau = [b, 0], delay = [b, 1] (gate_check_td)
    int gate_check_td;
    double gate_tb[2], gate_db[2];
};

struct ActArgs {
    double pg[kMaxD], dg[kMaxD], lo[kMaxD], hi[kMaxD];
    // validity gate: joint limits as the reference holds them, and as exact fp32 thresholds (smallest fp32 >= low, largest fp32 <= high:
    // an fp32 position lies in [low, high] exactly when it lies in [glo32, ghi32])
    double glo[kMaxD], ghi[kMaxD];
    float glo32[kMaxD], ghi32[kMaxD];
};

constexpr int kStageStride = 256;   // floats between output arrays in the wave's LDS staging area (>= NTW*16*D)
constexpr int kStageFloats = 4 * kStageStride;   // pos | vel | actions or DMP forcing | controller constants

// ---- k_episode_return (mpk_episode.hip): third kernel argument and LDS geometry --------------------------------------------------
struct EpArgs {
    double* ret;               // [B] out: aggregated reward of this plan's executed steps (0 without a reward)
    const double* goal;        // [B, 2] SimpleReacher goal (reward = 1)
    const int32_t* step0;      // [B] environment step counter at the plan's first step when there is no replanning state (else: traj_steps)
    int32_t* seg_out;          // [B] out, optional: executed steps (with a replanning state seg_len carries them already)
    int steps_before_reward;
    int agg;                   // 0 sum, 1 mean, 2 last
    int km;                    // contraction columns / 4 (the kernel is compiled for up to 16 columns; MFMAs beyond km are skipped)
    int wpb;                   // waves per workgroup (4 or 8): host side only
};

constexpr int kEpImg = 4 * kStageStride + 8;     // floats per group: desired pos | vel (reused as the float64 position image) | float64 actions
constexpr int kEpImgPlain = 2 * kStageStride + 8;   // ... without a reward: desired pos | vel
constexpr int kEpSlots = 8;                      // episode slots per wave with a reward (NQ * NTW <= 8)
constexpr int kEpMaxPass = 2;                    // reward passes per tile: up to 8 episodes per unit (the launcher keeps NQ * NTW <= 8 with a reward)
constexpr int kEpSlotInts = 8;                   // per episode slot: executed steps, step offset, episode (or -1), pad, goal x, goal y (float64)


enum : int { XK_ZERO = 0, XK_PARAM = 1, XK_IPOS = 2, XK_IVEL = 3, XK_ONE = 4 };

// which raw input feeds element k of a DoF's extended parameter column, and its offset inside the DoF's local block
template <int MP>
__device__ __forceinline__ int x_kind(const DevCfg& c, int k, int* loc) {
    // select form (no early returns): this runs in the latency-critical prologue of every trajectory kernel
    const int nb = c.nb;
    if (MP == MPK_MP_PRODMP) {
        const bool isw = k < nb, isg = k == nb;
        // the offset is used for an UNCONDITIONAL load (the kind decides afterwards whether the value counts), so it must
        // stay inside the DoF's local block whatever is disabled: with disable_goal the block has nb entries (no goal at
        // [nb]), with disable_weights one (the goal at [0]).  Round 3's fuzz soak found the old `isw ? k : ...`: the last
        // DoF of the last episode read one float (disable_goal) or up to nb - 1 floats (disable_weights) past the end of
        // `params` -- a memory fault once every ~10^4 random configurations, when the buffer ends on a page boundary.
        *loc = (isw && !c.disable_weights) ? k : ((isg && !c.disable_goal && !c.disable_weights) ? nb : 0);
        const int kw = c.disable_weights ? XK_ZERO : XK_PARAM, kg = c.disable_goal ? XK_ZERO : XK_PARAM;
        const int klast = (k == nb + 3 && c.goal_off_on) ? XK_ONE : XK_ZERO;
        return isw ? kw : (isg ? kg : (k == nb + 1 ? XK_IPOS : (k == nb + 2 ? XK_IVEL : klast)));
    } else if (MP == MPK_MP_PROMP) {
        const bool isw = k < nb;
        *loc = isw ? k : 0;
        return isw ? XK_PARAM : ((k == nb && c.KT > nb) ? XK_IPOS : XK_ZERO);
    } else {
        const bool isw = k < nb;
        *loc = isw ? k : 0;
        return isw ? XK_PARAM : XK_ZERO;
    }
}

// lane-constant description of a lane's role in the 16x16 tile machinery
template <int KM>
struct LaneMap {
    int col, q, bl, d, dsafe, NTW;
    bool dvalid;
    bool isp[KM], isip[KM], isiv[KM];
    float cst[KM];       // what an element that is no input carries: 0, or 1 for the goal-offset column
    unsigned poff[KM];   // element offset of B-fragment element m inside the group's params block
    unsigned ioff;       // element offset inside the group's init_pos / init_vel / c_pos / c_vel block
    unsigned wofs;       // LDS transpose: write offset of (row 4q, this column)
    int sseg, w4;        // episode-in-group and float offset of the float4 this lane stores
    unsigned rofs, gofs; // LDS read offset / global offset (relative to the tile base) of that float4
};

template <int MP, int KM>
__device__ __forceinline__ LaneMap<KM> make_lane_map(const TrajArgs& a, int lane) {
    const DevCfg& c = a.c;
    LaneMap<KM> L;
    const int D = c.D, DP = 1 << a.sh;
    L.NTW = 16 >> a.sh;
    L.col = lane & 15; L.q = lane >> 4;
    L.bl = L.col >> a.sh; L.d = L.col & (DP - 1);
    L.dvalid = L.d < D;
    L.dsafe = L.dvalid ? L.d : D - 1;
#pragma unroll
    for (int m = 0; m < KM; ++m) {
        int loc;
        const int kind = x_kind<MP>(c, 4 * m + L.q, &loc);
        L.isp[m] = L.dvalid && kind == XK_PARAM;
        L.isip[m] = L.dvalid && kind == XK_IPOS;
        L.isiv[m] = L.dvalid && kind == XK_IVEL;
        L.cst[m] = L.dvalid && kind == XK_ONE ? 1.0f : 0.0f;
        L.poff[m] = (unsigned)(L.bl * c.P + c.off + L.dsafe * c.Kloc + loc);
    }
    L.ioff = (unsigned)(L.bl * D + L.dsafe);
    L.wofs = (unsigned)(L.bl * a.pitch + 4 * L.q * D + L.d);
    L.sseg = (int)(((unsigned)lane * a.inv_cps) >> 16);
    L.w4 = (lane - L.sseg * a.cps) * 4;
    L.rofs = (unsigned)(L.sseg * a.pitch + L.w4);
    L.gofs = (unsigned)(L.sseg * c.T * D + L.w4);
    return L;
}

// floats by which episode b's trajectories start past a 16-byte boundary (0 unless the image is shifted)
__device__ __forceinline__ unsigned ep_shift(const TrajArgs& a, int b) {
    return a.shifted ? (((unsigned)b & 3u) * (unsigned)a.td3) & 3u : 0u;
}

// raw inputs of one episode group for this lane (plain loads, no control flow)
template <int KM>
struct GroupIn {
    float raw[KM];
    float ip, iv;
    double cp, cv;
};

template <class T>
__device__ __forceinline__ T ld_off(const T* base, unsigned byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

template <int MP, bool ACT, int KM>
__device__ __forceinline__ GroupIn<KM> load_group(const TrajArgs& a, const LaneMap<KM>& L, int g) {
    const DevCfg& c = a.c;
    GroupIn<KM> in;
    // the last group may be ragged: clamp its missing episodes onto the group's first one (computed, never stored)
    const int b0 = g * L.NTW;
    const bool bv = b0 + L.bl < a.B;
    // wave-uniform block bases + 32-bit per-lane BYTE offsets: the loads take the (scalar base, vector offset) form
    // instead of a 64-bit address addition per load on the vector ALU
    const float* pb = a.params + (size_t)b0 * c.P;
    const unsigned io = bv ? L.ioff : (unsigned)L.dsafe;
    const unsigned pclamp = bv ? 0u : (unsigned)(L.bl * c.P);
#pragma unroll
    for (int m = 0; m < KM; ++m) in.raw[m] = ld_off(pb, 4u * (L.poff[m] - pclamp));
    in.ip = MP != MPK_MP_DMP ? ld_off(a.init_pos + (size_t)b0 * c.D, 4u * io) : 0.0f;
    in.iv = MP == MPK_MP_PRODMP ? ld_off(a.init_vel + (size_t)b0 * c.D, 4u * io) : 0.0f;
    in.cp = 0.0; in.cv = 0.0;
    if (ACT) { in.cp = ld_off(a.c_pos + (size_t)b0 * c.D, 8u * io); in.cv = ld_off(a.c_vel + (size_t)b0 * c.D, 8u * io); }
    return in;
}

template <int KM>
__device__ __forceinline__ void finish_group(const LaneMap<KM>& L, const GroupIn<KM>& in, float (&xb)[KM]) {
#pragma unroll
    for (int m = 0; m < KM; ++m) xb[m] = L.isp[m] ? in.raw[m] : (L.isip[m] ? in.ip : (L.isiv[m] ? in.iv : L.cst[m]));
}

// park the controller constants of every DoF in the wave's 4th staging slot (static kernarg indices: no spill)
__device__ __forceinline__ void park_gains(const ActArgs& act, int lane, int d, float* sSt) {
    double pgd = 0.0, dgd = 0.0, lod = 0.0, hid = 0.0;
#pragma unroll
    for (int dd = 0; dd < kMaxD; ++dd)
        if (dd == d) { pgd = act.pg[dd]; dgd = act.dg[dd]; lod = act.lo[dd]; hid = act.hi[dd]; }
    if (lane < 16) {
        double* sg = reinterpret_cast<double*>(sSt + 3 * kStageStride);
        sg[lane] = pgd; sg[16 + lane] = dgd; sg[32 + lane] = lod; sg[48 + lane] = hid;
    }
    __builtin_amdgcn_wave_barrier();
}

// controller constants of one lane's DoF
struct Gains { double pg, dg, lo, hi; };

__device__ __forceinline__ Gains parked_gains(const double* sg) { return Gains{sg[0], sg[16], sg[32], sg[48]}; }

// The same constants straight from the kernel-argument segment with per-lane (vector) loads: the segment is ordinary
// device memory, so a lane-dependent index costs four 8-byte loads issued next to the kernel's first input loads,
// where selecting among scalar kernarg registers costs eight dependent s_load round trips and 16 exec-masked moves
// before any input load is issued (the tile-major kernel's whole life is ~8 us: its prologue is not free).
// `act` is the second kernel argument of every trajectory kernel.
constexpr size_t kActArgsOffset = (sizeof(TrajArgs) + alignof(ActArgs) - 1) / alignof(ActArgs) * alignof(ActArgs);
__device__ __forceinline__ Gains kernarg_gains(int d) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + kActArgsOffset;
    typedef const __attribute__((address_space(4))) double* dptr;
    Gains gn;
    gn.pg = ((dptr)(base + offsetof(ActArgs, pg)))[d];
    gn.dg = ((dptr)(base + offsetof(ActArgs, dg)))[d];
    gn.lo = ((dptr)(base + offsetof(ActArgs, lo)))[d];
    gn.hi = ((dptr)(base + offsetof(ActArgs, hi)))[d];
    return gn;
}

struct GateLim { double lo, hi; float lo32, hi32; };
__device__ __forceinline__ GateLim kernarg_gate(int d) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + kActArgsOffset;
    typedef const __attribute__((address_space(4))) double* dptr;
    typedef const __attribute__((address_space(4))) float* fptr;
    GateLim g;
    g.lo = ((dptr)(base + offsetof(ActArgs, glo)))[d];
    g.hi = ((dptr)(base + offsetof(ActArgs, ghi)))[d];
    g.lo32 = ((fptr)(base + offsetof(ActArgs, glo32)))[d];
    g.hi32 = ((fptr)(base + offsetof(ActArgs, ghi32)))[d];
    return g;
}

// Validity gate of the lane-quarter closed-loop kernels (preprocessing_and_validity_callback between plan and rollout,
// black_box_wrapper.py:155-172; TableTennisEnv.check_traj_validity / _get_traj_invalid_penalty, table_tennis_env.py:282-309): ONE extra
// pass over the row tiles of the unit's NQ groups BEFORE the recurrences start -- position C tiles only (KM MFMAs per group and tile,
// nothing stored), every position against its joint limits (exact fp32 thresholds, running max / min) -- so the step loop that follows
// simply runs with nst = 0 for an invalid plan: actions 0, plant state untouched, condition = row 0, done = 1 (replan_write).  Only a
// unit that holds a violation repeats the pass in float64 for the penalty's excess sums (reduced over the rows of a lane, the four
// lane quarters, the DoF lanes of an episode: a fixed order).  Returns, to the SERIAL lane of (group L.q, episode L.bl), whether that
// plan leaves the limits; over / under: its summed excess above / below (0 where nothing is violated).
//   ap: the lane's A-fragment base (sA + L.q * TS + L.col), position rows first (o = 0); km: MFMAs per tile actually needed
// the running state of a unit's scan: maximum / minimum of every position seen per group (this lane's rows and column), and which row
// tiles hold a violation anywhere in the wave
template <int NQ>
struct GateScan {
    float mx[NQ], mn[NQ];
    unsigned long long tiles;
    __device__ __forceinline__ void init(const GateLim& gl) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) { mx[j] = gl.lo32; mn[j] = gl.hi32; }
        tiles = 0ull;
    }
};
// one position C tile of group j, row tile rt, into the scan (v_max3_f32 / v_min3_f32; NaN positions pass, as in the reference's
// `np.any(pos > high)`, table_tennis_env.py:307); returns the lane's "this tile violates" for the caller's wave-level test
template <int NQ>
__device__ __forceinline__ bool gate_scan_tile(GateScan<NQ>& g, const int j, f32x4 acc, const int rt, const int lq, const bool dvalid,
                                               const GateLim& gl, const int T) {
    if ((rt + 1) * 16 > T) {                            // (wave-uniform: the horizon's last tile -- rows past it count as inside the limits)
        const int row0 = rt * 16 + 4 * lq;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = row0 + r < T ? acc[r] : gl.lo32;
    }
    const float hi4 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), __builtin_fmaxf(acc[2], acc[3]));
    const float lo4 = __builtin_fminf(__builtin_fminf(acc[0], acc[1]), __builtin_fminf(acc[2], acc[3]));
    g.mx[j] = __builtin_fmaxf(g.mx[j], hi4); g.mn[j] = __builtin_fminf(g.mn[j], lo4);
    return dvalid && (hi4 > gl.hi32 || lo4 < gl.lo32);
}
// the verdict for the SERIAL lane of (group L.q, episode L.bl) after every row tile went through the scan, and -- only in a unit that
// holds a violation -- the penalty's excess sums: the flagged row tiles again, float64, reduced over the rows of a lane, the four lane
// quarters and the DoF lanes of an episode (a fixed order)
template <int KM, int NQ>
__device__ __forceinline__ bool gate_verdict(const TrajArgs& a, const LaneMap<KM>& L, const GateScan<NQ>& g, const float* __restrict__ ap,
                                             const int TS, const int km, const float (&xz)[NQ][KM], const int g0, const GateLim& gl,
                                             double& over, double& under) {
    const int T = a.c.T, NRT = (T + 15) >> 4;
    // the lanes that hold episode bl's columns of a C tile: its DP columns, in all four lane quarters
    const int DP = 1 << a.sh;
    const unsigned long long em = (unsigned long long)(((1u << DP) - 1u) << (L.bl * DP)) * 0x0001000100010001ull;
    unsigned long long any = 0ull;
    bool mine = false;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const unsigned long long mj = __ballot(L.dvalid && (g.mx[j] > gl.hi32 || g.mn[j] < gl.lo32) && (g0 + j) * L.NTW + L.bl < a.B);
        any |= mj;
        if (L.q == j) mine = (mj & em) != 0ull;
    }
    over = 0.0; under = 0.0;
    if (any != 0ull) {                                  // (wave-uniform)
        double ov[NQ], un[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) { ov[j] = 0.0; un[j] = 0.0; }
        for (int rt = 0; rt < NRT; ++rt) {
            if (rt < 64 && !((g.tiles >> rt) & 1ull)) continue;
            float af[KM];
#pragma unroll
            for (int m = 0; m < KM; ++m) af[m] = m < km ? ap[(4 * m) * TS + rt * 16] : 0.0f;
            const int row0 = rt * 16 + 4 * L.q;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < KM; ++m)
                    if (m < km) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], xz[j][m], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double x = (double)acc[r];
                    const bool in = row0 + r < T && L.dvalid;
                    ov[j] += in ? fmax(x - gl.hi, 0.0) : 0.0;
                    un[j] += in ? fmax(gl.lo - x, 0.0) : 0.0;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            for (int sft = 16; sft <= 32; sft <<= 1) { ov[j] += __shfl_xor(ov[j], sft); un[j] += __shfl_xor(un[j], sft); }
            for (int sft = 1; sft < DP; sft <<= 1) { ov[j] += __shfl_xor(ov[j], sft); un[j] += __shfl_xor(un[j], sft); }
            if (L.q == j) { over = ov[j]; under = un[j]; }
        }
    }
    return mine;
}
// the separate first pass of the kernels that STORE (k_traj_quad / duo / mono: a speculative rollout would have to take its action
// stores back): position C tiles only, four row tiles per trip with their A fragments read together (the lone wave of a few thousand
// episodes pays latencies, not instructions)
template <int KM, int NQ>
__device__ __forceinline__ bool gate_pass(const TrajArgs& a, const LaneMap<KM>& L, const float* __restrict__ ap, const int TS, const int km,
                                          const float (&xb)[NQ][KM], const int g0, const GateLim& gl, double& over, double& under) {
    const int T = a.c.T, NRT = (T + 15) >> 4;
    // (a group past the launch's last one contracts zeros: no test per group inside the loop)
    float xz[NQ][KM];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int m = 0; m < KM; ++m) xz[j][m] = g0 + j < a.G ? xb[j][m] : 0.0f;
    GateScan<NQ> g;
    g.init(gl);
    constexpr int UN = 4;
    for (int rt0 = 0; rt0 < NRT; rt0 += UN) {
        float af[UN][KM];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int m = 0; m < KM; ++m) af[u][m] = (m < km && rt0 + u < NRT) ? ap[(4 * m) * TS + (rt0 + u) * 16] : 0.0f;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            bool tv = false;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < KM; ++m)
                    if (m < km) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][m], xz[j][m], acc, 0, 0, 0);
                tv = gate_scan_tile<NQ>(g, j, acc, rt0 + u, L.q, L.dvalid, gl, T) || tv;
            }
            if (rt0 + u < 64 && rt0 + u < NRT && __any(tv) != 0) g.tiles |= 1ull << (rt0 + u);
        }
    }
    return gate_verdict<KM, NQ>(a, L, g, ap, TS, km, xz, g0, gl, over, under);
}

// The step loop of black_box_wrapper.py:175-203 on the reference's torque double integrator (base_reacher_torque.py:25-26)
// for the 16 steps of one row tile of ONE (episode, DoF) lane: float64, no FMA contraction -- numpy's promotion in
// pd_controller.py:21-29.  The desired states of the tile are pulled into registers first, then the chain runs as
// straight-line code WITHOUT control flow: a step past the executed ones (t >= nst) is computed and discarded by selects
// (its action is written as 0).  Measured on one wave (tools/dev/trace_kernel.py, tools/dev/rec_latency.hip,
// profiles/r02_closed_loop.md): with two exec-mask branches per step (t == tcond, t < nst) a step cost 260 cycles; this
// form costs 83 in isolation (57 for the bare chain of 11 float64 operations, the rest conversions and the LDS write).
// Feeding the chain float64 values from LDS (conversions done by all 64 lanes beforehand) measured the same 83 in
// isolation and SLOWER in the kernel (an extra LDS pass and barrier per tile: 18.3 vs 14.4 us), so it stays as it is.
// MASKED = false is the version for a tile every step of which is executed by every lane of the wave (the caller tests
// that wave-uniformly).  sP / sV / sA: the lane's (row 0, column) slots of the desired pos / vel / action images,
// `stride` floats per row.
// Round 4 (tools/probes/fp64_rate_probe.hip, asm of the consumer wave): a wave issues an independent instruction every 4 cycles
// and a dependent one every 5.9, so the bare step (9 dependent of 11 float64 operations) is 54 cycles and every OTHER instruction
// of the same wave adds 2 - 4; left alone the compiler sinks each step's LDS read next to its use (`s_waitcnt lgkmcnt(1)` in front
// of every step: ~40 cycles of LDS latency exposed per step).  PRE = 1: all 32 reads issued, ONE wait, all conversions, then the
// chain with nothing but its own operations, the action conversion and the LDS write in between.
// GATE (round 6, k_phase_fused): the desired positions the chain pulls into registers anyway are also tested against the joint limits
// [glo32, ghi32] (exact fp32 thresholds of the validity gate); *gate_bad |= any of the tile's `rows` positions outside -- no LDS read, no
// wait of its own; a wave that holds a violation adds the tile's float64 excess above / below [gate_lo, gate_hi] to gate_sum[0] / [1].
template <int CTRL, bool MASKED, bool INTEGRATE = true, int KEEP64 = 0, int PRE = 0, bool WRITE_A = true, bool GATE = false>
__device__ __forceinline__ void pd_tile_steps(const float* __restrict__ sP, const float* __restrict__ sV,
                                              float* __restrict__ sA, const int stride, const int t0, const int nst,
                                              const double pgd, const double dgd, const double lod, const double hid,
                                              const double dtp, double& qs, double& qds, double* __restrict__ q64 = nullptr,
                                              double* __restrict__ u64 = nullptr, const int rows = 16, const float glo32 = 0.0f,
                                              const float ghi32 = 0.0f, int* __restrict__ gate_bad = nullptr, const double gate_lo = 0.0,
                                              const double gate_hi = 0.0, double* __restrict__ gate_sum = nullptr) {
    // INTEGRATE = false: MPK_PLANT_STATIC (the state never changes).  KEEP64 = 1: the plant position after the step and the
    // clipped action also stay in LDS as float64, q64 / u64 = the lane's column of a [16 columns][16 steps] image (the reward pass of
    // the reacher rollout reads them);
    // KEEP64 = 2: the action only (a tile none of whose steps carries the reward's distance term: round 5)
    float pr[16], vr[16];
#pragma unroll
    for (int tl = 0; tl < 16; ++tl) { pr[tl] = sP[tl * stride]; vr[tl] = sV[tl * stride]; }
    if (GATE) {
        // running maximum / minimum of the tile's positions: two v_max3_f32 / v_min3_f32 per four values, compared once (rows past the
        // horizon replaced by a value inside the limits -- a select, no control flow).  NaN positions pass, as they do in the reference's
        // `np.any(pos > high)` (table_tennis_env.py:307)
        float mx = glo32, mn = ghi32;
#pragma unroll
        for (int tl = 0; tl < 16; tl += 2) {
            const float p0 = (!MASKED || tl < rows) ? pr[tl] : glo32, p1 = (!MASKED || tl + 1 < rows) ? pr[tl + 1] : glo32;
            mx = __builtin_fmaxf(mx, __builtin_fmaxf(p0, p1));
            mn = __builtin_fminf(mn, __builtin_fminf(p0, p1));
        }
        const bool bh = mx > ghi32, bl_ = mn < glo32;
        gate_bad[0] |= (int)(bh || bl_);
        // (wave-uniform, per side: a wave usually violates one of the two limits) the penalty's float64 excess sums of this tile, in time
        // order, from the registers
        if (__any(bh) != 0) {
            double ov = 0.0;
#pragma unroll
            for (int tl = 0; tl < 16; ++tl) {
                if (MASKED && tl >= rows) break;
                ov += fmax((double)pr[tl] - gate_hi, 0.0);
            }
            gate_sum[0] += ov;
        }
        if (__any(bl_) != 0) {
            double un = 0.0;
#pragma unroll
            for (int tl = 0; tl < 16; ++tl) {
                if (MASKED && tl >= rows) break;
                un += fmax(gate_lo - (double)pr[tl], 0.0);
            }
            gate_sum[1] += un;
        }
    }
    double dpr[16], dvr[16];
    if (PRE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int tl = 0; tl < 16; ++tl) {
            if (CTRL != MPK_CTRL_VELOCITY) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dpr[tl]) : "v"(pr[tl]));
            if (CTRL != MPK_CTRL_POSITION) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dvr[tl]) : "v"(vr[tl]));
        }
    }
    // MASKED tiles (the horizon's last, partial tile; the tile in which a plan's executed steps end): the steps no lane of the wave
    // executes are not computed -- their actions are 0, written by a short loop behind the chain, and only for the `rows` steps of
    // the tile that lie inside the horizon (nothing past it is ever stored).  nlive = steps at which some lane is still live: where
    // every lane executes the same number of steps (the rule, one compare says so) it is arithmetic, else a binary search over four
    // ballots (live is monotone in the step).  Round 4, second session: with T = 100 the seventh tile has four steps, and
    // computing-and-discarding the other twelve cost as much as a full tile (trace of k_traj_pipe: 2 000 - 2 600 cycles).
    int nlive = 16;
    if (MASKED) {
        const int n0 = __builtin_amdgcn_readfirstlane(nst);
        if (__all(nst == n0)) {
            nlive = min(max(n0 - t0, 0), 16);
        } else {
            int lo = 0, hi = 16;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (__any(t0 + mid < nst)) lo = mid + 1; else hi = mid;
            }
            nlive = lo;
        }
    }
#pragma unroll
    for (int tl = 0; tl < 16; ++tl) {
        if (MASKED && tl >= nlive) break;                 // (wave-uniform)
        const double dp = PRE ? dpr[tl] : (double)pr[tl], dv = PRE ? dvr[tl] : (double)vr[tl];
        double u;
        if (CTRL == MPK_CTRL_MOTOR) u = pgd * (dp - qs) + dgd * (dv - qds);
        else if (CTRL == MPK_CTRL_POSITION) u = dp;
        else u = dv;
        u = fmin(fmax(u, lod), hid);
        const double qds_n = INTEGRATE ? qds + dtp * u : qds;
        const double qs_n = INTEGRATE ? qs + dtp * qds_n : qs;
        // (WRITE_A = false: no float32 action image -- the episode-return kernel stores no actions)
        if (MASKED) {
            const bool live = t0 + tl < nst;
            qds = live ? qds_n : qds;
            qs = live ? qs_n : qs;
            u = live ? u : 0.0;
            if (WRITE_A) sA[tl * stride] = (float)u;
        } else {
            qds = qds_n; qs = qs_n;
            if (WRITE_A) sA[tl * stride] = (float)u;
        }
        // (column-major images: [column][step] -- the reward pass reads 16 consecutive steps of one column with 16 neighbouring
        // lanes; step-major, those reads were 128 bytes apart: one LDS bank pair for all of them.  The writes here are the ones 128
        // bytes apart now, 16 lanes of one instruction -- but nothing waits for a write)
        if (KEEP64 == 1) q64[tl] = qs;
        if (KEEP64) u64[tl] = u;
    }
    if (MASKED) {
#pragma unroll 1
        for (int tl = nlive; tl < rows; ++tl) {
            if (WRITE_A) sA[tl * stride] = 0.0f;
            if (KEEP64 == 1) q64[tl] = qs;
            if (KEEP64) u64[tl] = 0.0;
        }
    }
}

// clip(u, lo, hi) of the step loop as the two instructions it is: fmin / fmax make the compiler re-quiet a loop-invariant
// bound before every use (a v_max_f64 x, x per bound and step -- two of the ~14 float64 operations of a step).  lo / hi
// are finite controller bounds or +-inf, never NaN; u is quieted by the instructions themselves (IEEE mode).
__device__ __forceinline__ double clip_f64(double u, double lo, double hi) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(u), "v"(lo));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(r), "v"(hi));
    return r;
}

// all lanes of the wave: does every serial lane execute every step of the tile [t0, t0 + 16)?  (wave-uniform)
__device__ __forceinline__ bool tile_fully_executed(bool serial, int nst, int t0) {
    return __all(!serial || nst >= t0 + 16) != 0;
}

// DMP's explicit Euler recurrence for the 16 steps of one row tile of one (episode, DoF) lane, fp32, one
// rounding per op, branch-free like pd_tile_steps: a step at or past T - 1 leaves the state alone by select.
__device__ __forceinline__ void dmp_tile_steps(const float* __restrict__ sF, float* __restrict__ sP, float* __restrict__ sV,
                                               const float* __restrict__ ds16, const int stride, const int t0,
                                               const int T, const float alpha, const float beta, const float eg,
                                               const TauDiv& td, float& ey, float& ez) {
    float fr[16], dsr[16];
#pragma unroll
    for (int tl = 0; tl < 16; ++tl) fr[tl] = sF[tl * stride];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 x = reinterpret_cast<const float4*>(ds16)[j];
        dsr[4 * j] = x.x; dsr[4 * j + 1] = x.y; dsr[4 * j + 2] = x.z; dsr[4 * j + 3] = x.w;
    }
#pragma unroll
    for (int tl = 0; tl < 16; ++tl) {
        sP[tl * stride] = ey;
        sV[tl * stride] = div_tau(ez, td);               // vel = z / tau, off the dependent chain
        const float t1 = eg - ey;
        const float t2 = beta * t1;
        const float t3 = t2 - ez;
        const float t4 = alpha * t3;
        const float acc = t4 + fr[tl];
        const float ez_n = ez + dsr[tl] * acc;
        const float ey_n = ey + dsr[tl] * ez_n;
        const bool live = t0 + tl < T - 1;
        ez = live ? ez_n : ez;
        ey = live ? ey_n : ey;
    }
}

// epilogue of one C tile into the wave-private LDS transpose buffer (rows beyond T land in rows never stored)
template <int MP, int CT>
__device__ __forceinline__ void tile_epilogue(const f32x4& acc0, const f32x4& acc1, const f32x4& acc2,
                                              const float (&dtd)[4], double cp, double cv, const Gains& gn,
                                              float* sSt, unsigned wofs, int D, const int astride = kStageStride,
                                              const int nrows = 4) {
    // astride: floats between the pos / vel / action images; nrows: rows of this lane's four that exist (k_traj_flat's
    // whole-trajectory images have no spare rows behind step T - 1; the transpose buffers do: 4)
    const double pgd = gn.pg, dgd = gn.dg, lod = gn.lo, hid = gn.hi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r >= nrows) break;
        const float p = acc0[r];
        float v;
        if (MP == MPK_MP_PRODMP) v = acc1[r];            // 1/tau is folded into the velocity rows
        else v = (acc1[r] - acc2[r]) * dtd[r];           // forward difference of fp32 positions x (1 / dt)
        float* w = sSt + wofs + r * D;
        w[0] = p;
        w[astride] = v;
        if (CT >= 0 && CT < 3) {
            // float64 without FMA: numpy's promotion in pd_controller.py:21-29 (fp32 desired (+) fp64 state)
            double u;
            if (CT == MPK_CTRL_MOTOR) u = pgd * ((double)p - cp) + dgd * ((double)v - cv);
            else if (CT == MPK_CTRL_POSITION) u = (double)p;
            else u = (double)v;
            u = fmin(fmax(u, lod), hid);
            w[2 * astride] = (float)u;
        }
        // closed loop: actions of steps the plan does not execute are 0; the recurrence lanes overwrite the executed ones
        if (CT >= 3) w[2 * astride] = 0.0f;
    }
}

// ---- LDS accesses with the offset as an instruction immediate (round 4: the kernels are instruction-issue bound, and most of what
// the compiler adds around a 16 x 16 tile is address arithmetic on run-time strides) ------------------------------------------------
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)reinterpret_cast<uintptr_t>(p); }
template <int OFF>
__device__ __forceinline__ void lds_w32(const unsigned ad, const float v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(ad), "v"(v), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ float lds_r32(const unsigned ad) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(ad), "n"(OFF));
    return v;
}

// generic (slow) tile store: partial last row tile whose length is not a multiple of 4, or unaligned outputs.
// Takes plain values (a reference to the kernarg struct would force the whole struct into scratch).
__device__ __noinline__ void store_tile_generic(float* pos, float* vel, float* actions, int mask, int B, int T, int D,
                                                int NTW, const float* sSt, int lane, int b0, int rt, int rows) {
    const int SEG = 16 * D, len = rows * D;      // generic path: never shifted, pitch == SEG
    for (int j = 0; j < 3; ++j) {
        if (!((mask >> j) & 1)) continue;
        float* outp = j == 0 ? pos : (j == 1 ? vel : actions);
        for (int sb = 0; sb < NTW; ++sb) {
            const int bb = b0 + sb;
            if (bb >= B) continue;
            float* gp = outp + ((size_t)bb * T + rt * 16) * D;
            for (int e = lane; e < len; e += 64) gp[e] = sSt[j * kStageStride + sb * SEG + e];
        }
    }
}

// One coalesced float4 store per output array (every (episode, output) segment of a row tile is contiguous in HBM).
// WT = write-through (sc1) stores: for cache-resident batches the dirty lines then leave the L2 while the kernel is
// still computing instead of in one write-back burst at the kernel boundary (rocprof: 11.5 -> 9.8 us at B = 4096);
// for HBM-streaming batches plain stores are faster (3.5 vs 2.8 TB/s at B = 1M), so k_traj_stream keeps WT = false.
// Write-through (sc1) stores.  Default: inline-asm global stores (hipcc does not count them in its vmcnt bookkeeping).
// Round 3 tested the alternative on the suspicion that a later `s_waitcnt vmcnt(N)` for prefetched inputs -- N short by the
// uncounted stores, the queue retiring in order -- makes waves wait for store acknowledgements: (a) the same stores as
// compiler-visible buffer stores (MPK_WT_ASM=0: resource built per store from a wave-uniform base), (b) range-check-
// predicated straight-line stores so that no branch hides them from the count (MPK_WT_PRED=1), (c) the tile-major loop
// re-ordered to collect the next item's inputs before its stores.  Headline launch, alternating builds on one box: asm
// 8.18 us, (a) 8.27 - 8.30, (a + b) 12.1, (a + b + c) 11.2 - 11.4: with seven waves per SIMD the wave that waits is covered
// by the others, while anything that delays or fattens the store issue costs directly.  Kept as build knobs, default off.
#ifndef MPK_WT_PRED
#define MPK_WT_PRED 0            // 1: range-check-predicated straight-line stores in tile_store_sel (A/B build knob)
#endif
#ifndef MPK_WT_ASM
#define MPK_WT_ASM 1             // 0: compiler-visible buffer stores instead of the inline-asm global stores
#endif
#ifndef MPK_STORE_AUX
#define MPK_STORE_AUX 16         // sc1 (build-time knob for A/B runs: 17 = sc0 sc1, 2 = nt, 0 = plain)
#endif
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x2_t __attribute__((ext_vector_type(2)));
struct WtDst { __amdgpu_buffer_rsrc_t rsrc; unsigned off; };
__device__ __forceinline__ WtDst wt_dst(const float* p) {
    const unsigned long long pu = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pu), hi = __builtin_amdgcn_readfirstlane((unsigned)(pu >> 32));
    const unsigned long long base = (((unsigned long long)hi << 32) | lo) - (1ull << 30);
    WtDst d;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, -1, 0x00020000);
    d.off = (unsigned)pu - (unsigned)base;
    return d;
}

template <bool WT>
__device__ __forceinline__ void store16(float* p, const f32x4& v) {
    if (WT && MPK_WT_ASM) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else if (WT) {
        const WtDst d = wt_dst(p);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), d.rsrc, (int)d.off, 0, MPK_STORE_AUX);
    } else {
        *reinterpret_cast<f32x4*>(p) = v;
    }
}

template <bool WT>
__device__ __forceinline__ void store8(float* p, const f32x2& v) {
    if (WT && MPK_WT_ASM) {
        // same cache policy as the 16-byte stores next to it: plain stores into lines that also take write-through
        // stores cost the tile-major kernel half its bandwidth (cfg5 at B = 1024: 14.2 vs 8 us)
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else if (WT) {
        const WtDst d = wt_dst(p);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, v), d.rsrc, (int)d.off, 0, MPK_STORE_AUX);
    } else {
        *reinterpret_cast<f32x2*>(p) = v;
    }
}

template <bool WT>
__device__ __forceinline__ void store4(float* p, float v) {
    if (WT && MPK_WT_ASM) {
        asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else if (WT) {
        const WtDst d = wt_dst(p);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), d.rsrc, (int)d.off, 0, MPK_STORE_AUX);
    } else {
        *p = v;
    }
}

// Write-through store of 16 bytes at `base + off` bytes, PREDICATED by the buffer's range check instead of a branch: a lane
// that must not store passes kWtSkip (beyond num_records = 2 GiB: the hardware discards the store).  Straight-line stores
// are what lets the compiler count them (s_waitcnt vmcnt of a later load wait stays exact); `base` is wave-uniform (an
// output array of the launch -- write-through launches write < 2 GiB per array, enforced by the launchers).
constexpr unsigned kWtSkip = 0xFFFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(const float* base_uniform) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, (int)0x80000000u, 0x00020000);
}
__device__ __forceinline__ void wt_store16(__amdgpu_buffer_rsrc_t r, unsigned off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), r, (int)off, 0, MPK_STORE_AUX);
}

// MASK: which output arrays of the staging image leave (bit 0 pos, bit 1 vel, bit 2 actions)
template <int MASK, int KM, bool WT>
__device__ __forceinline__ void tile_store_sel(const TrajArgs& a, const LaneMap<KM>& L, const float* sSt, int lane,
                                               int b0, int rt, int rows) {
    constexpr bool SP = (MASK & 1) != 0, SV = (MASK & 2) != 0, SA = (MASK & 4) != 0;
    const int D = a.c.D, T = a.c.T, len = rows * D;
    if (a.vec_ok) {
        const int bb = b0 + L.sseg;
        const int lo = (int)ep_shift(a, bb), hi = lo + len, c0 = L.w4;   // valid elements of the padded segment
        const bool in_seg = L.sseg < L.NTW && bb < a.B && c0 < hi && c0 + 4 > lo;
        if (WT && MPK_WT_PRED) {
            // whole chunks: straight-line, range-check-predicated buffer stores (no branch between the wave's loads and
            // its stores: the compiler's vmcnt bookkeeping stays exact, see wt_store16)
            const bool whole = in_seg && c0 >= lo && c0 + 4 <= hi;
            const unsigned off = whole ? (unsigned)((((size_t)bb * T + rt * 16) * D - lo + c0) * sizeof(float)) : kWtSkip;
            const unsigned ro = L.sseg < L.NTW ? L.rofs : 0u;                  // (lanes without a segment read slot 0)
            if (SP) wt_store16(wt_rsrc(a.pos), off, *reinterpret_cast<const f32x4*>(sSt + ro));
            if (SV) wt_store16(wt_rsrc(a.vel), off, *reinterpret_cast<const f32x4*>(sSt + kStageStride + ro));
            if (SA) wt_store16(wt_rsrc(a.actions), off, *reinterpret_cast<const f32x4*>(sSt + 2 * kStageStride + ro));
            if (!a.shifted) return;                                            // T * D % 4 == 0: every chunk is whole
        }
        if (in_seg) {
            const size_t go = ((size_t)bb * T + rt * 16) * D - lo + c0;    // 16-byte aligned by construction
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0;
            if (SP) d0 = *reinterpret_cast<const f32x4*>(sSt + L.rofs);
            if (SV) d1 = *reinterpret_cast<const f32x4*>(sSt + kStageStride + L.rofs);
            if (SA) d2 = *reinterpret_cast<const f32x4*>(sSt + 2 * kStageStride + L.rofs);
            if (c0 >= lo && c0 + 4 <= hi) {
                if (!(WT && MPK_WT_PRED)) {
                    if (SP) store16<WT>(a.pos + go, d0);
                    if (SV) store16<WT>(a.vel + go, d1);
                    if (SA) store16<WT>(a.actions + go, d2);
                }
            } else if (a.td3 == 2) {
                // T*D = 2 mod 4 (e.g. 350 x 7): segment starts and lengths are even, so a partial chunk is exactly its
                // upper half (the chunk straddles the segment start) or its lower half (the end): ONE 8-byte store per
                // array for the head and tail lanes together instead of up to four scalar stores in four branches
                const bool head = c0 < lo;
                const int o = head ? 2 : 0;
                const f32x2 p2 = {head ? d0[2] : d0[0], head ? d0[3] : d0[1]};
                const f32x2 v2 = {head ? d1[2] : d1[0], head ? d1[3] : d1[1]};
                if (SP) store8<WT>(a.pos + go + o, p2);
                if (SV) store8<WT>(a.vel + go + o, v2);
                if (SA) {
                    const f32x2 a2 = {head ? d2[2] : d2[0], head ? d2[3] : d2[1]};
                    store8<WT>(a.actions + go + o, a2);
                }
            } else {                                   // the (at most two) partial chunks of a segment
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c0 + e >= lo && c0 + e < hi) {
                        if (SP) store4<WT>(a.pos + go + e, d0[e]);
                        if (SV) store4<WT>(a.vel + go + e, d1[e]);
                        if (SA) store4<WT>(a.actions + go + e, d2[e]);
                    }
                }
            }
        }
    } else {
        store_tile_generic(a.pos, a.vel, a.actions, MASK, a.B, T, D, L.NTW, sSt, lane, b0, rt, rows);
    }
}

template <int NST, int KM, bool WT>
__device__ __forceinline__ void tile_store(const TrajArgs& a, const LaneMap<KM>& L, const float* sSt, int lane,
                                           int b0, int rt, int rows) {
    tile_store_sel<(NST > 2 ? 7 : 3), KM, WT>(a, L, sSt, lane, b0, rt, rows);
}

// Every kernel-argument field the tile-major prologue needs, demanded in scalar registers at the top of the kernel:
// the compiler then issues ALL their scalar loads in one batch (one scalar-cache miss round trip) instead of where
// each field is first used, which chains two or three dependent misses (~0.2 us each) in front of the first input
// load.  The tile-major kernel lives for ~8 us, so that is worth removing.
__device__ __forceinline__ void demand_args(const TrajArgs& a, unsigned grid_x) {
    asm volatile("" ::"s"(grid_x), "s"(a.c.D), "s"(a.c.nb), "s"(a.c.KT), "s"(a.c.P), "s"(a.c.Kloc), "s"(a.c.off), "s"(a.c.T),
                 "s"(a.c.disable_weights), "s"(a.c.disable_goal), "s"(a.c.goal_off_on), "s"(a.A), "s"(a.aux), "s"(a.TS), "s"(a.params),
                 "s"(a.init_pos), "s"(a.init_vel), "s"(a.pos), "s"(a.vel), "s"(a.actions), "s"(a.c_pos), "s"(a.c_vel),
                 "s"(a.sh), "s"(a.G), "s"(a.vec_ok), "s"(a.pitch), "s"(a.cps), "s"(a.shifted), "s"(a.td3),
                 "s"(a.inv_cps), "s"(a.nrt_magic), "s"(a.gstride));
}

// 16 consecutive floats at a wave-uniform, 16-byte aligned LDS address (the scaled-time steps of a row tile)
__device__ __forceinline__ void load_ds16(const float* __restrict__ p, float (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 x = reinterpret_cast<const float4*>(p)[j];
        v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
    }
}

}  // namespace mpk
