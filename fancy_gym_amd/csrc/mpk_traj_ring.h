// k_traj_ring: the HBM-streaming open-loop step as a wave-specialised producer / store-engine pipeline (round 4)
#pragma once
#include "mpk_tile.h"

namespace mpk {

// ---- episode-major, wave-specialised store engine: k_traj_ring (round 4) ----------------------------------------------
// Round 3's counters named what bounds every launch whose outputs stream to HBM (profiles/r03_streaming_pmc_vs_fill.md): the
// CU keeps 58 write requests in flight where a plain fill keeps 75 at the same ~400-cycle latency -- waves that contract,
// transpose AND store leave the CU's write queue to drain whenever they compute.  Here the two jobs belong to different
// waves of ONE persistent workgroup per CU:
//   waves 0 .. NP-1  (producers)  contract ALL row tiles of "their" episode group on the matrix cores into a slot of an LDS
//                    ring of whole-trajectory images -- no global store, ever; the next group's inputs are requested
//                    before the contraction starts;
//   waves NP ..      (store engine)  do nothing but ds_read_b128 -> global_store_dwordx4, back to back, eight reads then
//                    eight stores, no MFMA / transpose / barrier / vmcnt wait in between, at raised wave priority.
// Work is handed over per BATCH of M consecutive episode groups (M * NTW consecutive episodes): the ring holds NBUF batch
// buffers laid out [array][slot in batch][NTW episodes x T x D], i.e. array-major ACROSS the batch, so the engine writes each
// output array of a batch as ONE contiguous run of M * NTW * T * D floats (22.4 KB at cfg2 with M = 4) -- the "compact write
// front": batch b belongs to workgroup b % gridDim.x, so at any time the whole chip writes inside a window of
// gridDim.x batches (~6 MB) per array, like a fill does.
// Hand-over: per slot a `full` and an `empty` counter in LDS (monotonic: use k of a slot waits for empty == k * NS and
// publishes full = k + 1); release = s_waitcnt lgkmcnt(0) of the publishing wave (a wave's DS operations retire in order, and
// all 64 lanes issue together), acquire = the polling load itself (workgroup-scope atomics: the compiler adds the waits).
// Every spin is bounded (a protocol bug must fail a test, not hang a GPU).
// Same tile arithmetic as k_traj_stream / k_traj_flat (same device functions): same bits.
constexpr int kRingThreads = 768;             // launch bound (12 waves: up to 168 registers); the launcher picks (NP + NS) * 64 <= this
#ifndef MPK_RING_BY_ARRAY
#define MPK_RING_BY_ARRAY 0     // 1: open loop with as many engine waves as output arrays: one array per wave.  A/B build knob: within the
                              // noise of interleaved chunks (three waves: 397 - 416 vs 406 - 422 us; two waves, trajectory only: 283 - 287 vs 279 - 284)
#endif
constexpr int kRingThreadsClosed = 1024;      // closed loop: three roles, 93 registers: up to 16 waves
constexpr int kRingSyncInts = 96;             // full[32] | empty[32] | tickets[8] | tickets published | pad
constexpr unsigned kRingSpinLimit = 1u << 21; // ~0.3 s of polling with the sleep below: then give up -- LOUDLY (ring_fail)

// A role that gives up waiting leaves outputs of its launch unwritten (and, closed loop, the plant / replanning state of some
// episodes advanced): it says so in the handle's fault word before it leaves -- host memory, so that the next entry point on the
// handle (and mpk_check_range) turns it into MPK_EHIP without a synchronisation of its own (round 5; rounds 1 - 4 returned MPK_OK).
// Codes: which role waited for what.
enum : int { kRingFailProducerEmpty = 1, kRingFailTicket = 2, kRingFailEngineFull = 4, kRingFailWriterPost = 8,
             kRingFailConsumerFull = 16, kRingFailConsumerTake = 32 };
__device__ __noinline__ void ring_fail(int* fault, int code) {
    if (fault) __hip_atomic_fetch_or(fault, code << 8 | 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ bool ring_wait(int* flag, int want, int* fault, int code) {
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > kRingSpinLimit) { ring_fail(fault, code); return false; }
    }
    return true;
}

// The store engine's inner loop, shared by k_traj_ring and k_traj_burst: the NST output arrays of a batch leave one after the
// other, each as ONE contiguous run of n4 float4 (`astride` floats between the arrays' LDS regions, `go` = the batch's first
// element in every output array); wave s of NS takes the 1 KB chunks s, s + NS, ...: U reads, then U stores, nothing else.
template <int NST, int U>
__device__ __forceinline__ void ring_flush(const TrajArgs& a, const float* sB, const int astride, const size_t go, const int n4,
                                           const int s, const int NS, const int lane) {
#pragma unroll 1
    for (int arr = 0; arr < NST; ++arr) {
        float* const outp = (arr == 0 ? a.pos : (arr == 1 ? a.vel : a.actions)) + go;
        const f32x4* src = reinterpret_cast<const f32x4*>(sB + arr * astride);
#pragma unroll 1
        for (int i0 = s * 64; i0 < n4; i0 += 64 * NS * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = i0 + u * 64 * NS + lane;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < n4) v[u] = src[idx];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = i0 + u * 64 * NS + lane;
                if (idx < n4 && !(a.ring_dbg & 2)) {
                    if (a.wt) store16<true>(outp + 4 * (size_t)idx, v[u]);
                    else store16<false>(outp + 4 * (size_t)idx, v[u]);
                }
            }
        }
    }
}

// All row tiles of ONE episode group contracted into its whole-trajectory image sI (pos at sI, vel at sI + astride, actions at
// sI + 2 astride; no global store): the producers' job in k_traj_ring, a wave's first phase in k_traj_burst.  Row tiles
// [rt0, rt1).  A fragments (basis rows of a row tile) are read from LDS one tile AHEAD into registers -- left to the compiler each
// MFMA waits for its own ds_read (it cannot hoist them above the image writes: it cannot prove they do not alias).
template <int MP, int CT, int KM>
__device__ __forceinline__ void ring_contract(const TrajArgs& a, const LaneMap<KM>& L, const float* ap, const float* sAux,
                                              const float (&xb)[KM], const double cp, const double cv, const Gains& gn, float* sI,
                                              const int astride, const int rt0, const int rt1) {
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    const int KP = 4 * KM, TS = a.TS, D = a.c.D, T = a.c.T, TD = T * D;
    const unsigned wbase = (unsigned)(L.bl * TD + 4 * L.q * D + L.d);   // (episode, row 4q, column) inside a slot image
    float afn[NOUT][KM];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int m = 0; m < KM; ++m) afn[o][m] = ap[(o * KP + 4 * m) * TS + rt0 * 16];
    for (int rt = rt0; rt < rt1; ++rt) {
        float af[NOUT][KM];
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int m = 0; m < KM; ++m) af[o][m] = afn[o][m];
        if (rt + 1 < rt1) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int m = 0; m < KM; ++m) afn[o][m] = ap[(o * KP + 4 * m) * TS + (rt + 1) * 16];
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < KM; ++m) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][m], xb[m], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][m], xb[m], acc1, 0, 0, 0);
            if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[NOUT > 2 ? 2 : 0][m], xb[m], acc2, 0, 0, 0);
        }
        float dtd[4] = {1.f, 1.f, 1.f, 1.f};
        if (MP == MPK_MP_PROMP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dtd[r] = sAux[rt * 16 + 4 * L.q + r];
        }
        const int nrows = min(4, T - rt * 16 - 4 * L.q);      // rows of this lane that exist (<= 0: none)
        if (L.dvalid && nrows > 0)
            tile_epilogue<MP, CT>(acc0, acc1, acc2, dtd, cp, cv, gn, sI, wbase + (unsigned)(rt * 16 * D), D, astride, nrows);
    }
}

// ---- the producers' and the engine's inner loops on an instruction diet (round 4) -------------------------------------------------
// PMC of the streaming row (profiles/r04_ring_pmc.md): k_traj_flat issues 99 vector + 36 scalar + 18 LDS instructions per
// (wave, row tile), and with two waves per SIMD a wave's instruction count is its time (one instruction per ~6 cycles for a lone
// wave: profiles/r04_fp64_rate_probe.md); 40 of the 99 are the
// float64 controller (fixed by the bit-exactness contract), most of the rest is address arithmetic the compiler cannot fold because
// D, T and the image strides are run-time values.  With D a COMPILE-TIME constant (DC: the reference's MP-registered environments
// have 5 or 7 DoF) every LDS access of a row tile takes its offset as an instruction immediate from four address registers that
// advance once per two tiles:
//   * A fragments: ds_read_b32 offset:(tile in pair) * 64, ping-pong register sets over an unrolled pair of tiles (no copies);
//   * the C tile's 12 image stores: ds_write_b32 offset:((tile in pair) * 16 + row) * DC * 4 from one address register per array;
//   * the last, partial row tile is the only one with row predicates.
// The arithmetic is ring_contract's (the same MFMA operands and order, tile_epilogue's controller expression): same bits.
// PROG (closed loop): `prog` counts the row tiles of the image that are COMPLETE -- the wait in front of a tile's MFMAs is
// lgkmcnt(0), so when tile rt starts every LDS write of tile rt - 1 has retired: lane `plane` adds one there (tiles rt0 + 1 .. last),
// the caller adds the last one after the call.  The consumers start on a batch while its later tiles are still being contracted.
template <int MP, int CT, int KM, int DC, bool PROG = false>
__device__ __forceinline__ void ring_contract_d(const TrajArgs& a, const LaneMap<KM>& L, const float* ap, const float* sAux,
                                                const float (&xb)[KM], const double cp, const double cv, const Gains& gn,
                                                float* sI, const int astride, const int rt0, const int rt1, int* prog = nullptr,
                                                const bool plane = false) {
    // row tiles [rt0, rt1) of the group (a group of a long horizon is contracted by several waves, each its own range)
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    constexpr bool ACT = CT >= 0;
    static_assert(DC > 0 && DC <= 16 && 35 * DC * 4 < 65536, "compile-time DoF");
    const int KP = 4 * KM, TS = a.TS, T = a.c.T, TD = T * DC;
    if (rt0 >= rt1) return;
    unsigned adr[NOUT][KM];                                          // A fragment (o, m) of tile rt0 for this lane
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int m = 0; m < KM; ++m) adr[o][m] = lds_addr(ap + (o * KP + 4 * m) * TS + rt0 * 16);
    unsigned wpos = lds_addr(sI + L.bl * TD + (rt0 * 16 + 4 * L.q) * DC + L.d);   // (episode, row 4q of tile rt0, column), pos image
    unsigned wvel = wpos + 4u * (unsigned)astride, wact = wvel + 4u * (unsigned)astride;
    const double pgd = gn.pg, dgd = gn.dg, lod = __builtin_canonicalize(gn.lo), hid = __builtin_canonicalize(gn.hi);
    float afA[NOUT][KM], afB[NOUT][KM];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int m = 0; m < KM; ++m) afA[o][m] = lds_r32<0>(adr[o][m]);
    // one row tile: `cur` holds its A fragments, the next tile's are requested into `nxt`; RTI = tile inside the unrolled pair
    auto tile = [&](auto rti_tag, auto tail_tag, const int rt, float (&cur)[NOUT][KM], float (&nxt)[NOUT][KM]) {
        constexpr int RTI = decltype(rti_tag)::value;
        constexpr bool TAIL = decltype(tail_tag)::value;
        // this tile's fragments have landed: the wait, then a (free) volatile statement per register that makes every use of `cur`
        // depend on it -- the compiler does not know that an asm ds_read's result arrives later, and volatile asms keep their order
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (PROG && rt > rt0 && plane) __hip_atomic_fetch_add(prog, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int m = 0; m < KM; ++m) asm volatile("" : "+v"(cur[o][m]));
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < KM; ++m) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[0][m], xb[m], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[1][m], xb[m], acc1, 0, 0, 0);
            if (NOUT > 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[NOUT > 2 ? 2 : 0][m], xb[m], acc2, 0, 0, 0);
        }
        if (!TAIL) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int m = 0; m < KM; ++m) nxt[o][m] = lds_r32<(RTI + 1) * 64>(adr[o][m]);
        }
        float dtd[4] = {1.f, 1.f, 1.f, 1.f};
        if (MP == MPK_MP_PROMP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dtd[r] = sAux[rt * 16 + 4 * L.q + r];
        }
        const int nrows = TAIL ? T - rt * 16 - 4 * L.q : 4;           // rows of this lane that exist (tail tile only: may be <= 0)
        if (L.dvalid) {
            auto row = [&](auto r_tag) {
                constexpr int R = decltype(r_tag)::value;
                constexpr int OFF = (RTI * 16 + R) * DC * 4;
                if (TAIL && R >= nrows) return;
                const float p = acc0[R];
                float v;
                if (MP == MPK_MP_PRODMP) v = acc1[R];              // 1/tau is folded into the velocity rows
                else v = (acc1[R] - acc2[R]) * dtd[R];             // forward difference of fp32 positions x (1 / dt)
                lds_w32<OFF>(wpos, p);
                lds_w32<OFF>(wvel, v);
                if (ACT) {
                    // float64 without FMA: numpy's promotion in pd_controller.py:21-29 (fp32 desired (+) fp64 state)
                    double u;
                    if (CT == MPK_CTRL_MOTOR) u = pgd * ((double)p - cp) + dgd * ((double)v - cv);
                    else if (CT == MPK_CTRL_POSITION) u = (double)p;
                    else u = (double)v;
                    lds_w32<OFF>(wact, (float)clip_f64(u, lod, hid));
                }
            };
            row(std::integral_constant<int, 0>()); row(std::integral_constant<int, 1>());
            row(std::integral_constant<int, 2>()); row(std::integral_constant<int, 3>());
        }
    };
    auto bump = [&](const int ntile) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int m = 0; m < KM; ++m) adr[o][m] += 64u * (unsigned)ntile;
        const unsigned db = (unsigned)(ntile * 16 * DC * 4);
        wpos += db; wvel += db; wact += db;
    };
    const std::integral_constant<int, 0> I0; const std::integral_constant<int, 1> I1;
    const std::false_type FULL; const std::true_type TAILT;
    const int NFall = T >> 4;                                        // full row tiles of the horizon
    const int NF = rt1 < NFall ? rt1 : NFall;                        // ... the full ones of this range end here
    const bool tail = (T & 15) != 0 && rt1 > NFall;                  // the partial last tile belongs to this range
    int rt = rt0;
    for (; rt + 2 <= NF; rt += 2) {
        tile(I0, FULL, rt, afA, afB);
        tile(I1, FULL, rt + 1, afB, afA);
        bump(2);
    }
    if (rt < NF) {
        tile(I0, FULL, rt, afA, afB);
        bump(1);
        if (tail) tile(I0, TAILT, rt + 1, afB, afA);
    } else if (tail) {
        tile(I0, TAILT, rt, afA, afB);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (fragments requested past the last tile: landed, unused)
}

// The store engine's loop on the same diet: a wave's FULL 1 KB chunks of an array run go out two at a time -- two ds_read_b128,
// one wait, two global stores in the (scalar base, 32-bit vector offset) form, four integer adds -- and only its last (partial)
// chunks take the bounds-checked path (ring_flush from chunk `k0`).
template <bool WT>
__device__ __forceinline__ void ring_store_pair(const float* gbase, const unsigned vo0, const unsigned vo1, const f32x4& v0,
                                                const f32x4& v1) {
    if (WT) {
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(vo0), "v"(v0), "s"(gbase) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(vo1), "v"(v1), "s"(gbase) : "memory");
    } else {
        asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(vo0), "v"(v0), "s"(gbase) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(vo1), "v"(v1), "s"(gbase) : "memory");
    }
}

template <int NST, bool WT, bool WIDE = false>
__device__ __forceinline__ void ring_flush_d(const TrajArgs& a, const float* sB, const int astride, const size_t go, const int n4,
                                             const int s, const int NS, const int lane, const int arr0 = 0, const int arr1 = NST) {
    const unsigned cstride = (unsigned)NS * 1024u;                    // bytes between this wave's chunks
    const int nfull = n4 >> 6;                                       // full chunks of the run
    const int kf = nfull > s ? (nfull - s + NS - 1) / NS : 0;        // ... of which this wave's
#pragma unroll 1
    for (int arr = arr0; arr < arr1; ++arr) {
        float* const outp = (arr == 0 ? a.pos : (arr == 1 ? a.vel : a.actions)) + go;
        const float* src = sB + arr * astride;
        unsigned voff = (unsigned)(s * 64 + lane) * 16u;
        unsigned lad = lds_addr(src) + voff;
        int k = 0;
        if (!(a.ring_dbg & 2)) {
            // one pair at a time: two reads, wait, two stores.  (A two-pair software pipeline -- the next pair's reads in flight
            // while this pair's stores issue -- keeps MORE stores queued and measured slower in the same run: 468 vs 414 us on a
            // slow-placement box, 443 vs 400 elsewhere; like round 3's occupancy experiments, more in the write queue is not
            // better.  profiles/r04_ring.md)
            if (WIDE != ((a.ring_dbg & 64) != 0)) {
                // four chunks at a time where two arrays pass through the engine (NST == 2).  Closed loop: the LONE engine wave is
                // bound by its own rate where the consumers are light (replanning step at 262 144 episodes: 544 -> 472 us; full
                // horizon 488 -> 479); open loop, trajectory only, two engine waves: 292 -> 282 us at 262 144; with the actions as a
                // third array it changes nothing (414 vs 414).  "ring_dbg" 64 flips the choice (A/B runs)
#pragma unroll 1
                for (; k + 4 <= kf; k += 4) {
                    f32x4 v0, v1, v2, v3;
                    const unsigned lad1 = lad + cstride, vo1 = voff + cstride, lad2 = lad1 + cstride, vo2 = vo1 + cstride;
                    const unsigned lad3 = lad2 + cstride, vo3 = vo2 + cstride;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(lad));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v1) : "v"(lad1));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v2) : "v"(lad2));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v3) : "v"(lad3));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "memory");
                    ring_store_pair<WT>(outp, voff, vo1, v0, v1);
                    ring_store_pair<WT>(outp, vo2, vo3, v2, v3);
                    lad += 4 * cstride; voff += 4 * cstride;
                }
            }
#pragma unroll 1
            for (; k + 2 <= kf; k += 2) {
                f32x4 v0, v1;
                const unsigned lad1 = lad + cstride, vo1 = voff + cstride;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(lad));
                asm volatile("ds_read_b128 %0, %1" : "=v"(v1) : "v"(lad1));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1) : : "memory");
                ring_store_pair<WT>(outp, voff, vo1, v0, v1);
                lad += 2 * cstride; voff += 2 * cstride;
            }
        }
        // the wave's remaining chunks (at most one full one and the run's partial one)
#pragma unroll 1
        for (; (s + NS * k) * 64 < n4; ++k) {
            const int idx = (s + NS * k) * 64 + lane;
            if (idx < n4) {
                const f32x4 v = reinterpret_cast<const f32x4*>(src)[idx];
                if (!(a.ring_dbg & 2)) store16<WT>(outp + 4 * (size_t)idx, v);
            }
        }
    }
}

// CLOSED loop (CT >= 3; round 4, second session): two more roles.  The serial recurrence (controller + plant, float64, no FMA: the
// chain of every closed-loop kernel, 9 dependent operations per step) cannot sit on the producers -- one group per wave is four
// times the serial instructions -- and the lane-quarter kernels (k_traj_quad / duo), which run four recurrences per wave, write
// 896-byte pieces of 6 - 12 output streams per wave: stores alone 153 of 166 us at 65 536 episodes (profiles/r04_closed_loop.md).
// Here:  producers   contract the (pos, vel) images of their groups, no controller, no store; every completed ROW TILE is published;
//        the engine  writes pos and vel of a batch as two contiguous runs as soon as the batch is complete;
//        consumers   (waves NP + NS .. + NC): wave c takes the batches c, c + NC, ... of the workgroup and runs the FOUR recurrences
//                    of a batch at once, one group per lane quarter, reading the desired states from the images the producers left
//                    (offsets as immediates: DC is a compile-time constant) as the tiles arrive; serial inputs one batch ahead; the
//                    actions of a row tile into a wave-private LDS tile set.  Integer replanning state, boundary-condition gather and
//                    plant state are the consumer's, exactly as in k_traj_quad;
//        writers     (one per consumer): pull a posted tile set into registers, hand the staging back, issue the stores -- a third of
//                    the bytes in 896-byte pieces, two thirds in 22 KB runs; the consumers never issue a global store.
// A batch buffer (pos | vel of M = 4 groups) is free again when the engine's NS waves AND the consumer have released it (NS + 1
// increments of the slot's `empty` counter).  Same device functions as the other closed-loop kernels: same bits.
// (profiles/r04_ring_closed.md: geometry sweep, ablations, per-role timelines.)
template <int MP, int CT, int KM, int DC>
__global__ void __launch_bounds__(CT >= 3 ? kRingThreadsClosed : kRingThreads) k_traj_ring(const TrajArgs a, const ActArgs act) {
    static_assert(MP != MPK_MP_DMP && (CT < 3 || DC > 0), "promp / prodmp; closed loop with the DoF count compiled in");
    extern __shared__ __attribute__((aligned(16))) float sTab[];   // [NOUT][KP][TS] rows | [TS] aux | sync | ring (| action tiles)
    constexpr bool CLOSED = CT >= 3;
    constexpr bool ACT = CT >= 0 && !CLOSED;                       // actions computed by the producers' epilogue (frozen state)
    constexpr int CTP = CLOSED ? -1 : CT;                          // the contraction's controller
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    constexpr int NST = 2 + (ACT ? 1 : 0);                         // output arrays that pass through the ring
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = 4 * KM, TS = a.TS, D = c.D, T = c.T, TD = T * D;
    const int NP = a.ring_np, NS = a.ring_ns, M = a.ring_m, NBUF = a.ring_nbuf, R = NBUF * M;
    (void)act;
    float* sA = sTab;
    float* sAux = sTab + NOUT * KP * TS;
    int* sSync = reinterpret_cast<int*>(sAux + TS);
    float* sRing = sAux + TS + kRingSyncInts;
    const int IMG = a.flat_img;                                   // floats per (array, slot) image: NTW * T * D (M * IMG a multiple of 4)
    const int BUF = NST * M * IMG;                                 // floats per batch buffer
    {   // basis tables -> LDS by every wave of the workgroup; sync counters zeroed
        const float4* src = reinterpret_cast<const float4*>(a.A);
        float4* dst = reinterpret_cast<float4*>(sA);
        const int nA4 = (NOUT * KP * TS) >> 2, nX4 = TS >> 2;
        for (int i = threadIdx.x; i < nA4; i += blockDim.x) dst[i] = src[i];
        for (int i = threadIdx.x; i < nX4; i += blockDim.x) reinterpret_cast<float4*>(sAux)[i] = reinterpret_cast<const float4*>(a.aux)[i];
        if (threadIdx.x < kRingSyncInts) sSync[threadIdx.x] = 0;
    }
    __syncthreads();                                              // the only workgroup barrier: the roles part here
    const int nWG = (int)gridDim.x, wg = (int)blockIdx.x;
    const int NBT = (a.G + M - 1) / M;                            // batches of the launch
    // Which batches a workgroup takes, and WHEN.  Default (a.ring_ctr): in order, from ONE device-wide counter -- a ticket is TB
    // consecutive batches, fetched (two tickets ahead, by producer wave 0, under its contraction) with one returning atomic.
    // The pure-store probe says why (profiles/r04_store_engine_probe.md): on a box where EVERY static assignment of batches to
    // persistent workgroups -- contiguous ranges per CU, batch b to workgroup b % gridDim.x, grid-stride -- writes the streaming
    // row's three arrays in 405 - 416 us (on other boxes: 356), in-order dynamic assignment writes them in 336 us, like a fill
    // (320 - 347 us everywhere): what a fill has is that its write front advances IN ORDER across the whole chip, because the
    // dispatcher hands out its workgroups in order; persistent streams drift apart.  One counter word sustains ~88 tickets / us,
    // so a ticket covers >= 2 batches (134 KB of output).  Without a counter (a.ring_ctr == nullptr; "ring_dbg" bit 2 / 4): static
    // contiguous ranges per workgroup / batch b to workgroup b % gridDim.x, for A/B runs.
    // Closed loop: batch b -> workgroup b % gridDim.x by default (the chip still writes inside a window of gridDim.x batches, and the
    // consumers know their next batch without waiting for a ticket: 139 against 147 us at 65 536 episodes); "ring_dbg" 4 = tickets there.
    const bool dynamic = a.ring_ctr != nullptr && (CLOSED ? (a.ring_dbg & 4) != 0 && !(a.ring_dbg & 16) : !(a.ring_dbg & (4 | 16)));
    const bool compact = CLOSED ? !(a.ring_dbg & (4 | 16)) : (a.ring_dbg & 4) != 0;
    const int P = a.ring_parts > 0 ? a.ring_parts : 1;           // waves that share a group's row tiles (long horizons)
    const int TB = a.ring_tb > 0 ? a.ring_tb : 1, IPT = TB * M * P;  // batches / work units per ticket
    const int NT = (NBT + TB - 1) / TB;
    const int per = (NBT + nWG - 1) / nWG;
    int nb_local;
    if (compact) nb_local = wg < NBT ? (NBT - wg + nWG - 1) / nWG : 0;
    else { nb_local = NBT - wg * per; nb_local = nb_local < 0 ? 0 : (nb_local > per ? per : nb_local); }
    int* const sFull = sSync;
    int* const sEmpty = sSync + 32;
    int* const sTick = sSync + 64;                                 // [8] ticket values, [8] = number of tickets published
    int* const sTickN = sSync + 72;
    int* const sDone = sSync + 73;                                 // wave 0 has left: whatever is not published is past the end
    // tickets in flight: wave 0 asks for the ticket of its NEXT item (NP units on) while the tickets it requests in this item are
    // published only at the item's end -- so what is published must reach two rounds ahead: ceil(2 NP / IPT) tickets, + the one being
    // fetched + 1.  (Rounds 4 - 5 had ceil(NP / IPT): enough while a ticket covered several rounds -- five batches --, a wave 0 that
    // waits for itself with tickets of one batch; the launcher keeps IPT >= NP / 2, i.e. ahead <= 6 of the 8 ticket slots.)
    const int ahead = 2 + (2 * NP + IPT - 1) / IPT;
    // global batch of local batch bl; -1: this workgroup is done (waits for the ticket in dynamic mode); -2: protocol time-out
    auto batch_at = [&](int bl) -> int {
        if (!dynamic) return bl < nb_local ? (compact ? bl * nWG + wg : wg * per + bl) : -1;
        const int tl = bl / TB;
        unsigned spins = 0;
        while (__hip_atomic_load(sTickN, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= tl) {
            if (__hip_atomic_load(sDone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 &&
                __hip_atomic_load(sTickN, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= tl) return -1;
            __builtin_amdgcn_s_sleep(4);
            if (++spins > kRingSpinLimit) { ring_fail(a.fault, kRingFailTicket); return -2; }
        }
        const int t = __hip_atomic_load(&sTick[tl & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return t < NT ? t * TB + (bl - tl * TB) : -1;
    };
    // The ticket counter cleans up after itself: a workgroup's wave 0 is the only one that fetches tickets, and when it leaves it
    // says so in the word behind the counter; the last workgroup to say so zeroes both words for the next launch that takes this
    // counter slot (stream order).  Zeroing it in front of every launch cost a launch (a one-thread kernel: +11 us on the 400 us
    // streaming row) or faulted (a hipMemsetAsync NODE inside captured graphs: profiles/r04_ring_closed.md).
    auto leave0 = [&]() {
        if (dynamic && wave == 0 && lane == 0) {
            if (atomicAdd(a.ring_ctr + 1, 1u) == gridDim.x - 1) {
                __hip_atomic_store(a.ring_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.ring_ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (wave < NP) {
        // ---------------- producers: work unit n = ((local batch) * M + (slot in batch)) * P + (part of the group's row tiles);
        // wave p takes n = p, p + NP, ... ----
        int n = wave;
        int requested = 0;                                        // wave 0: tickets requested so far
        if (dynamic && wave == 0) {
            unsigned t0 = 0;
            if (lane == 0) t0 = atomicAdd(a.ring_ctr, (unsigned)ahead);
            t0 = __builtin_amdgcn_readfirstlane(t0);
            if (lane < ahead) __hip_atomic_store(&sTick[lane], (int)t0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(sTickN, ahead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            requested = ahead;
        }
        const int MP_ = M * P;                                    // units per batch
        int b = batch_at(n / MP_);
        if (b < 0) {
            if (dynamic && wave == 0 && lane == 0) __hip_atomic_store(sDone, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            leave0();
            return;
        }
        const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
        const int NRT = (T + 15) >> 4;
        int g = b * M + (n % MP_) / P;
        GroupIn<KM> cur = load_group<MP, ACT, KM>(a, L, g < a.G ? g : a.G - 1);
        Gains gn{0.0, 0.0, 0.0, 0.0};
        if (ACT) gn = kernarg_gains(L.dvalid ? L.d : 0);
        const float* ap = sA + L.q * TS + L.col;
        float xb[KM];
        finish_group<KM>(L, cur, xb);
        double cp = cur.cp, cv = cur.cv;
        for (;;) {
            const int bl = n / MP_, j = (n - bl * MP_) / P, part = n - (bl * M + j) * P;
            const int tper = (NRT + P - 1) / P, rt0 = part * tper, rt1 = min(NRT, rt0 + tper);
            // wave 0: the tickets the items of the NEXT round will need (up to two tickets past this one) are requested now
            // and published after this item's contraction -- the atomic's round trip hides under it
            unsigned tnew = 0;
            int nreq = 0;
            if (dynamic && wave == 0) {
                nreq = bl / TB + ahead - requested;
                if (nreq > 3) nreq = 3;
                if (nreq > 0 && lane == 0) tnew = atomicAdd(a.ring_ctr, (unsigned)nreq);
            }
            const int nn = n + NP;
            const int bx = batch_at(nn / MP_);                    // (static: known; dynamic: its ticket is two rounds old)
            if (bx == -2) { leave0(); return; }
            const int gx = bx >= 0 ? bx * M + (nn % MP_) / P : g;
            GroupIn<KM> nxt = cur;
            if (CLOSED || !(a.ring_dbg & 8)) nxt = load_group<MP, ACT, KM>(a, L, gx < a.G ? gx : a.G - 1);   // in flight across the whole group
            const int buf = bl % NBUF, k = bl / NBUF, slot = buf * M + j;
            if (CLOSED && n / NP < 12) MPK_STAMP(3 * (n / NP));                     // unit start (next unit's loads issued)
            if (!ring_wait(&sEmpty[slot], k * (NS + (CLOSED ? 1 : 0)), a.fault, kRingFailProducerEmpty)) { leave0(); return; }
            if (CLOSED && n / NP < 12) MPK_STAMP(3 * (n / NP) + 1);                 // buffer acquired   // the engine (and the consumer) have released use k - 1 of this slot
            int pub = CLOSED ? NRT : 1;                           // what this unit adds to the slot's `full` count at its end
            if (g < a.G && !(a.ring_dbg & 1)) {
                float* const sI = sRing + buf * BUF + j * IMG;
                if constexpr (DC > 0) {
                    if (a.ring_dbg & 32) ring_contract<MP, CTP, KM>(a, L, ap, sAux, xb, cp, cv, gn, sI, M * IMG, rt0, rt1);
                    else if constexpr (CLOSED) {
                        // (closed loop: P = 1, the unit is the whole group; tiles are published as they complete)
                        ring_contract_d<MP, CTP, KM, DC, true>(a, L, ap, sAux, xb, cp, cv, gn, sI, M * IMG, rt0, rt1, &sFull[slot], lane == 0);
                        pub = 1;
                    } else ring_contract_d<MP, CTP, KM, DC>(a, L, ap, sAux, xb, cp, cv, gn, sI, M * IMG, rt0, rt1);
                } else {
                    ring_contract<MP, CTP, KM>(a, L, ap, sAux, xb, cp, cv, gn, sI, M * IMG, rt0, rt1);
                }
            }
            if (nreq > 0) {                                       // wave 0 only (wave-uniform)
                const int base = __builtin_amdgcn_readfirstlane((int)tnew);
                if (lane < nreq) __hip_atomic_store(&sTick[(requested + lane) & 7], base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                requested += nreq;
            }
            // publish: every DS write of this wave has retired, then one more finished part of the slot (and wave 0's new tickets)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (CLOSED && n / NP < 12) MPK_STAMP(3 * (n / NP) + 2);                 // contracted
            if (lane == 0) {
                // ("ring_dbg" 128, fault injection for the tests: the workgroup's second batch is never published -- every role that
                // depends on it must give up after kRingSpinLimit polls and say so)
                if (!((a.ring_dbg & 128) && bl == 1)) __hip_atomic_fetch_add(&sFull[slot], pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (nreq > 0) __hip_atomic_store(sTickN, requested, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (bx < 0) {
                // this wave is done.  Wave 0 says so: a wave that still waits for a ticket nobody will publish any more is past the
                // end (tickets are monotonic: wave 0 left because ITS next ticket was)
                if (dynamic && wave == 0 && lane == 0) __hip_atomic_store(sDone, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                leave0();
                return;
            }
            finish_group<KM>(L, nxt, xb);
            cp = nxt.cp; cv = nxt.cv;
            n = nn; g = gx;
        }
    } else if (!CLOSED || wave < NP + NS) {
        // ---------------- store engine: wave s of NS takes the 1 KB chunks s, s + NS, ... of every array run of every batch ----
        const int s = wave - NP;
        __builtin_amdgcn_s_setprio(3);
        const int NTW = 16 >> a.sh;
        for (int bl = 0;; ++bl) {
            const int b = batch_at(bl);                           // global batch: episodes [b * M * NTW, ...)
            if (b < 0) return;
            const int buf = bl % NBUF, k = bl / NBUF;
            for (int j = 0; j < M; ++j)
                if (!ring_wait(&sFull[buf * M + j], (k + 1) * (CLOSED ? (T + 15) >> 4 : P), a.fault, kRingFailEngineFull)) return;   // (closed loop: `full` counts row tiles)
            if (CLOSED && bl < 12) MPK_STAMP_AT(40 + 2 * bl, NP * 64);             // batch complete
            const long e0 = (long)b * M * NTW;
            const long left = (long)a.B - e0;
            const int ne = (int)(left < (long)(M * NTW) ? left : (long)(M * NTW));
            const int n4 = ne > 0 ? (ne * TD) >> 2 : 0;           // float4 chunks per array run (a full batch is a whole number
            const int nrem = ne > 0 ? (ne * TD) & 3 : 0;          // of them; the launch's last, ragged batch may leave 1 - 3 floats)
            const float* sB = sRing + buf * BUF;
            if (a.ring_dbg & 32) ring_flush<NST, 8>(a, sB, M * IMG, (size_t)e0 * TD, n4, s, NS, lane);
            else if ((CLOSED && NS == 2) || (MPK_RING_BY_ARRAY && !CLOSED && NS == NST)) {
                // closed loop, two engine waves: one ARRAY each (pos / vel), every run written front to back by one wave (measured
                // equal to interleaved 1 KB chunks: profiles/r04_ring_closed.md)
                if (a.wt) ring_flush_d<NST, true, NST == 2>(a, sB, M * IMG, (size_t)e0 * TD, n4, 0, 1, lane, s, s + 1);
                else ring_flush_d<NST, false, NST == 2>(a, sB, M * IMG, (size_t)e0 * TD, n4, 0, 1, lane, s, s + 1);
            }
            else if (a.wt) ring_flush_d<NST, true, NST == 2>(a, sB, M * IMG, (size_t)e0 * TD, n4, s, NS, lane);
            else ring_flush_d<NST, false, NST == 2>(a, sB, M * IMG, (size_t)e0 * TD, n4, s, NS, lane);
            if (nrem && s == 0 && lane < nrem && !(a.ring_dbg & 2)) {
#pragma unroll 1
                for (int arr = 0; arr < NST; ++arr) {
                    float* const outp = (arr == 0 ? a.pos : (arr == 1 ? a.vel : a.actions)) + (size_t)e0 * TD;
                    const float v = sB[arr * M * IMG + 4 * n4 + lane];
                    if (a.wt) store4<true>(outp + 4 * (size_t)n4 + lane, v); else store4<false>(outp + 4 * (size_t)n4 + lane, v);
                }
            }
            // release the batch buffer: every DS read of this wave has returned (the data sit in registers or are on their way)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (CLOSED && bl < 12) MPK_STAMP_AT(41 + 2 * bl, NP * 64);             // flushed
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane < M) __hip_atomic_fetch_add(&sEmpty[buf * M + lane], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        // ---------------- consumers (closed loop): wave c runs the recurrences of the batches c, c + NC, ... ----------------
        if constexpr (CLOSED) {
            const int NC = a.ring_nc;
            const bool writer = wave >= NP + NS + NC;              // action writer of consumer ci (a.ring_aw: one per consumer)
            const int ci = wave - NP - NS - (writer ? NC : 0);
            const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
            const int NTW = L.NTW, NRT = (T + 15) >> 4;
            float* const sStg = sRing + NBUF * BUF + ci * (4 * kStageStride);
            int* const sPost = sSync + 76 + ci;                    // action tiles the consumer has left in its staging tile set
            int* const sTake = sSync + 84 + ci;                    // ... the writer has pulled into registers
            if (writer) {
                // ---------------- action writers: the consumer's tile sets -> global memory.  A store instruction waits at issue while
                // the CU's memory pipeline is full -- and the store engine keeps it full on purpose: inside the consumer each of the four
                // stores of a tile took ~250 - 600 cycles, a third of the recurrence's time (profiles/r04_ring_closed.md).  Here the
                // waiting is a wave's that has nothing else to do: it walks the consumer's batch sequence, pulls a posted tile set into
                // registers, hands the staging back at once and then issues the stores.
                int n = 0;                                         // tile sets taken so far
                for (int bl = ci;; bl += NC) {
                    const int b = batch_at(bl);
                    if (b < 0) return;
#pragma unroll 1
                    for (int rt = 0; rt < NRT; ++rt, ++n) {
                        const int rows = min(16, T - rt * 16);
                        if (!ring_wait(sPost, n + 1, a.fault, kRingFailWriterPost)) return;
                        f32x4 v[4];
                        bool on[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int bb = (b * M + j) * NTW + L.sseg;
                            on[j] = j < M && L.sseg < NTW && bb < a.B && L.w4 < rows * DC;
                            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (on[j]) v[j] = *reinterpret_cast<const f32x4*>(sStg + j * kStageStride + L.rofs);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __hip_atomic_store(sTake, n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (!(a.ring_dbg & 2)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (on[j]) {
                                    const int bb = (b * M + j) * NTW + L.sseg;
                                    float* const dst = a.actions + ((size_t)bb * T + rt * 16) * DC + L.w4;
                                    if (a.wt) store16<true>(dst, v[j]); else store16<false>(dst, v[j]);
                                }
                            }
                        }
                    }
                }
            }
            const bool handoff = a.ring_aw != 0;                   // the tile sets leave through the writer wave
            int ntile = 0;                                         // tile sets posted so far
            __builtin_amdgcn_s_setprio(3);                         // the chain is the critical path of a batch
            const Gains gq = kernarg_gains(L.dvalid ? L.d : 0);
            const double pgd = gq.pg, dgd = gq.dg, lod = __builtin_canonicalize(gq.lo), hid = __builtin_canonicalize(gq.hi);
            // wave-private action tiles, one per group of the batch: [episode][16 rows x DC] as the transpose images of the other kernels
            float* const sAq = sStg + L.q * kStageStride + L.bl * (16 * DC) + L.d;     // (row 0, this column) of group q's tile
            // a lane's serial inputs for one batch: group b * M + q, column (bl, d); read one batch ahead of the recurrence (the
            // integer replanning state of the episode is advanced at fetch time by its d == 0 lane, as in k_traj_quad)
            struct SerialIn { double qs, qds; int nst, bq; bool on; };
            auto load_serial = [&](const int b) {
                SerialIn si{0.0, 0.0, 0, 0, false};
                const int gsel = b * M + L.q;
                si.bq = gsel * NTW + L.bl;
                si.on = L.dvalid && L.q < M && gsel < a.G && si.bq < a.B;
                if (si.on) {
                    const size_t ix = (size_t)si.bq * DC + L.d;
                    si.qs = a.q_state[ix]; si.qds = a.qd_state[ix];
                    si.nst = T;
                    if (a.rp.traj_steps) si.nst = replan_rule(a.rp, si.bq, T, L.d == 0);
                    else if (a.n_steps) si.nst = min(a.n_steps[si.bq], T);
                }
                return si;
            };
            // the batch after this wave's current one, if its ticket is already known (static assignment: always)
            auto batch_known = [&](const int bl) -> bool {
                return !dynamic || __hip_atomic_load(sTickN, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) > bl / TB;
            };
            int b = batch_at(ci);
            if (b < 0) return;
            SerialIn sn = load_serial(b);
            for (int bl = ci;; bl += NC) {
                const int buf = bl % NBUF, k = bl / NBUF;
                const SerialIn sc = sn;
                int bnext = -3;                                    // -3: not known yet
                if (batch_known(bl + NC)) {
                    bnext = batch_at(bl + NC);
                    if (bnext == -2) return;
                    if (bnext >= 0) sn = load_serial(bnext);       // in flight across this batch's recurrence
                }
                if (bl / NC < 10) MPK_STAMP_AT(80 + 4 * (bl / NC), (NP + NS) * 64);    // batch start
                const bool serial = sc.on;
                const int bq = sc.bq, nst = sc.nst;
                double qs = sc.qs, qds = sc.qds;
                const int tcond = (serial && a.rp.cond_pos) ? min(max(nst - 1, 0), T - 1) : -1;
                const float* const sP = sRing + buf * BUF + (L.q < M ? L.q : 0) * IMG + L.bl * TD + L.d;   // desired pos, step 0
                const float* const sV = sP + M * IMG;
                int seen = 0;                                      // row tiles of the batch known to be complete (minimum over its groups)
#pragma unroll 1
                for (int rt = 0; rt < NRT; ++rt) {
                    const int rows = min(16, T - rt * 16);
                    // tile rt of every group of the batch has been contracted (`full` counts row tiles: k * NRT + rt + 1)
                    if (seen <= rt) {
                        unsigned spins = 0;
                        for (;;) {
                            int mn = 1 << 30;
                            for (int j = 0; j < M; ++j)
                                mn = min(mn, __hip_atomic_load(&sFull[buf * M + j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                            seen = mn - k * NRT;
                            if (seen > rt) break;
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > kRingSpinLimit) { ring_fail(a.fault, kRingFailConsumerFull); return; }
                        }
                    }
                    if (rt == 0 && bl / NC < 10) MPK_STAMP_AT(81 + 4 * (bl / NC), (NP + NS) * 64);   // tile 0 there
                    if (rt == NRT - 1 && bl / NC < 10) MPK_STAMP_AT(82 + 4 * (bl / NC), (NP + NS) * 64);   // last tile there
                    if (tcond >= rt * 16 && tcond < rt * 16 + 16) { // condition_on_desired: the desired state at the last executed step
                        const size_t si = (size_t)bq * DC + L.d;
                        a.rp.cond_pos[si] = sP[tcond * DC];
                        a.rp.cond_vel[si] = sV[tcond * DC];
                    }
                    if (handoff && !ring_wait(sTake, ntile, a.fault, kRingFailConsumerTake)) return;   // the writer holds the previous tile set in registers
                    if (!(a.ring_dbg & 1)) {
                        if (__any(serial && rt * 16 < nst)) {
                            const bool full_tile = tile_fully_executed(serial, nst, rt * 16);
                            if (serial) {
                                if (full_tile)
                                    pd_tile_steps<CT - 3, false>(sP + rt * 16 * DC, sV + rt * 16 * DC, sAq, DC, rt * 16, nst, pgd, dgd, lod, hid,
                                                                 a.plant_dt, qs, qds);
                                else
                                    pd_tile_steps<CT - 3, true>(sP + rt * 16 * DC, sV + rt * 16 * DC, sAq, DC, rt * 16, nst, pgd, dgd, lod, hid,
                                                                a.plant_dt, qs, qds, nullptr, nullptr, rows);
                            }
                        } else if (serial) {                       // no episode of the batch executes a step of this tile: actions 0
#pragma unroll
                            for (int tl = 0; tl < 16; ++tl) sAq[tl * DC] = 0.0f;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    ++ntile;
                    if (handoff) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __hip_atomic_store(sPost, ntile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    // without a writer: the action tiles of the batch's groups as one coalesced float4 store per group (T * D is a
                    // multiple of 4 here, so every 16-byte chunk of a tile segment is whole)
                    if (!handoff && !(a.ring_dbg & 2)) {
#pragma unroll 1
                        for (int j = 0; j < M; ++j) {
                            const int bb = (b * M + j) * NTW + L.sseg;
                            if (L.sseg < NTW && bb < a.B && L.w4 < rows * DC) {
                                const f32x4 v = *reinterpret_cast<const f32x4*>(sStg + j * kStageStride + L.rofs);
                                float* const dst = a.actions + ((size_t)bb * T + rt * 16) * DC + L.w4;
                                if (a.wt) store16<true>(dst, v); else store16<false>(dst, v);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (serial) {
                    const size_t si = (size_t)bq * DC + L.d;
                    a.q_state[si] = qs; a.qd_state[si] = qds;
                }
                // release the batch buffer: every image read of this wave has returned
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (bl / NC < 10) MPK_STAMP_AT(83 + 4 * (bl / NC), (NP + NS) * 64);    // batch done
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane < M) __hip_atomic_fetch_add(&sEmpty[buf * M + lane], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (bnext == -3) {                                 // the next batch's ticket was not out when this one started
                    bnext = batch_at(bl + NC);
                    if (bnext >= 0) sn = load_serial(bnext);
                }
                if (bnext < 0) return;
                b = bnext;
            }
        }
    }
}

// ---- episode-major, SHORT-LIVED workgroups: k_traj_burst (round 4) ------------------------------------------------------------
// The pure-store probe (tools/probes/store_engine_probe.hip, profiles/r04_store_engine_probe.md) says what a fill has that no
// persistent kernel of this library had: its workgroups are short-lived and dispatched in address order.  Three arrays of the
// streaming row written by short-lived workgroups take 338 - 349 us on every box; ANY persistent store pattern (grid-stride, the
// ring's engine, contiguous ranges per CU) takes 356 us on a "fast" box and 413 us on a "slow" one -- the box-to-box spread of
// rounds 2 and 3 is a property of persistent write streams.  So here a workgroup lives for ONE batch of M consecutive episode
// groups: its waves contract the groups into a [array][group][NTW x T x D] image (WPG waves per group split the row tiles),
// one barrier, then ALL waves write the batch's output arrays as contiguous runs (ring_flush) and the workgroup ends; the
// dispatcher starts batch b + (resident workgroups) in its place.  Two or three workgroups fit a CU, so one computes while another
// stores.  The basis tables are staged per workgroup (7.6 KB from L2 against 67 KB written: the price of being short-lived).
template <int MP, int CT, int KM>
__global__ void __launch_bounds__(512) k_traj_burst(const TrajArgs a, const ActArgs act) {
    static_assert(MP != MPK_MP_DMP && CT < 3, "open loop, promp / prodmp");
    extern __shared__ __attribute__((aligned(16))) float sTab[];   // [NOUT][KP][TS] rows | [TS] aux | batch image
    constexpr bool ACT = CT >= 0;
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    constexpr int NST = 2 + (ACT ? 1 : 0);
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = 4 * KM, TS = a.TS, T = c.T, TD = T * c.D;
    const int M = a.ring_m, WPG = a.ring_np, NW = (int)(blockDim.x >> 6);   // groups per batch, waves per group, waves
    (void)act;
    float* sA = sTab;
    float* sAux = sTab + NOUT * KP * TS;
    float* sB = sAux + TS;
    const int IMG = a.flat_img;
    const LaneMap<KM> L = make_lane_map<MP, KM>(a, lane);
    const int NTW = L.NTW, NRT = (T + 15) >> 4;
    const long b = blockIdx.x;
    const int j = wave / WPG, part = wave - j * WPG;              // this wave: group j of the batch, part `part` of its row tiles
    const int g = (int)(b * M) + j;
    const bool have = j < M && g < a.G;
    // the group's inputs are requested before the table copy
    GroupIn<KM> cur;
    if (have) cur = load_group<MP, ACT, KM>(a, L, g);
    Gains gn{0.0, 0.0, 0.0, 0.0};
    if (ACT) gn = kernarg_gains(L.dvalid ? L.d : 0);
    // "ring_dbg" bit 4: no table copy -- every wave reads the A fragments of ITS row tiles straight from the (cache-resident) table,
    // as the tile-major kernel does: a workgroup that lives for one group would otherwise copy 7.6 KB to contract 16.8 KB
    const bool staged = !(a.ring_dbg & 16);
    if (staged) {
        const float4* src = reinterpret_cast<const float4*>(a.A);
        float4* dst = reinterpret_cast<float4*>(sA);
        const int nA4 = (NOUT * KP * TS) >> 2, nX4 = TS >> 2;
        for (int i = threadIdx.x; i < nA4; i += blockDim.x) dst[i] = src[i];
        for (int i = threadIdx.x; i < nX4; i += blockDim.x) reinterpret_cast<float4*>(sAux)[i] = reinterpret_cast<const float4*>(a.aux)[i];
        __syncthreads();
    }
    if (have && !(a.ring_dbg & 1)) {
        float xb[KM];
        finish_group<KM>(L, cur, xb);
        const int per = (NRT + WPG - 1) / WPG;
        const int rt0 = part * per, rt1 = min(NRT, rt0 + per);
        if (staged) ring_contract<MP, CT, KM>(a, L, sA + L.q * TS + L.col, sAux, xb, cur.cp, cur.cv, gn, sB + j * IMG, M * IMG, rt0, rt1);
        else ring_contract<MP, CT, KM>(a, L, a.A + L.q * TS + L.col, a.aux, xb, cur.cp, cur.cv, gn, sB + j * IMG, M * IMG, rt0, rt1);
    }
    __syncthreads();
    const long e0 = b * M * NTW;
    const long left = (long)a.B - e0;
    const int ne = (int)(left < (long)(M * NTW) ? left : (long)(M * NTW));
    ring_flush<NST, 8>(a, sB, M * IMG, (size_t)e0 * TD, (ne * TD) >> 2, wave, NW, lane);
}

// ---- k_traj_flat with the DoF count compiled in (round 4) ----------------------------------------------------------------------------
// k_traj_flat (mpk_traj_flat.h: one wave = one episode group, whole-trajectory images, two or three persistent 4-wave workgroups per
// CU) with the ring's inner loops: the contraction with every LDS offset as an immediate (ring_contract_d) and the flush of the
// group's NTW consecutive episodes as ONE run per array, full 1 KB chunks two at a time in the (scalar base, 32-bit offset) store
// form (ring_flush_d).  Same arithmetic: same bits.  "ring_dbg" bit 32: the flush of k_traj_flat (arrays interleaved per chunk).
template <int MP, int CT, int KM, int DC>
__global__ void __launch_bounds__(256, 2) k_traj_flat_d(const TrajArgs a, const ActArgs act) {
    static_assert(MP != MPK_MP_DMP && CT < 3, "open loop, promp / prodmp");
    extern __shared__ __attribute__((aligned(16))) float sTab[];   // [NOUT][KP][TS] rows + [TS] aux + 4 x image
    constexpr bool ACT = CT >= 0;
    constexpr int NOUT = MP == MPK_MP_PRODMP ? 2 : 3;
    constexpr int NST = 2 + (ACT ? 1 : 0);
    const DevCfg& c = a.c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = 4 * KM, TS = a.TS, T = c.T, TD = T * DC;
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
    float xb[KM];
    finish_group<KM>(L, cur, xb);
    double cp = cur.cp, cv = cur.cv;
    while (g < a.G) {
        const int b0 = g * NTW;
        const int gn_ = g + wstride;
        const GroupIn<KM> nxt = load_group<MP, ACT, KM>(a, L, gn_ < a.G ? gn_ : g);   // in flight across the whole group
        ring_contract_d<MP, CT, KM, DC>(a, L, ap, sAux, xb, cp, cv, gn, sI, IMG, 0, NRT);
        __builtin_amdgcn_wave_barrier();
        const int left = a.B - b0;
        const int ne = left < NTW ? left : NTW;
        const size_t go = (size_t)b0 * TD;
        if (a.ring_dbg & 32) {
            const int TD4 = TD >> 2;
            for (int e = 0; e < ne; ++e) {
                const float* se = sI + e * TD;
                const size_t ge = go + (size_t)e * TD;
                for (int i = lane; i < TD4; i += 64) {
                    const f32x4 p4 = *reinterpret_cast<const f32x4*>(se + 4 * i);
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(se + IMG + 4 * i);
                    if (a.wt) { store16<true>(a.pos + ge + 4 * i, p4); store16<true>(a.vel + ge + 4 * i, v4); }
                    else { store16<false>(a.pos + ge + 4 * i, p4); store16<false>(a.vel + ge + 4 * i, v4); }
                    if (ACT) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(se + 2 * IMG + 4 * i);
                        if (a.wt) store16<true>(a.actions + ge + 4 * i, a4); else store16<false>(a.actions + ge + 4 * i, a4);
                    }
                }
            }
        } else if (a.wt) {
            ring_flush_d<NST, true>(a, sI, IMG, go, (ne * TD) >> 2, 0, 1, lane);
        } else {
            ring_flush_d<NST, false>(a, sI, IMG, go, (ne * TD) >> 2, 0, 1, lane);
        }
        __builtin_amdgcn_wave_barrier();                          // the image is free again
        finish_group<KM>(L, nxt, xb);
        cp = nxt.cp; cv = nxt.cv;
        g = gn_;
    }
}

}  // namespace mpk
