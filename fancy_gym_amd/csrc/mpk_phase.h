// What the per-episode-phase kernel families share (k_traj_phase in mpk_traj_phase.hip, k_phase_fused in mpk_phase_fused.hip): the
// neighbour-lane moves of ProMP's forward difference and the (episode, step) item contractions with the DoF count compiled in.
// Both families call THESE functions, so that a fused launch and the trajectory-only launch produce the same bits.
#pragma once
#include "mpk_tile.h"

namespace mpk {

// the value of the neighbouring lane (lane + 1 / lane - 1 of the 64) as ONE vector instruction (DPP wave shift) instead
// of an LDS round trip (ds_bpermute behind __shfl_*): the ProMP velocity takes two of them per (step, DoF).  The lane
// without a neighbour reads 0; nothing uses it.
__device__ __forceinline__ float lane_above(float x) {      // x of lane + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_below(float x) {      // x of lane - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}

// The DC (position, velocity) contractions of one (episode, step) item with the DoF count at compile time: every column first, then
// the DC chains side by side -- each chain's own order of operations (ascending k, one v_pk_fma_f32 per k) is the run-time loop's, so
// the results are the same bits; one chain after the other was 8 dependent FMAs behind two LDS reads, DC times in a row.
template <int DC, int KQ>
__device__ __forceinline__ void dofs_unrolled(const float* __restrict__ sX, const float* hq, const float inv_tau, float* __restrict__ o0,
                                              float* __restrict__ o1) {
    constexpr int KS = KQ * 4;
    float x[DC][KS];
#pragma unroll
    for (int d = 0; d < DC; ++d)
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(sX + d * KS + 4 * j);
            x[d][4 * j + 0] = v.x; x[d][4 * j + 1] = v.y; x[d][4 * j + 2] = v.z; x[d][4 * j + 3] = v.w;
        }
    f32x2 pv[DC];
#pragma unroll
    for (int d = 0; d < DC; ++d) pv[d] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int d = 0; d < DC; ++d)
            pv[d] = __builtin_elementwise_fma(f32x2{hq[2 * k], hq[2 * k + 1]}, f32x2{x[d][k], x[d][k]}, pv[d]);
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        o0[d] = pv[d][0];
        o1[d] = pv[d][1] * inv_tau;
    }
}

// ... and ProMP's: position chains side by side, then each DoF's forward difference over the lanes (lane_above / lane_below, see there)
template <int DC, int KQ>
__device__ __forceinline__ void dofs_unrolled_promp(const float* __restrict__ sX, const float (&h)[KQ * 4], const float rdt, const bool last_row,
                                                    float* __restrict__ o0, float* __restrict__ o1) {
    constexpr int KS = KQ * 4;
    float x[DC][KS];
#pragma unroll
    for (int d = 0; d < DC; ++d)
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(sX + d * KS + 4 * j);
            x[d][4 * j + 0] = v.x; x[d][4 * j + 1] = v.y; x[d][4 * j + 2] = v.z; x[d][4 * j + 3] = v.w;
        }
    float p[DC];
#pragma unroll
    for (int d = 0; d < DC; ++d) p[d] = 0.0f;
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int d = 0; d < DC; ++d) p[d] = fmaf(h[k], x[d][k], p[d]);
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        const float nx = lane_above(p[d]);
        float v = (nx - p[d]) * rdt;
        const float pv = lane_below(v);         // last row repeats the difference before it
        if (last_row) v = pv;
        o0[d] = p[d];
        o1[d] = v;
    }
}

}  // namespace mpk
