// Internal declarations shared by the host side (mpk_host.cpp) and the gfx950 kernels (mpk_kernels.hip).
// Not part of the public ABI (that is include/mpk.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mpk.h"

namespace mpk {

constexpr int kMaxKP = 16;     // padded contraction length supported by the MFMA kernel (multiple of 4)
constexpr int kMaxD = 16;      // DoF supported by the MFMA kernel (one 16-column tile per episode group)
constexpr int kMaxDofArgs = 32;  // per-DoF constants carried in kernel arguments

// ---- host-side float64 tables (construction time) ------------------------------------------------------------
struct HostTables {
    // normalised RBFs: centres in phase space and bandwidths (n_total entries incl. zero padding)
    int n_total = 0;
    std::vector<double> centers, bw;
    // ProDMP pre-computed grid
    int n_pc = 0;
    float scaled_dt = 0.f;  // fp32: basis_dt / tau0, used bit-exactly by times_to_indices
    std::vector<double> y1, y2, dy1, dy2, pos_basis, vel_basis, scale;  // pos/vel_basis: [n_pc, nb+1]
};

void build_rbf(const mpk_config& c, HostTables& t);
void build_prodmp(const mpk_config& c, HostTables& t);
// fp32 time grid linspace(0, duration, T+1)[1:] following torch's symmetric scalar recipe
std::vector<float> build_times(double duration, int T);
int steps_for(double duration, double dt);

// ---- kernel-selection overrides (mpk_set_option); -1 = automatic ----------------------------------------------------
struct Tuning {
    int mapping = -1, bulk = -1, quad = -1, pd_quad = -1, write_through = -1, ipw = -1, phase = -1, phase_table = -1,
        phase_chunk = -1, pd_simple = -1, split = -1, lds_pad = -1, pipe = -1, flat = -1, phase_flat = -1,
        ring = -1, ring_np = -1, ring_ns = -1, ring_m = -1, ring_dbg = -1, ring_parts = -1, tiles_wpb = -1, serial_order = -1, ring_nc = -1,
        pd_generic = -1, dmp_response = -1, ablations = -1, ring_tb = -1, pd_helper = -1, phase_waves = -1, phase_split = -1, phase_pipe = -1, pd_pipe = -1;
};

// ---- device-side configuration (kernel argument, by value) --------------------------------------------------
struct DevCfg {
    int mp_type, phase_type, basis_type;
    int D, nb, n_total, zs;        // n_total: RBF count incl. zero padding; zs: zero-start offset
    int KT, KP;                    // contraction length (learnable + boundary-condition columns), padded to 4
    int P, Kloc, off;              // params per episode, local params per DoF, offset of the local block
    int T;
    int learn_tau, learn_delay, relative_goal, disable_goal, disable_weights;
    int rbf_uniform;               // equally spaced centres, one bandwidth: Gaussians by product recurrence (RbfRecur)
    int relgoal_before_scale;      // MPK_RELGOAL_BEFORE_SCALE
    int dmp_resp;                  // a DMP handle's RESPONSE configuration (round 5): mp_type says PRODMP -- the two-output contraction
                                   // kernels run -- and k_build_shared fills their rows from the explicit Euler map instead (see there)
    int goal_off_on;               // MPK_GOAL_OFFSET_ADD with a non-zero offset: one extra contraction column (x = 1)
    float goal_offset;
    int n_pc, len_factor;
    float tau, delay, alpha_phase, scaled_dt;
    float tau_lo, tau_hi, delay_lo, delay_hi;
    float ws, gs, dmp_alpha, dmp_beta;
    // device tables
    const double* tab;             // ProDMP: [y1|y2|dy1|dy2|pos_basis|vel_basis|weights_goal_scale]; RBF: [centers|bw]
    const float* rows32;           // ProDMP, <= 16 columns: [n_pc][2*KS + 4] = [Psi_0..Psi_nb 0.. y1 y2 | dPsi.. 0.. dy1 dy2 | lo x 4]
    int rows32_stride;             // 2*KS + 4 floats (KS = 8 or 16); DMP handles: the per-episode kernels' interpolation table
                                   // of the forcing rows over the scaled time, [kFastRows = 451][stride = 8] (mpk_traj_phase.hip fast_rows_build)
    const float* base_times;       // [T]
    float t_last;                  // base_times[T - 1] (host side: bounds the scaled time a launch can reach)
};

struct RolloutDev {
    int controller_type, plant_type;
    double dt;
    double pg[kMaxDofArgs], dg[kMaxDofArgs], lo[kMaxDofArgs], hi[kMaxDofArgs];
};

// integer replanning state advanced INSIDE the closed-loop trajectory kernel (mpk_replan_step): all device pointers,
// traj_steps == nullptr switches it off.  Same rule as k_replan_advance, same gather as k_condition_gather.
struct ReplanDev {
    int32_t* traj_steps = nullptr;   // [B] in/out
    int32_t* plan_steps = nullptr;   // [B] in/out
    uint8_t* done = nullptr;         // [B] in/out
    int32_t* seg_len = nullptr;      // [B] out
    uint8_t* done_out = nullptr;     // [B] out, optional: snapshot of `done` after this plan
    float* cond_pos = nullptr;       // [B, D] out, optional: desired state at the last executed step
    float* cond_vel = nullptr;
    int every = 1, max_planning_times = 0, horizon = 0;
};

// validity gate of the fused closed-loop entry points (mpk_validity_gate of include/mpk.h with the limits copied in)
struct GateDev {
    double lo[kMaxD], hi[kMaxD];     // joint limits
    int check_td = 0;
    double tau_b[2] = {0.0, 0.0}, delay_b[2] = {0.0, 0.0};
    const float* raw_params = nullptr;   // [B, P] the action as passed (NULL: the params of the call)
    uint8_t* valid = nullptr;            // [B] out
    double* penalty = nullptr;           // [B] out, optional
};

// shared-phase table workspace produced by k_build_shared and consumed by k_traj_shared
struct SharedTables {
    float* A = nullptr;    // [n_out][KP][TS]
    float* aux = nullptr;  // [TS]
    int TS = 0, n_out = 0;
};

struct Handle;  // defined in mpk_host.cpp

// ---- launchers implemented in mpk_kernels.hip (all enqueue on `stream`, none synchronise) -------------------
int launch_build_shared(const DevCfg& c, float init_time, const SharedTables& st, int32_t* idx_out,
                        int32_t* range_flag, void* stream);
int launch_traj_shared(const DevCfg& c, const SharedTables& st, const float* params, const float* init_pos,
                       const float* init_vel, float* pos, float* vel, float* actions, const RolloutDev* rc,
                       const double* c_pos, const double* c_vel, double* q_state, double* qd_state,
                       const int32_t* n_steps, int B, int num_cu, void* stream, const char** kernel_name,
                       const Tuning& tune, const ReplanDev* rp = nullptr, unsigned* ticket = nullptr, int* fault = nullptr,
                       const GateDev* gate = nullptr);
int launch_episode_return(const DevCfg& c, const SharedTables& st, const float* params, const float* init_pos, const float* init_vel,
                          const RolloutDev& rc, double* q_state, double* qd_state, const int32_t* n_steps, const ReplanDev* rp,
                          int reward_type, const double* goal, const int32_t* step0, int steps_before_reward, int agg, double* ret,
                          int32_t* seg_out, int B, int num_cu, void* stream, const char** kernel_name, const Tuning& tune,
                          const GateDev* gate = nullptr);
int launch_reward_aggregate(const double* rewards, const int32_t* seg_len, int agg, double* out, int B, int T, void* stream);
// shared phase, more than kMaxKP contraction columns: k-chunked GEMM on the matrix cores (trajectory only)
int launch_traj_wide(const DevCfg& c, const SharedTables& st, const float* params, const float* init_pos,
                     const float* init_vel, float* pos, float* vel, int B, int num_cu, void* stream,
                     const char** kernel_name);
int launch_traj_rows(const DevCfg& c, const float* params, const float* init_pos, const float* init_vel,
                     const float* init_time, float init_time_shared, float* pos, float* vel, int32_t* range_flag,
                     int B, int num_cu, void* stream, const char** kernel_name, const Tuning& tune);
int launch_pd_rollout(const RolloutDev& rc, int D, const float* des_pos, const float* des_vel, double* q,
                      double* qd, const int32_t* n_steps, float* actions, int B, int T, void* stream,
                      const Tuning& tune, int* fault = nullptr);
// the fused entry points with a per-episode phase (learned tau / delay), promp / prodmp with <= 8 contraction columns and <= 16 DoF
// (mpk_phase_fused.hip): rc.plant_type static = actions for the frozen state (q, qd), double integrator = closed loop; pos == nullptr:
// nothing per step is stored (mpk_episode_return).  MPK_ENOTIMPL for other shapes.
bool phase_fused_capable(const DevCfg& c);
int launch_phase_fused(const DevCfg& c, const float* params, const float* init_pos, const float* init_vel, float init_time_shared,
                       float* pos, float* vel, float* actions, const RolloutDev& rc, double* q, double* qd, const int32_t* n_steps,
                       const ReplanDev* rp, const GateDev* gate, double* ret, int32_t* seg_out, int32_t* range_flag, int B, int num_cu,
                       void* stream, const char** kernel_name, const Tuning& tune, int* fault);
// per-episode-phase DMP: the interpolation table of the forcing rows (mpk_traj_phase.hip fast_rows_build), built once per handle
int fast_rows_floats(const DevCfg& c);      // 0: none for this shape
int fast_rows_stride(const DevCfg& c);      // floats per node = the consuming kernels' KS
int launch_fast_rows_table(const DevCfg& c, float* out, void* stream);
// MPK_DMP_FIRST_IS_STEP: (init_pos, init_vel) advanced by one Euler step from init_time to the first grid time
int launch_dmp_prestep(const DevCfg& c, const float* params, const float* init_pos, const float* init_vel,
                       const float* init_time, float init_time_shared, float* pos1, float* vel1, int B, void* stream);
int launch_condition_gather(const float* pos, const float* vel, const int32_t* seg_len, float* cond_pos, float* cond_vel,
                            int B, int T, int D, void* stream);
int launch_reacher_rollout(const RolloutDev& rc, int D, const float* des_pos, const float* des_vel, double* q,
                           double* qd, const int32_t* n_steps, const int32_t* step0, const double* goal,
                           int steps_before_reward, float* actions, double* rewards, int B, int T, void* stream,
                           const Tuning& tune, int* fault);
int launch_episode_reset(const double* init_q, const double* init_qd, double* q, double* qd, float* cond_pos,
                         float* cond_vel, int32_t* traj_steps, int32_t* plan_steps, uint8_t* done, int B, int D,
                         void* stream);
int launch_gate_flags(const uint8_t* valid, const uint8_t* was_done, const uint8_t* done, uint8_t* terminated, uint8_t* truncated, int B,
                      void* stream);
int launch_replan_advance(int32_t* traj_steps, int32_t* plan_steps, int32_t* seg_len, uint8_t* done, int every,
                          int max_planning_times, int horizon, int T, int B, void* stream, const uint8_t* valid = nullptr);
int launch_validity(const float* pos, const float* params, int P, int D, const double* lo, const double* hi,
                    int check_td, const double* tb, const double* db, uint8_t* valid, double* penalty, int B, int T,
                    void* stream);

int launch_scaled_basis(const DevCfg& c, const float* times, int n, float* out, void* stream);
int launch_div_sweep(float d, uint32_t first, uint64_t count, unsigned long long* mismatches, void* stream);

size_t shared_tables_floats(const DevCfg& c, int* TS, int* n_out);
bool shared_tables_lean(const DevCfg& c);     // k_traj_wide-only shapes: position (prodmp: + velocity) rows, no step-major copy
bool traj_wide_fits(const DevCfg& c);         // false: launch_traj_wide would return MPK_ENOTIMPL (ask before building its table)


void set_error(const std::string& msg);

}  // namespace mpk
