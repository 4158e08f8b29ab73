// Host side of libmpk.so: configuration checks, float64 table pre-compute (construction time), device buffers,
// and the extern "C" entry points declared in include/mpk.h.  Kernels live in mpk_kernels.hip.
//
// The table pre-compute follows the published ProDMP / RBF mathematics as restated in SURVEY.md Appendix A.4
// (the reference delegates it to mp_pytorch's ProDMPBasisGenerator at construction:
// fancy_gym/black_box/factory/basis_generator_factory.py:14-17).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>

#include "mpk_internal.h"

namespace mpk {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

#define MPK_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#call) + ": " + hipGetErrorString(e_));                      \
            return MPK_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

// Entry points run on the handle's device and leave the caller's current device as they found it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false, ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) {
            ok = hipSetDevice(dev) == hipSuccess;
            switched = ok && prev >= 0;
        }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define MPK_ON_DEVICE(dev)                                                                     \
    DeviceGuard device_guard_(dev);                                                            \
    if (!device_guard_.ok) { set_error("hipSetDevice failed"); return MPK_EHIP; }

// process-wide kernel-selection defaults (mpk_set_option with a NULL handle)
static Tuning g_tune;
// Ticket counters per handle (k_traj_ring), 128 bytes apart.  A counter cleans up after itself when its launch ends, so two launches
// may share one only if they cannot overlap: a slot belongs to ONE ordering domain -- a stream (eager launches) or a (stream capture,
// stream) pair (every launch a capture records on one stream; replays of different graphs may overlap each other and eager work, so
// captures never share a slot with anything) -- and is never handed to another (round 5; rounds 3 - 4 took slots round robin modulo
// 256, so a 13th captured graph of 20 ring launches re-used the slots of the first).  When the pool is exhausted a launch takes the
// static batch assignment instead (no counter: correct, ~5 % slower on the streaming row).
constexpr unsigned kTicketSlots = 4096;

// ------------------------------------------------------------------------------------------------------------
// times
// ------------------------------------------------------------------------------------------------------------
int steps_for(double duration, double dt) {
    // python round(duration / dt): half-to-even on the double quotient
    return (int)std::nearbyint(duration / dt);
}

std::vector<float> build_times(double duration, int T) {
    // linspace(0, duration, T+1)[1:] in fp32: step = (end-start)/(steps-1); i < steps/2 -> start + step*i,
    // else end - step*(steps-1-i); one rounding per op (this file is compiled with -ffp-contract=off).
    const int steps = T + 1;
    std::vector<float> out(T);
    const float start = 0.f, end = (float)duration;
    const float step = (end - start) / (float)(steps - 1);
    for (int i = 1; i < steps; ++i) {
        volatile float prod;
        float v;
        if (i < steps / 2) {
            prod = step * (float)i;
            v = start + prod;
        } else {
            prod = step * (float)(steps - 1 - i);
            v = end - prod;
        }
        out[i - 1] = v;
    }
    return out;
}

// ------------------------------------------------------------------------------------------------------------
// phase helpers (float64, construction time)
// ------------------------------------------------------------------------------------------------------------
static double unbound_phase(const mpk_config& c, double t) {
    const double s = (t - c.delay) / c.tau;
    return c.phase_type == MPK_PHASE_LINEAR ? s : std::exp(-c.alpha_phase * s);
}
static double bound_phase(const mpk_config& c, double t) {
    double s = (t - c.delay) / c.tau;
    if (c.phase_type == MPK_PHASE_LINEAR) return std::fmin(std::fmax(s, 0.0), 1.0);
    return std::exp(-c.alpha_phase * std::fmax(s, 0.0));
}

void build_rbf(const mpk_config& c, HostTables& t) {
    int n = c.num_basis;
    int outside = c.num_basis_outside;
    if (c.basis_type == MPK_BASIS_ZERO_RBF) {
        n += c.num_basis_zero_start + c.num_basis_zero_goal;
        outside = 0;
    }
    t.n_total = n;
    t.centers.assign(n, 0.0);
    t.bw.assign(n, 0.0);
    const double dist = n > 1 ? c.tau / (double)(n - 2 * outside - 1) : c.tau;
    const double lo = -outside * dist + c.delay;
    const double hi = c.tau + outside * dist + c.delay;
    for (int k = 0; k < n; ++k) {
        // numpy.linspace: start + k*step, last point pinned to the end value
        double ct = n > 1 ? lo + k * ((hi - lo) / (double)(n - 1)) : lo;
        if (n > 1 && k == n - 1) ct = hi;
        t.centers[k] = unbound_phase(c, ct);
    }
    for (int k = 0; k < n; ++k) {
        double gap = 1.0;  // single basis: no gap exists upstream; one phase unit (MPK_SINGLE_RBF_UNIT_GAP; _REFUSE: check_cfg)
        if (n > 1) gap = k < n - 1 ? t.centers[k + 1] - t.centers[k] : t.centers[n - 1] - t.centers[n - 2];
        t.bw[k] = c.basis_bandwidth_factor / (gap * gap);
    }
}

static void rbf_row(const HostTables& t, double x, int n_used, int first, double* out) {
    // normalised over ALL n_total RBFs, returns columns [first, first + n_used)
    double sum = 0.0;
    std::vector<double> b(t.n_total);
    for (int k = 0; k < t.n_total; ++k) {
        const double dx = x - t.centers[k];
        b[k] = std::exp(-(dx * dx * t.bw[k]) / 2.0);
        sum += b[k];
    }
    for (int k = 0; k < n_used; ++k) out[k] = t.n_total > 1 ? b[first + k] / sum : b[first + k];
}

void build_prodmp(const mpk_config& c, HostTables& t) {
    build_rbf(c, t);  // plain RBFs over the exp phase
    const int nb = c.num_basis, K = nb + 1;
    const float sdt = (float)c.basis_dt / (float)c.tau;  // fp32 division, as the reference's fp32 tensor op
    t.scaled_dt = sdt;
    const int per_unit = (int)std::nearbyint(1.0 / (double)sdt);
    const int N = c.pre_compute_length_factor * per_unit + 1;
    t.n_pc = N;
    t.y1.resize(N); t.y2.resize(N); t.dy1.resize(N); t.dy2.resize(N);
    t.pos_basis.assign((size_t)N * K, 0.0);
    t.vel_basis.assign((size_t)N * K, 0.0);
    t.scale.assign(K, 0.0);
    const double alpha = c.basis_alpha, half = 0.5 * alpha;
    const double L = (double)c.pre_compute_length_factor;
    const double step = L / (double)(N - 1);
    std::vector<double> s(N), dp1((size_t)N * nb), dp2((size_t)N * nb), phi(nb);
    std::vector<double> p1(nb, 0.0), p2(nb, 0.0);
    for (int i = 0; i < N; ++i) {
        s[i] = (i == N - 1) ? L : i * step;
        const double e = std::exp(half * s[i]);
        t.y1[i] = std::exp(-half * s[i]);
        t.y2[i] = s[i] * t.y1[i];
        t.dy1[i] = -half * t.y1[i];
        t.dy2[i] = -half * t.y2[i] + t.y1[i];
        const double pc_time = s[i] * c.tau + c.delay;
        const double x = bound_phase(c, pc_time);
        rbf_row(t, x, nb, 0, phi.data());
        for (int k = 0; k < nb; ++k) {
            dp1[(size_t)i * nb + k] = ((s[i] * e) * x) * phi[k];
            dp2[(size_t)i * nb + k] = (e * x) * phi[k];
        }
        if (i > 0) {
            const double ds = s[i] - s[i - 1];
            for (int k = 0; k < nb; ++k) {  // cumulative trapezoid
                p1[k] += ds * (dp1[(size_t)i * nb + k] + dp1[(size_t)(i - 1) * nb + k]) / 2.0;
                p2[k] += ds * (dp2[(size_t)i * nb + k] + dp2[(size_t)(i - 1) * nb + k]) / 2.0;
            }
        }
        const double q1 = (half * s[i] - 1.0) * e + 1.0;
        const double q2 = half * (e - 1.0);
        for (int k = 0; k < nb; ++k) {
            t.pos_basis[(size_t)i * K + k] = p2[k] * t.y2[i] - p1[k] * t.y1[i];
            t.vel_basis[(size_t)i * K + k] = p2[k] * t.dy2[i] - p1[k] * t.dy1[i];
        }
        t.pos_basis[(size_t)i * K + nb] = q2 * t.y2[i] - q1 * t.y1[i];
        t.vel_basis[(size_t)i * K + nb] = q2 * t.dy2[i] - q1 * t.dy1[i];
    }
    for (int k = 0; k < K; ++k) {
        double m = -std::numeric_limits<double>::infinity();
        for (int i = 0; i < N; ++i) m = std::fmax(m, t.pos_basis[(size_t)i * K + k]);
        t.scale[k] = 1.0 / m;
    }
}

// ------------------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------------------
struct CacheEntry {
    bool valid = false;
    uint32_t key = 0;  // bit pattern of the fp32 init_time
    int T = 0;
    SharedTables st;
    size_t floats = 0;   // allocation size of st.A
    uint64_t stamp = 0;
    // pinned: a captured graph reads this slot, it is never evicted (mpk_unpin_tables / mpk_set_duration release it).
    // deferred: the table was built while a stream capture was active -- the builder kernel is a node of THAT graph, so
    // the content exists only after a replay (and every replay rewrites it); it serves launches of that capture only.
    // A table built eagerly and later used by a capture is pinned but not deferred: its content is already there.
    bool pinned = false, deferred = false;
    unsigned long long capture_id = 0;
};

struct Handle {
    mpk_config cfg{};
    HostTables tab;
    DevCfg dev{};
    int num_cu = 256;
    double duration = 0.0, dt = 0.0;
    std::vector<float> times;
    double* d_tab = nullptr;
    std::vector<float> rows32;     // ProDMP: fp32 row table of the per-episode-phase kernel (see mpk_create)
    float* d_rows32 = nullptr;
    int rows32_stride = 0;
    float* d_times = nullptr;
    int32_t* d_flag = nullptr;   // range-error flag written by kernels
    unsigned* d_tickets = nullptr;   // k_traj_ring: kTicketSlots device-wide batch counters, one cache line apart (see kTicketSlots)
    unsigned ticket_next = 0;        // slots handed out so far
    std::vector<unsigned> ticket_free;   // slots of captures released by mpk_unpin_tables / mpk_set_duration (their graphs are dead)
    std::map<std::pair<unsigned long long, uintptr_t>, unsigned> ticket_of;   // (capture id or 0, stream) -> slot
    std::mutex ticket_mu;
    int* h_fault = nullptr;          // k_traj_ring's fault word: mapped host memory (ring_fail), read here without synchronising
    int* d_fault = nullptr;          // ... its device address
    int32_t* d_idx = nullptr;    // scratch for mpk_prodmp_indices
    int idx_cap = 0;
    static constexpr int kCache = 64;    // distinct init_times of one replanning episode (a 100-step horizon replanned every 2 steps)
    CacheEntry cache[kCache];
    // DMP with a shared phase as a contraction (round 5, k_build_shared `dmp_resp`): the configuration the two-output matrix-core
    // kernels see and the response tables' own slots; resp_ok: shape and stability allow it (fill_devcfg)
    DevCfg dev_resp{};
    bool resp_ok = false;
    CacheEntry cache_resp[kCache];
    std::string kernel_name_buf;       // "k_traj_tiles<prodmp,act>" reported as "k_traj_tiles<dmp_resp,act>"
    uint64_t stamp = 0;
    const char* last_kernel = "";
    Tuning tune;                 // per-handle overrides (mpk_set_option); -1 = follow the process-wide default
    float* d_pre = nullptr;      // MPK_DMP_FIRST_IS_STEP: [2][pre_cap] boundary state after the pre-step
    size_t pre_cap = 0;
};

static Tuning effective_tuning(const Handle* h);   // defined beside the option table

static int check_cfg(const mpk_config& c) {
    if (c.abi_version != MPK_ABI_VERSION) { set_error("mpk_config.abi_version mismatch"); return MPK_EINVAL; }
    if (c.mp_type < 0 || c.mp_type > 2) { set_error("unknown mp_type"); return MPK_EINVAL; }
    if (c.phase_type < 0 || c.phase_type > 1) { set_error("unknown phase_type"); return MPK_EINVAL; }
    if (c.basis_type < 0 || c.basis_type > 2) { set_error("unknown basis_type"); return MPK_EINVAL; }
    if (c.num_dof < 0) { set_error("num_dof must be >= 0"); return MPK_EINVAL; }
    if (c.num_basis < 1) { set_error("num_basis must be >= 1 (basis length 0 is not implemented upstream)"); return MPK_EINVAL; }
    if (c.basis_type == MPK_BASIS_PRODMP && c.phase_type != MPK_PHASE_EXP) {
        // factory/basis_generator_factory.py:16 asserts this
        set_error("prodmp basis requires the exp phase generator");
        return MPK_EINVAL;
    }
    if ((c.mp_type == MPK_MP_PRODMP) != (c.basis_type == MPK_BASIS_PRODMP)) {
        // factory/trajectory_generator_factory.py:17 asserts the prodmp pairing
        set_error("prodmp trajectory generator and prodmp basis generator must be used together");
        return MPK_EINVAL;
    }
    if (!(c.tau > 0.0)) { set_error("tau must be > 0"); return MPK_EINVAL; }
    if (!(c.dt > 0.0) || !(c.duration > 0.0)) { set_error("dt and duration must be > 0"); return MPK_EINVAL; }
    if (c.basis_type == MPK_BASIS_PRODMP && (c.pre_compute_length_factor < 1 || c.pre_compute_length_factor > 6)) {
        set_error("pre_compute_length_factor must be in [1, 6]");
        return MPK_EINVAL;
    }
    if (c.basis_type == MPK_BASIS_RBF && c.num_basis > 1 && c.num_basis - 2 * c.num_basis_outside - 1 <= 0) {
        set_error("num_basis_outside too large for num_basis");
        return MPK_EINVAL;
    }
    if (c.mp_type == MPK_MP_PRODMP && c.disable_goal && c.disable_weights) {
        set_error("disable_goal and disable_weights cannot both be set");
        return MPK_EINVAL;
    }
    if (c.relative_goal_mode < 0 || c.relative_goal_mode > 1 || c.goal_offset_mode < 0 || c.goal_offset_mode > 1 ||
        c.single_rbf_mode < 0 || c.single_rbf_mode > 1 || c.dmp_first_sample < 0 || c.dmp_first_sample > 1) {
        set_error("relative_goal_mode, goal_offset_mode, single_rbf_mode and dmp_first_sample take 0 or 1");
        return MPK_EINVAL;
    }
    if (c.goal_offset_mode == MPK_GOAL_OFFSET_ADD && !std::isfinite(c.goal_offset)) {
        set_error("goal_offset must be finite");
        return MPK_EINVAL;
    }
    if (c.single_rbf_mode == MPK_SINGLE_RBF_REFUSE && c.basis_type != MPK_BASIS_PRODMP) {
        const int n = c.num_basis + (c.basis_type == MPK_BASIS_ZERO_RBF ? c.num_basis_zero_start + c.num_basis_zero_goal : 0);
        if (n == 1) {
            set_error("a single radial basis function has no neighbouring centre to take its bandwidth from "
                      "(single_rbf_mode = MPK_SINGLE_RBF_REFUSE)");
            return MPK_EINVAL;
        }
    }
    return MPK_OK;
}

static int local_per_dof(const mpk_config& c) {
    switch (c.mp_type) {
        case MPK_MP_PROMP: return c.num_basis;
        case MPK_MP_DMP: return c.num_basis + 1;
        default: return (c.disable_weights ? 0 : c.num_basis) + (c.disable_goal ? 0 : 1);
    }
}

static void quantise(mpk_config& q) {
    // the reference holds these constants as fp32 tensors / mixes them into fp32 tensor ops: quantise once so the
    // float64 table build sees exactly the values the reference's arithmetic sees
    auto f32 = [](double v) { return (double)(float)v; };
    q.tau = f32(q.tau); q.delay = f32(q.delay); q.alpha_phase = f32(q.alpha_phase);
    q.basis_bandwidth_factor = f32(q.basis_bandwidth_factor); q.basis_alpha = f32(q.basis_alpha);
    q.basis_dt = f32(q.basis_dt); q.weights_scale = f32(q.weights_scale); q.goal_scale = f32(q.goal_scale);
    q.dmp_alpha = f32(q.dmp_alpha); q.goal_offset = f32(q.goal_offset);
}

static void fill_devcfg(Handle* h) {
    const mpk_config& c = h->cfg;
    DevCfg& d = h->dev;
    d = DevCfg{};
    d.mp_type = c.mp_type; d.phase_type = c.phase_type; d.basis_type = c.basis_type;
    d.D = c.num_dof; d.nb = c.num_basis; d.n_total = h->tab.n_total;
    d.zs = c.basis_type == MPK_BASIS_ZERO_RBF ? c.num_basis_zero_start : 0;
    const bool zero_pad = c.basis_type == MPK_BASIS_ZERO_RBF;
    {   // linear phase: centres equally spaced in phase, one bandwidth -> Gaussians by product recurrence, unless the first
        // one could underflow before the recurrence climbs back (bw (n - 1)^2 D^2 / 2 beyond the float64 exponent range)
        const HostTables& t = h->tab;
        bool uni = c.basis_type != MPK_BASIS_PRODMP && c.phase_type == MPK_PHASE_LINEAR && t.n_total >= 2;
        if (uni) {
            const int n = t.n_total;
            const double D = (t.centers[n - 1] - t.centers[0]) / (double)(n - 1);
            for (int k = 0; k < n && uni; ++k)
                uni = std::fabs(t.centers[k] - (t.centers[0] + k * D)) <= 1e-12 * std::fabs(D) * n &&
                      std::fabs(t.bw[k] - t.bw[0]) <= 1e-12 * t.bw[0];
            // phase in [0, 1] against centres that may lie outside it
            const double span = std::fmax(std::fabs(1.0 - t.centers[0]), std::fabs(0.0 - t.centers[0])) + std::fabs(D) * n;
            uni = uni && 0.5 * t.bw[0] * span * span < 600.0;
        }
        d.rbf_uniform = uni ? 1 : 0;
    }
    d.relgoal_before_scale = c.relative_goal_mode == MPK_RELGOAL_BEFORE_SCALE ? 1 : 0;
    d.goal_off_on = c.mp_type == MPK_MP_PRODMP && c.goal_offset_mode == MPK_GOAL_OFFSET_ADD && c.goal_offset != 0.0 ? 1 : 0;
    d.goal_offset = (float)c.goal_offset;
    if (c.mp_type == MPK_MP_PRODMP) d.KT = c.num_basis + 3 + d.goal_off_on;   // weights, goal, y_b, v_b (, goal offset)
    else if (c.mp_type == MPK_MP_PROMP) d.KT = c.num_basis + (zero_pad ? 1 : 0);  // weights (+ init_pos)
    else d.KT = c.num_basis;                                         // dmp forcing
    d.KP = (d.KT + 3) / 4 * 4;
    d.Kloc = local_per_dof(c);
    d.off = (c.learn_tau ? 1 : 0) + (c.learn_delay ? 1 : 0);
    d.P = d.off + d.D * d.Kloc;
    d.T = steps_for(h->duration, h->dt);
    d.learn_tau = c.learn_tau; d.learn_delay = c.learn_delay;
    d.relative_goal = c.relative_goal; d.disable_goal = c.disable_goal; d.disable_weights = c.disable_weights;
    d.n_pc = h->tab.n_pc; d.len_factor = c.pre_compute_length_factor;
    d.tau = (float)c.tau; d.delay = (float)c.delay; d.alpha_phase = (float)c.alpha_phase;
    d.scaled_dt = h->tab.scaled_dt;
    d.tau_lo = (float)c.tau_bound[0]; d.tau_hi = (float)c.tau_bound[1];
    d.delay_lo = (float)c.delay_bound[0]; d.delay_hi = (float)c.delay_bound[1];
    d.ws = (float)c.weights_scale; d.gs = (float)c.goal_scale;
    d.dmp_alpha = (float)c.dmp_alpha; d.dmp_beta = (float)(c.dmp_alpha / 4.0);
    d.tab = h->d_tab;
    d.rows32 = h->d_rows32;
    d.rows32_stride = h->rows32_stride;
    d.base_times = h->d_times;
    d.t_last = h->times.empty() ? 0.f : h->times.back();
    // DMP, shared phase: the response route (k_build_shared).  Columns weights | goal | y_b | v_b as ProDMP's, DMP's parameter block
    // ([w_1 .. w_nb, g] per DoF) is ProDMP's without disabled parts.  Only where the explicit Euler map is comfortably stable: with
    // beta = alpha / 4 its step matrix has det 1 - h and trace 2 - h - h^2 / 4 (h = alpha ds), eigenvalues inside the unit circle iff
    // h < 4 (sqrt 2 - 1) = 1.657; taken for h <= 1, where the responses decay monotonically -- beyond that the reference's own fp32
    // recurrence oscillates or diverges, and the serial kernels reproduce THAT operation for operation.
    h->resp_ok = false;
    h->dev_resp = d;
    if (c.mp_type == MPK_MP_DMP && !c.learn_tau && !c.learn_delay && d.D >= 1 && d.D <= kMaxD && c.num_basis + 3 <= kMaxKP &&
        !h->times.empty()) {
        float ds_max = 0.f;
        for (size_t i = 0; i + 1 < h->times.size(); ++i) {
            const float s0 = std::fmax((h->times[i] - d.delay) / d.tau, 0.f), s1 = std::fmax((h->times[i + 1] - d.delay) / d.tau, 0.f);
            ds_max = std::fmax(ds_max, s1 - s0);
        }
        DevCfg& r = h->dev_resp;
        r.mp_type = MPK_MP_PRODMP;
        r.dmp_resp = 1;
        r.rows32 = nullptr; r.rows32_stride = 0;      // (a DMP handle's row table is the per-episode kernels' interpolation table)
        r.KT = c.num_basis + 3;
        r.KP = (r.KT + 3) / 4 * 4;
        r.relative_goal = 0; r.disable_goal = 0; r.disable_weights = 0; r.goal_off_on = 0;
        h->resp_ok = (double)d.dmp_alpha * (double)ds_max <= 1.0;
    }
}

static int upload_times(Handle* h) {
    const int T = steps_for(h->duration, h->dt);
    if (T < 1) { set_error("duration/dt gives no time steps"); return MPK_EINVAL; }
    h->times = build_times(h->duration, T);
    if (h->d_times) { (void)hipFree(h->d_times); h->d_times = nullptr; }
    MPK_HIP(hipMalloc((void**)&h->d_times, sizeof(float) * T));
    MPK_HIP(hipMemcpy(h->d_times, h->times.data(), sizeof(float) * T, hipMemcpyHostToDevice));
    for (auto& e : h->cache) { e.valid = false; e.pinned = false; e.deferred = false; }
    for (auto& e : h->cache_resp) { e.valid = false; e.pinned = false; e.deferred = false; }
    return MPK_OK;
}

static void free_handle(Handle* h) {
    if (!h) return;
    DeviceGuard guard(h->cfg.device);
    for (auto* cache : {h->cache, h->cache_resp})
        for (int i = 0; i < Handle::kCache; ++i) {
            if (cache[i].st.A) (void)hipFree(cache[i].st.A);
            if (cache[i].st.aux) (void)hipFree(cache[i].st.aux);
        }
    if (h->d_tab) (void)hipFree(h->d_tab);
    if (h->d_rows32) (void)hipFree(h->d_rows32);
    if (h->d_times) (void)hipFree(h->d_times);
    if (h->d_flag) (void)hipFree(h->d_flag);
    if (h->d_tickets) (void)hipFree(h->d_tickets);
    if (h->h_fault) (void)hipHostFree(h->h_fault);
    if (h->d_idx) (void)hipFree(h->d_idx);
    if (h->d_pre) (void)hipFree(h->d_pre);
    delete h;
}

static bool shared_phase(const Handle* h, const float* init_time) {
    return !h->cfg.learn_tau && !h->cfg.learn_delay && init_time == nullptr;
}

// DMP, shared phase, stable Euler map, <= 16 columns: the response route ("dmp_response" 0 switches it off: A/B runs, and the tests
// that keep the serial kernels covered)
static bool dmp_response(const Handle* h, const float* init_time, const Tuning& tune) {
    return h->resp_ok && h->cfg.mp_type == MPK_MP_DMP && init_time == nullptr && tune.dmp_response != 0;
}

static bool mfma_capable(const Handle* h) {
    return h->dev.D >= 1 && h->dev.D <= kMaxD && h->dev.KP <= kMaxKP;
}

// more than kMaxKP contraction columns OR more than kMaxD DoF with a shared phase: the k-chunked matrix-core kernel
// (k_traj_wide, trajectory only; an episode with D > 16 spans ceil(D / 16) column groups)
// dmp with few columns and D > 16 stays on the per-episode kernels: its serial Euler recurrence, not the contraction, is
// the cost there, and those kernels keep more recurrences in flight (measured, profiles/r03_wide_bench.md)
static bool wide_capable(const Handle* h) {
    if (h->dev.D < 1) return false;
    if (h->dev.KP > kMaxKP) return true;
    return h->dev.D > kMaxD && h->cfg.mp_type != MPK_MP_DMP;
}

// Allocate every cache slot's table for the current (T, KP) up front: a cache miss later only launches the builder
// kernel, never hipMalloc -- so trajectory calls stay legal inside a hipGraph stream capture.
static int prealloc_slots(const DevCfg& dev, CacheEntry* cache, bool wanted) {
    int TS = 0, n_out = 0;
    const size_t nf = shared_tables_floats(dev, &TS, &n_out);
    // a wide table (hundreds of basis functions: 1.7 MB per slot at K = 1000, T = 200 -- shared_tables_lean) gets 8 slots instead of 64
    const int n_slots = !wanted ? 0 : (dev.KP > kMaxKP ? 8 : Handle::kCache);
    for (int slot = 0; slot < Handle::kCache; ++slot) {
        CacheEntry& e = cache[slot];
        const bool keep = slot < n_slots;
        if (e.st.A && (e.st.TS != TS || e.st.n_out != n_out || e.floats != nf || !keep)) {
            (void)hipFree(e.st.A); (void)hipFree(e.st.aux);
            e.st = SharedTables{};
        }
        if (!keep) { e.valid = false; e.pinned = false; e.deferred = false; continue; }
        e.floats = nf;
        if (!e.st.A) {
            MPK_HIP(hipMalloc((void**)&e.st.A, nf * sizeof(float)));
            MPK_HIP(hipMalloc((void**)&e.st.aux, (size_t)TS * sizeof(float)));
            e.st.TS = TS; e.st.n_out = n_out;
        }
        e.valid = false; e.pinned = false; e.deferred = false;
    }
    return MPK_OK;
}

static int prealloc_cache(Handle* h) {
    int rc = prealloc_slots(h->dev, h->cache, mfma_capable(h) || wide_capable(h));
    if (rc != MPK_OK) return rc;
    return prealloc_slots(h->dev_resp, h->cache_resp, h->resp_ok);
}

// returns the cached (or freshly built, enqueued on `stream`) shared tables for init_time; resp: the DMP response tables
static int get_shared(Handle* h, float init_time, void* stream, SharedTables* out, bool resp = false) {
    uint32_t key;
    std::memcpy(&key, &init_time, 4);
    const DevCfg& dev = resp ? h->dev_resp : h->dev;
    CacheEntry* const cache = resp ? h->cache_resp : h->cache;
    const int T = dev.T;
    ++h->stamp;
    bool capturing = false;
    unsigned long long cap_id = 0;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamGetCaptureInfo((hipStream_t)stream, &cs, &cap_id) == hipSuccess) capturing = cs == hipStreamCaptureStatusActive;
        else (void)hipGetLastError();
    }
    for (int i = 0; i < Handle::kCache; ++i) {
        CacheEntry& e = cache[i];
        if (!e.valid || e.key != key || e.T != T) continue;
        if (e.deferred && !(capturing && e.capture_id == cap_id)) continue;
        if (capturing) e.pinned = true;
        e.stamp = h->stamp; *out = e.st;
        return MPK_OK;
    }
    CacheEntry* victim = nullptr;
    for (int i = 0; i < Handle::kCache; ++i) {
        CacheEntry& e = cache[i];
        if (e.pinned || !e.st.A) continue;
        if (!e.valid) { victim = &e; break; }
        if (!victim || e.stamp < victim->stamp) victim = &e;
    }
    if (!victim) {
        set_error("every shared-table slot is pinned by a captured graph: release them with mpk_unpin_tables");
        return MPK_EINVAL;
    }
    // ProDMP range check on the host, with the same fp32 recipe the device uses (reference: RuntimeError)
    if (h->cfg.mp_type == MPK_MP_PRODMP) {
        const float tmax = h->times[T - 1] + init_time;
        const float s = std::fmax((tmax - h->dev.delay) / h->dev.tau, 0.f);
        if (s > (float)h->dev.len_factor) {
            set_error("Time is beyond the pre-computation range. Set larger pre-computation factor");
            return MPK_ERANGE;
        }
    }
    // every slot was allocated for the current (T, KP) by mpk_create / mpk_set_duration (prealloc_cache): a miss only
    // launches the builder -- nothing here allocates, frees or synchronises
    int TS = 0, n_out = 0;
    (void)shared_tables_floats(dev, &TS, &n_out);
    if (!victim->st.A || victim->st.TS != TS || victim->st.n_out != n_out) {
        set_error("internal: shared-table slot does not match the current time grid");
        return MPK_EINVAL;
    }
    int rc = launch_build_shared(dev, init_time, victim->st, nullptr, h->d_flag, stream);
    if (rc != MPK_OK) return rc;
    victim->valid = true; victim->key = key; victim->T = T; victim->stamp = h->stamp;
    victim->pinned = capturing; victim->deferred = capturing; victim->capture_id = cap_id;
    *out = victim->st;
    return MPK_OK;
}

static int fill_rollout(const Handle* h, const mpk_rollout_cfg* rc, RolloutDev* out) {
    if (!rc) { set_error("rollout cfg is NULL"); return MPK_EINVAL; }
    const int D = h->dev.D;
    if (D > kMaxDofArgs) { set_error("num_dof too large for the rollout kernels"); return MPK_EINVAL; }
    if (rc->controller_type < 0 || rc->controller_type > 2) { set_error("unknown controller_type"); return MPK_EINVAL; }
    if (rc->plant_type < 0 || rc->plant_type > 1) { set_error("unknown plant_type"); return MPK_EINVAL; }
    if (!rc->act_low || !rc->act_high) { set_error("act_low/act_high are required"); return MPK_EINVAL; }
    if (rc->controller_type == MPK_CTRL_MOTOR && (!rc->p_gains || !rc->d_gains)) {
        set_error("motor controller needs p_gains and d_gains");
        return MPK_EINVAL;
    }
    out->controller_type = rc->controller_type; out->plant_type = rc->plant_type; out->dt = rc->dt;
    for (int d = 0; d < D; ++d) {
        out->pg[d] = rc->p_gains ? rc->p_gains[d] : 0.0;
        out->dg[d] = rc->d_gains ? rc->d_gains[d] : 0.0;
        out->lo[d] = rc->act_low[d]; out->hi[d] = rc->act_high[d];
    }
    return MPK_OK;
}

}  // namespace mpk

using namespace mpk;
static int pending_ring_fault(Handle* h);     // defined beside traj_common
static void release_capture_tickets(Handle* h);   // defined beside mpk_unpin_tables

// ============================================================================================================
// extern "C"
// ============================================================================================================
extern "C" {

const char* mpk_last_error(void) { return g_err.c_str(); }
int mpk_abi_version(void) { return MPK_ABI_VERSION; }

#ifndef MPK_SOURCE_HASH
#define MPK_SOURCE_HASH "unstamped"
#endif
// one literal serves both readers: the symbol returns the part behind the marker, tools grep the file for the marker
static const char kSourceHashMarker[] = "MPK_SOURCE_HASH=" MPK_SOURCE_HASH;
const char* mpk_source_hash(void) { return kSourceHashMarker + 16; }

int mpk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mpk_create(const mpk_config* cfg, mpk_handle* out) {
    if (!cfg || !out) { set_error("NULL argument"); return MPK_EINVAL; }
    int rc = check_cfg(*cfg);
    if (rc != MPK_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device visible: libmpk has no CPU fallback");
        return MPK_ENODEV;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { set_error("device ordinal out of range"); return MPK_EINVAL; }
    MPK_ON_DEVICE(cfg->device);
    Handle* h = new Handle();
    h->cfg = *cfg;
    quantise(h->cfg);
    cfg = &h->cfg;
    h->duration = cfg->duration; h->dt = cfg->dt;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) h->num_cu = prop.multiProcessorCount;
    // tables
    std::vector<double> packed;
    if (cfg->basis_type == MPK_BASIS_PRODMP) {
        build_prodmp(*cfg, h->tab);
        const HostTables& t = h->tab;
        packed.reserve((size_t)t.n_pc * (4 + 2 * (cfg->num_basis + 1)));
        packed.insert(packed.end(), t.y1.begin(), t.y1.end());
        packed.insert(packed.end(), t.y2.begin(), t.y2.end());
        packed.insert(packed.end(), t.dy1.begin(), t.dy1.end());
        packed.insert(packed.end(), t.dy2.begin(), t.dy2.end());
        packed.insert(packed.end(), t.pos_basis.begin(), t.pos_basis.end());
        packed.insert(packed.end(), t.vel_basis.begin(), t.vel_basis.end());
        for (int k = 0; k <= cfg->num_basis; ++k) {
            // weights_goal_scale[k]: an fp32 tensor product in the reference (scale_factor * weights/goal scale)
            float sc = k < cfg->num_basis ? (float)cfg->weights_scale : (float)cfg->goal_scale;
            if (cfg->auto_scale_basis) sc = (float)t.scale[k] * sc;
            packed.push_back((double)sc);
        }
        // row table for the per-episode-phase kernel, one table index per row of 2*KS + 4 floats (KS = 8 or 16):
        //   [(Psi_0, dPsi_0) (Psi_1, dPsi_1) .. (Psi_{KS-3}, dPsi_{KS-3}) | y1 y2 dy1 dy2]
        // the position / velocity basis values as fp32 PAIRS (columns past nb are 0) -- the operand pairs of the packed fp32
        // FMA that advances the position and the velocity chain in one instruction --, then y1, y2, dy1, dy2 as FLOAT64
        // (two float slots each, 16-byte aligned): the kernel forms the boundary-condition factors xi1..xi4 from them in
        // float64, where rounding y1, y2 to fp32 first would be amplified by the cancellation c1*y1 + c2*y2.  (Round 1 kept
        // separate halves and hi + lo float pairs: two FMAs per k and three instructions per value and step to rebuild.)
        const int K = cfg->num_basis + 1;
        if (K + 2 <= 16) {
            const int KS = K + 2 <= 8 ? 8 : 16, RS = 2 * KS + 4;
            h->rows32.assign((size_t)t.n_pc * RS, 0.0f);
            h->rows32_stride = RS;
            for (int i = 0; i < t.n_pc; ++i) {
                float* r = h->rows32.data() + (size_t)i * RS;
                for (int kk = 0; kk < K; ++kk) {
                    r[2 * kk] = (float)t.pos_basis[(size_t)i * K + kk];
                    r[2 * kk + 1] = (float)t.vel_basis[(size_t)i * K + kk];
                }
                const double v[4] = {t.y1[i], t.y2[i], t.dy1[i], t.dy2[i]};
                std::memcpy(r + 2 * KS - 4, v, sizeof(v));
            }
        }
    } else {
        build_rbf(*cfg, h->tab);
        packed.insert(packed.end(), h->tab.centers.begin(), h->tab.centers.end());
        packed.insert(packed.end(), h->tab.bw.begin(), h->tab.bw.end());
        {   // constants of the product recurrence (mpk_kernels.hip RbfRecur), used when rbf_uniform
            const int n = h->tab.n_total;
            const double D = n > 1 ? (h->tab.centers[n - 1] - h->tab.centers[0]) / (double)(n - 1) : 0.0;
            const double bw0 = h->tab.bw[0];
            packed.push_back(bw0 * D);
            packed.push_back(0.5 * bw0 * D * D);
            packed.push_back(std::exp(-bw0 * D * D));
        }
    }
    auto fail = [&](int code) { free_handle(h); return code; };
    if (hipMalloc((void**)&h->d_tab, packed.size() * sizeof(double)) != hipSuccess) { set_error("hipMalloc(tables) failed"); return fail(MPK_EHIP); }
    if (hipMemcpy(h->d_tab, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy(tables) failed"); return fail(MPK_EHIP); }
    if (!h->rows32.empty()) {
        if (hipMalloc((void**)&h->d_rows32, h->rows32.size() * sizeof(float)) != hipSuccess) { set_error("hipMalloc(row table) failed"); return fail(MPK_EHIP); }
        if (hipMemcpy(h->d_rows32, h->rows32.data(), h->rows32.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy(row table) failed"); return fail(MPK_EHIP); }
    }
    if (hipMalloc((void**)&h->d_flag, sizeof(int32_t)) != hipSuccess) { set_error("hipMalloc(flag) failed"); return fail(MPK_EHIP); }
    if (hipMemset(h->d_flag, 0, sizeof(int32_t)) != hipSuccess) { set_error("hipMemset(flag) failed"); return fail(MPK_EHIP); }
    if (hipHostMalloc((void**)&h->h_fault, 64, hipHostMallocMapped) != hipSuccess) { set_error("hipHostMalloc(fault word) failed"); return fail(MPK_EHIP); }
    *h->h_fault = 0;
    if (hipHostGetDevicePointer((void**)&h->d_fault, h->h_fault, 0) != hipSuccess) { set_error("hipHostGetDevicePointer(fault word) failed"); return fail(MPK_EHIP); }
    if (hipMalloc((void**)&h->d_tickets, kTicketSlots * 128) != hipSuccess) { set_error("hipMalloc(ticket counters) failed"); return fail(MPK_EHIP); }
    if (hipMemset(h->d_tickets, 0, kTicketSlots * 128) != hipSuccess) { set_error("hipMemset(ticket counters) failed"); return fail(MPK_EHIP); }
    rc = upload_times(h);
    if (rc != MPK_OK) return fail(rc);
    fill_devcfg(h);
    if (const int nf = fast_rows_floats(h->dev)) {
        // DMP: the per-episode-phase kernels' interpolation table of the forcing rows, by the device's own row functions (the
        // table IS the exact rows at its nodes) -- a function of the configuration alone
        if (hipMalloc((void**)&h->d_rows32, (size_t)nf * sizeof(float)) != hipSuccess) { set_error("hipMalloc(forcing-row table) failed"); return fail(MPK_EHIP); }
        rc = launch_fast_rows_table(h->dev, h->d_rows32, nullptr);
        if (rc != MPK_OK) return fail(rc);
        if (hipDeviceSynchronize() != hipSuccess) { set_error("forcing-row table kernel failed"); return fail(MPK_EHIP); }
        h->rows32_stride = fast_rows_stride(h->dev);
        fill_devcfg(h);
    }
    rc = prealloc_cache(h);
    if (rc != MPK_OK) return fail(rc);
    *out = reinterpret_cast<mpk_handle>(h);
    return MPK_OK;
}

void mpk_destroy(mpk_handle hh) { free_handle(reinterpret_cast<Handle*>(hh)); }

int mpk_num_params(mpk_handle hh) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    return reinterpret_cast<Handle*>(hh)->dev.P;
}
int mpk_num_steps(mpk_handle hh) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    return reinterpret_cast<Handle*>(hh)->dev.T;
}
int mpk_num_dof(mpk_handle hh) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    return reinterpret_cast<Handle*>(hh)->dev.D;
}

int mpk_params_bounds(mpk_handle hh, float* low, float* high) {
    if (!hh || !low || !high) { set_error("NULL argument"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    int i = 0;
    if (h->cfg.learn_tau) { low[i] = (float)h->cfg.tau_bound[0]; high[i] = (float)h->cfg.tau_bound[1]; ++i; }
    if (h->cfg.learn_delay) { low[i] = (float)h->cfg.delay_bound[0]; high[i] = (float)h->cfg.delay_bound[1]; ++i; }
    for (; i < h->dev.P; ++i) { low[i] = -std::numeric_limits<float>::infinity(); high[i] = std::numeric_limits<float>::infinity(); }
    return MPK_OK;
}

int mpk_set_duration(mpk_handle hh, double duration, double dt) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!(duration > 0.0) || !(dt > 0.0)) { set_error("dt and duration must be > 0"); return MPK_EINVAL; }
    if (duration == h->duration && dt == h->dt) return MPK_OK;
    MPK_ON_DEVICE(h->cfg.device);
    // tables of the previous grid may be in flight
    MPK_HIP(hipDeviceSynchronize());
    h->duration = duration; h->dt = dt;
    release_capture_tickets(h);     // (graphs captured for the previous grid must not be replayed: include/mpk.h)
    int rc = upload_times(h);
    if (rc != MPK_OK) return rc;
    fill_devcfg(h);
    return prealloc_cache(h);
}

namespace {
struct OptKey { const char* name; int Tuning::*field; int64_t lo, hi; };
const OptKey kOptKeys[] = {
    {"mapping", &Tuning::mapping, 0, 2},         {"bulk", &Tuning::bulk, 0, 2},
    {"quad", &Tuning::quad, 0, 4},               {"pd_quad", &Tuning::pd_quad, 0, 3},
    {"write_through", &Tuning::write_through, 0, 1}, {"ipw", &Tuning::ipw, 0, 1 << 20},
    {"phase", &Tuning::phase, 0, 1},             {"phase_table", &Tuning::phase_table, 0, 1},
    {"phase_chunk", &Tuning::phase_chunk, 0, 16}, {"pd_simple", &Tuning::pd_simple, 0, 1},
    {"split", &Tuning::split, 0, 1},             {"lds_pad", &Tuning::lds_pad, 0, 48},
    {"pipe", &Tuning::pipe, 0, 1},               {"flat", &Tuning::flat, 0, 1},
    {"phase_flat", &Tuning::phase_flat, 0, 1},   {"ring", &Tuning::ring, 0, 2},
    {"ring_np", &Tuning::ring_np, 1, 14},        {"ring_ns", &Tuning::ring_ns, 1, 8},
    {"ring_m", &Tuning::ring_m, 1, 8},           {"ring_dbg", &Tuning::ring_dbg, 0, 255},
    {"ring_parts", &Tuning::ring_parts, 1, 8},   {"tiles_wpb", &Tuning::tiles_wpb, 1, 8},
    {"serial_order", &Tuning::serial_order, 0, 2}, {"ring_nc", &Tuning::ring_nc, 1, 6},
    {"pd_generic", &Tuning::pd_generic, 0, 1},   {"dmp_response", &Tuning::dmp_response, 0, 1},
    {"ablations", &Tuning::ablations, 0, 1},     {"ring_tb", &Tuning::ring_tb, 1, 64},
    {"pd_helper", &Tuning::pd_helper, 0, 1},     {"phase_waves", &Tuning::phase_waves, 1, 32},
    {"phase_split", &Tuning::phase_split, 1, 64},   {"phase_pipe", &Tuning::phase_pipe, 0, 1},
    {"pd_pipe", &Tuning::pd_pipe, 0, 1},
};
const OptKey* find_opt(const char* key) {
    if (!key) return nullptr;
    for (const OptKey& k : kOptKeys)
        if (std::strcmp(k.name, key) == 0) return &k;
    return nullptr;
}
}  // namespace

namespace mpk {
// a handle's own setting wins over the process-wide default, key by key
static Tuning effective_tuning(const Handle* h) {
    Tuning t = g_tune;
    for (const OptKey& k : kOptKeys)
        if (h->tune.*(k.field) >= 0) t.*(k.field) = h->tune.*(k.field);
    return t;
}
}  // namespace mpk

int mpk_set_option(mpk_handle hh, const char* key, int64_t value) {
    const OptKey* k = find_opt(key);
    if (!k) { set_error(std::string("unknown option key: ") + (key ? key : "(null)")); return MPK_EINVAL; }
    if (value != MPK_OPT_AUTO && (value < k->lo || value > k->hi)) {
        set_error(std::string("option value out of range for ") + key);
        return MPK_EINVAL;
    }
    Tuning& t = hh ? reinterpret_cast<Handle*>(hh)->tune : g_tune;
    t.*(k->field) = (int)value;
    return MPK_OK;
}

int mpk_get_option(mpk_handle hh, const char* key, int64_t* value) {
    const OptKey* k = find_opt(key);
    if (!k || !value) { set_error(std::string("unknown option key: ") + (key ? key : "(null)")); return MPK_EINVAL; }
    const Tuning t = hh ? effective_tuning(reinterpret_cast<Handle*>(hh)) : g_tune;
    *value = t.*(k->field);
    return MPK_OK;
}

int mpk_check_range(mpk_handle hh, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    MPK_ON_DEVICE(h->cfg.device);
    int32_t flag = 0;
    MPK_HIP(hipMemcpyAsync(&flag, h->d_flag, sizeof(flag), hipMemcpyDeviceToHost, (hipStream_t)stream));
    MPK_HIP(hipStreamSynchronize((hipStream_t)stream));
    {
        const int fr = pending_ring_fault(h);       // (the stream has drained: every fault of its launches is visible)
        if (fr != MPK_OK) {
            // a range flag raised in the same interval is reported WITH the fault, not deferred to the next call
            if (flag) {
                MPK_HIP(hipMemsetAsync(h->d_flag, 0, sizeof(flag), (hipStream_t)stream));
                set_error(std::string(mpk_last_error()) + "; ALSO: Time is beyond the pre-computation range. Set larger pre-computation factor");
            }
            return fr;
        }
    }
    if (!flag) return MPK_OK;
    MPK_HIP(hipMemsetAsync(h->d_flag, 0, sizeof(flag), (hipStream_t)stream));
    set_error("Time is beyond the pre-computation range. Set larger pre-computation factor");
    return MPK_ERANGE;
}

int mpk_poll_fault(mpk_handle hh) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    return pending_ring_fault(reinterpret_cast<Handle*>(hh));
}

// the ticket counters of stream captures go back to the pool when the caller declares those graphs dead (round 6, ADVICE r05: an
// application that re-captures its episode every iteration ran out of the 4 096 slots for good); a counter is zero whenever no launch
// of its domain is in flight (the launch's last workgroup zeroes it), so a recycled slot needs no memset
static void release_capture_tickets(Handle* h) {
    std::lock_guard<std::mutex> lock(h->ticket_mu);
    for (auto it = h->ticket_of.begin(); it != h->ticket_of.end();) {
        if (it->first.first != 0ull) { h->ticket_free.push_back(it->second); it = h->ticket_of.erase(it); }
        else ++it;
    }
}

int mpk_unpin_tables(mpk_handle hh) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    release_capture_tickets(h);
    for (auto* cache : {h->cache, h->cache_resp})
        for (int i = 0; i < Handle::kCache; ++i) {
            CacheEntry& e = cache[i];
            if (e.pinned) { e.pinned = false; if (e.deferred) e.valid = false; e.deferred = false; }
        }
    return MPK_OK;
}

int mpk_times(mpk_handle hh, float* times) {
    if (!hh || !times) { set_error("NULL argument"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    std::memcpy(times, h->times.data(), sizeof(float) * h->times.size());
    return MPK_OK;
}

// the ticket counter of a launch on `stream` (nullptr: pool exhausted -> static batch assignment); see kTicketSlots
static unsigned* ticket_slot(Handle* h, void* stream) {
    unsigned long long cap_id = 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamGetCaptureInfo((hipStream_t)stream, &cs, &cap_id) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (cs != hipStreamCaptureStatusActive) cap_id = 0;
    std::lock_guard<std::mutex> lock(h->ticket_mu);
    const auto key = std::make_pair(cap_id, reinterpret_cast<uintptr_t>(stream));
    auto it = h->ticket_of.find(key);
    if (it == h->ticket_of.end()) {
        unsigned slot;
        if (!h->ticket_free.empty()) { slot = h->ticket_free.back(); h->ticket_free.pop_back(); }
        else if (h->ticket_next < kTicketSlots) slot = h->ticket_next++;
        else return nullptr;
        it = h->ticket_of.emplace(key, slot).first;
    }
    return h->d_tickets + (size_t)it->second * 32;
}

// a role of an EARLIER ring launch on this handle gave up waiting (ring_fail): outputs of that launch are incomplete.  The word lives
// in host memory: reading it costs nothing and synchronises nothing; it is reported once.
static int pending_ring_fault(Handle* h) {
    if (!h->h_fault) return MPK_OK;
    const int f = __atomic_exchange_n(h->h_fault, 0, __ATOMIC_RELAXED);
    if (!(f & 2)) { if (f) __atomic_fetch_or(h->h_fault, f & ~2, __ATOMIC_RELAXED); return MPK_OK; }
    char msg[512];      // (256 until round 6: the text is ~310 characters, and what got cut was "outputs ... are incomplete")
    std::snprintf(msg, sizeof msg, "k_traj_ring: a wave of an earlier launch on this handle gave up waiting for its partner (role mask 0x%x: "
                  "1 producer / buffer, 2 ticket, 4 store engine / batch, 8 action writer, 16 consumer / tile, 32 consumer / writer, 64 reward helper, 128 k_phase_fused pipeline, 256 rollout pipeline): "
                  "outputs (closed loop: plant and replanning state too) of that launch are incomplete", (unsigned)f >> 8);
    set_error(msg);
    return MPK_EHIP;
}

// the one-launch entry points (fused actions, closed loop, replanning step): promp / prodmp with a shared phase on the matrix-core
// kernels, and DMP where its response route applies
static bool fused_capable(const Handle* h) {
    if (!shared_phase(h, nullptr)) return false;
    if (h->cfg.mp_type == MPK_MP_DMP) return dmp_response(h, nullptr, effective_tuning(h));
    return mfma_capable(h);
}

// ... and promp / prodmp with a LEARNED tau / delay on k_phase_fused (round 6: the reference's TableTennis-ProDMP / BeerPong-ProMP families)
static bool fused_phase_capable(const Handle* h) {
    return !shared_phase(h, nullptr) && h->cfg.mp_type != MPK_MP_DMP && phase_fused_capable(h->dev);
}

static int fill_gate(const Handle* h, const mpk_validity_gate* g, GateDev* out) {
    if (!g->pos_low || !g->pos_high || !g->valid) { set_error("validity gate: NULL pos_low / pos_high / valid"); return MPK_EINVAL; }
    if (h->dev.D > kMaxD) { set_error("validity gate: num_dof too large"); return MPK_EINVAL; }
    for (int d = 0; d < h->dev.D; ++d) { out->lo[d] = g->pos_low[d]; out->hi[d] = g->pos_high[d]; }
    out->check_td = g->check_tau_delay ? 1 : 0;
    if (out->check_td && h->dev.P < 2) { set_error("validity gate: check_tau_delay needs at least two parameters per episode"); return MPK_EINVAL; }
    out->tau_b[0] = g->tau_bound[0]; out->tau_b[1] = g->tau_bound[1];
    out->delay_b[0] = g->delay_bound[0]; out->delay_b[1] = g->delay_bound[1];
    out->raw_params = g->raw_params; out->valid = g->valid; out->penalty = g->penalty;
    return MPK_OK;
}

// one k_phase_fused launch (per-episode phase): the common tail of the four fused entry points
static int phase_fused_common(Handle* h, const float* params, const float* init_pos, const float* init_vel, double init_time_shared,
                              float* pos, float* vel, float* actions, const RolloutDev& rd, double* q, double* qd,
                              const int32_t* n_steps, const ReplanDev* rp, const GateDev* gate, double* ret, int32_t* seg_out,
                              int32_t B, void* stream) {
    if (!params || !init_pos || !init_vel) { set_error("NULL buffer"); return MPK_EINVAL; }
    {
        const int fr = pending_ring_fault(h);
        if (fr != MPK_OK) return fr;
    }
    return launch_phase_fused(h->dev, params, init_pos, init_vel, (float)init_time_shared, pos, vel, actions, rd, q, qd, n_steps, rp,
                              gate, ret, seg_out, h->d_flag, B, h->num_cu, stream, &h->last_kernel, effective_tuning(h), h->d_fault);
}

static int traj_common(Handle* h, const float* params, const float* init_pos, const float* init_vel,
                       const float* init_time, double init_time_shared, float* pos, float* vel, float* actions,
                       const RolloutDev* rd, const double* c_pos, const double* c_vel, int32_t B, void* stream,
                       double* q_state = nullptr, double* qd_state = nullptr, const int32_t* n_steps = nullptr,
                       const ReplanDev* rp = nullptr, const GateDev* gate = nullptr) {
    if (B < 0) { set_error("B must be >= 0"); return MPK_EINVAL; }
    if (B == 0 || h->dev.D == 0) return MPK_OK;     // empty batch: nothing to do (buffers may be NULL)
    if (!params || !init_pos || !init_vel || !pos || !vel) { set_error("NULL buffer"); return MPK_EINVAL; }
    MPK_ON_DEVICE(h->cfg.device);
    {
        const int fr = pending_ring_fault(h);
        if (fr != MPK_OK) return fr;
    }
    const Tuning tune = effective_tuning(h);
    if (h->cfg.mp_type == MPK_MP_DMP && h->cfg.dmp_first_sample == MPK_DMP_FIRST_IS_STEP) {
        // the boundary state advanced by one Euler step (k_dmp_prestep) is what the trajectory kernels start from.
        // The scratch grows with the largest batch seen: the first call with a larger batch allocates (include/mpk.h).
        const size_t need = (size_t)B * h->dev.D;
        if (need > h->pre_cap) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
                set_error("MPK_DMP_FIRST_IS_STEP: run one eager call with this batch size before capturing a graph");
                return MPK_EINVAL;
            }
            // the old scratch may still be read by work on ANY stream (and by graphs captured with the smaller batch:
            // those hold the freed pointer and must be re-captured -- include/mpk.h): drain the device, not one stream
            if (h->d_pre) { MPK_HIP(hipDeviceSynchronize()); (void)hipFree(h->d_pre); h->d_pre = nullptr; h->pre_cap = 0; }
            MPK_HIP(hipMalloc((void**)&h->d_pre, 2 * need * sizeof(float)));
            h->pre_cap = need;
        }
        float* p1 = h->d_pre;
        float* v1 = h->d_pre + h->pre_cap;
        int rc = launch_dmp_prestep(h->dev, params, init_pos, init_vel, init_time, (float)init_time_shared, p1, v1, B, stream);
        if (rc != MPK_OK) return rc;
        init_pos = p1; init_vel = v1;
    }
    if (shared_phase(h, init_time) && wide_capable(h) && !actions && !q_state && !rp && traj_wide_fits(h->dev)) {
        SharedTables st;
        int rc = get_shared(h, (float)init_time_shared, stream, &st);
        if (rc != MPK_OK) return rc;
        rc = launch_traj_wide(h->dev, st, params, init_pos, init_vel, pos, vel, B, h->num_cu, stream, &h->last_kernel);
        if (rc != MPK_ENOTIMPL) return rc;      // horizons beyond one row-tile block (promp / dmp): per-episode kernels
    }
    if (dmp_response(h, init_time, tune)) {
        // DMP as a two-output contraction of the Euler map's response rows (k_build_shared): every shared-phase kernel family of
        // ProDMP, fused actions and the closed loop included
        SharedTables st;
        int rc = get_shared(h, (float)init_time_shared, stream, &st, true);
        if (rc != MPK_OK) return rc;
        unsigned* ticket = ticket_slot(h, stream);
        const char* name = "";
        rc = launch_traj_shared(h->dev_resp, st, params, init_pos, init_vel, pos, vel, actions, rd, c_pos, c_vel,
                                q_state, qd_state, n_steps, B, h->num_cu, stream, &name, tune, rp, ticket, h->d_fault, gate);
        if (rc == MPK_OK) {
            h->kernel_name_buf = name;
            const size_t at = h->kernel_name_buf.find("prodmp");
            if (at != std::string::npos) h->kernel_name_buf.replace(at, 6, "dmp_resp");
            h->last_kernel = h->kernel_name_buf.c_str();
        }
        // (horizons beyond the episode-major kernels' LDS: the serial kernels below; fused entry points fall back to two launches)
        if (rc != MPK_ENOTIMPL || actions) return rc;
    }
    if (shared_phase(h, init_time) && mfma_capable(h)) {
        SharedTables st;
        int rc = get_shared(h, (float)init_time_shared, stream, &st);
        if (rc != MPK_OK) return rc;
        unsigned* ticket = ticket_slot(h, stream);
        rc = launch_traj_shared(h->dev, st, params, init_pos, init_vel, pos, vel, actions, rd, c_pos, c_vel,
                                q_state, qd_state, n_steps, B, h->num_cu, stream, &h->last_kernel, tune, rp, ticket, h->d_fault, gate);
        // horizons whose basis tables do not fit the episode-major kernel's LDS: the per-episode kernels below (dmp) or,
        // for fused actions / rollouts, the caller's two-launch path
        if (rc != MPK_ENOTIMPL || actions) return rc;
    }
    if (actions) { set_error("fused actions need a shared-phase configuration with D <= 16 and <= 16 basis columns"); return MPK_ENOTIMPL; }
    return launch_traj_rows(h->dev, params, init_pos, init_vel, init_time, (float)init_time_shared, pos, vel,
                            h->d_flag, B, h->num_cu, stream, &h->last_kernel, tune);
}

int mpk_trajectory(mpk_handle hh, const float* params, const float* init_pos, const float* init_vel,
                   const float* init_time, double init_time_shared, float* pos, float* vel, int32_t B,
                   void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    return traj_common(reinterpret_cast<Handle*>(hh), params, init_pos, init_vel, init_time, init_time_shared,
                       pos, vel, nullptr, nullptr, nullptr, nullptr, B, stream);
}

int mpk_trajectory_actions(mpk_handle hh, const float* params, const float* init_pos, const float* init_vel,
                           double init_time_shared, const mpk_rollout_cfg* rc, const double* c_pos,
                           const double* c_vel, float* pos, float* vel, float* actions, int32_t B, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (B == 0) return MPK_OK;
    if (!actions || !c_pos || !c_vel) { set_error("NULL buffer"); return MPK_EINVAL; }
    RolloutDev rd;
    int r = fill_rollout(h, rc, &rd);
    if (r != MPK_OK) return r;
    MPK_ON_DEVICE(h->cfg.device);
    if (rd.plant_type != MPK_PLANT_STATIC) { set_error("mpk_trajectory_actions tracks a frozen state (MPK_PLANT_STATIC); use mpk_trajectory_rollout"); return MPK_EINVAL; }
    if (fused_capable(h)) {
        r = traj_common(h, params, init_pos, init_vel, nullptr, init_time_shared, pos, vel, actions, &rd, c_pos, c_vel,
                        B, stream);
        if (r != MPK_ENOTIMPL) return r;
    } else if (fused_phase_capable(h) && B > 0 && pos && vel) {
        r = phase_fused_common(h, params, init_pos, init_vel, init_time_shared, pos, vel, actions, rd, const_cast<double*>(c_pos),
                               const_cast<double*>(c_vel), nullptr, nullptr, nullptr, nullptr, nullptr, B, stream);
        if (r != MPK_ENOTIMPL) return r;
    }
    // what the fused kernels do not cover (dmp with a learned phase, > 16 DoF or basis columns): same result
    // from two launches
    r = traj_common(h, params, init_pos, init_vel, nullptr, init_time_shared, pos, vel, nullptr, nullptr, nullptr,
                    nullptr, B, stream);
    if (r != MPK_OK) return r;
    return launch_pd_rollout(rd, h->dev.D, pos, vel, const_cast<double*>(c_pos), const_cast<double*>(c_vel), nullptr,
                             actions, B, h->dev.T, stream, effective_tuning(h), h->d_fault);
}

int mpk_trajectory_rollout(mpk_handle hh, const float* params, const float* init_pos, const float* init_vel,
                           double init_time_shared, const mpk_rollout_cfg* rc, double* q, double* qd,
                           const int32_t* n_steps, float* pos, float* vel, float* actions, int32_t B, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (B == 0) return MPK_OK;
    if (!actions || !q || !qd) { set_error("NULL buffer"); return MPK_EINVAL; }
    RolloutDev rd;
    int r = fill_rollout(h, rc, &rd);
    if (r != MPK_OK) return r;
    MPK_ON_DEVICE(h->cfg.device);
    if (rd.plant_type != MPK_PLANT_DOUBLE_INTEGRATOR) { set_error("mpk_trajectory_rollout integrates MPK_PLANT_DOUBLE_INTEGRATOR; for a frozen state use mpk_trajectory_actions"); return MPK_EINVAL; }
    if (fused_capable(h)) {
        r = traj_common(h, params, init_pos, init_vel, nullptr, init_time_shared, pos, vel, actions, &rd, nullptr,
                        nullptr, B, stream, q, qd, n_steps);
        if (r != MPK_ENOTIMPL) return r;
    } else if (fused_phase_capable(h) && B > 0 && pos && vel) {
        r = phase_fused_common(h, params, init_pos, init_vel, init_time_shared, pos, vel, actions, rd, q, qd, n_steps, nullptr, nullptr,
                               nullptr, nullptr, B, stream);
        if (r != MPK_ENOTIMPL) return r;
    }
    // dmp with a learned phase, horizons beyond the fused kernel's LDS tables, > 16 DoF or basis columns: trajectory kernel + rollout kernel, same result
    r = traj_common(h, params, init_pos, init_vel, nullptr, init_time_shared, pos, vel, nullptr, nullptr, nullptr,
                    nullptr, B, stream);
    if (r != MPK_OK) return r;
    return launch_pd_rollout(rd, h->dev.D, pos, vel, q, qd, n_steps, actions, B, h->dev.T, stream, effective_tuning(h), h->d_fault);
}

int mpk_replan_step_gated(mpk_handle hh, const float* params, const float* init_pos, const float* init_vel,
                          double init_time_shared, const mpk_rollout_cfg* rc, double* q, double* qd,
                          const mpk_replan_state* st, const mpk_validity_gate* gate, float* pos, float* vel, float* actions,
                          int32_t B, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (B < 0) { set_error("B must be >= 0"); return MPK_EINVAL; }
    if (B == 0) return MPK_OK;
    if (!st || !st->traj_steps || !st->plan_steps || !st->done || !st->seg_len) { set_error("NULL replanning state"); return MPK_EINVAL; }
    if ((st->cond_pos == nullptr) != (st->cond_vel == nullptr)) { set_error("cond_pos and cond_vel go together"); return MPK_EINVAL; }
    if (st->every < 1 || st->horizon < 1) { set_error("every and horizon must be >= 1"); return MPK_EINVAL; }
    if (!actions || !q || !qd) { set_error("NULL buffer"); return MPK_EINVAL; }
    if (st->cond_pos && (st->cond_pos == init_pos || st->cond_vel == init_vel)) { set_error("cond_pos / cond_vel must not alias init_pos / init_vel"); return MPK_EINVAL; }
    RolloutDev rd;
    int r = fill_rollout(h, rc, &rd);
    if (r != MPK_OK) return r;
    if (rd.plant_type != MPK_PLANT_DOUBLE_INTEGRATOR) { set_error("mpk_replan_step integrates MPK_PLANT_DOUBLE_INTEGRATOR"); return MPK_EINVAL; }
    GateDev gd;
    if (gate) {
        r = fill_gate(h, gate, &gd);
        if (r != MPK_OK) return r;
    }
    MPK_ON_DEVICE(h->cfg.device);
    ReplanDev rp;
    rp.traj_steps = st->traj_steps; rp.plan_steps = st->plan_steps; rp.done = st->done; rp.seg_len = st->seg_len;
    rp.done_out = st->done_out; rp.cond_pos = st->cond_pos; rp.cond_vel = st->cond_vel;
    rp.every = st->every; rp.max_planning_times = st->max_planning_times; rp.horizon = st->horizon;
    if (fused_capable(h)) {
        // ONE launch: integer state, trajectory, (validity gate,) controller + plant, condition gather
        r = traj_common(h, params, init_pos, init_vel, nullptr, init_time_shared, pos, vel, actions, &rd, nullptr,
                        nullptr, B, stream, q, qd, nullptr, &rp, gate ? &gd : nullptr);
        if (r != MPK_ENOTIMPL) return r;
    } else if (fused_phase_capable(h) && pos && vel) {
        r = phase_fused_common(h, params, init_pos, init_vel, init_time_shared, pos, vel, actions, rd, q, qd, nullptr, &rp,
                               gate ? &gd : nullptr, nullptr, nullptr, B, stream);
        if (r != MPK_ENOTIMPL) return r;
    }
    // what the fused kernels do not cover (dmp with a learned phase, long horizons, > 16 DoF or basis columns): the same
    // result from the separate kernels
    r = traj_common(h, params, init_pos, init_vel, nullptr, init_time_shared, pos, vel, nullptr, nullptr, nullptr,
                    nullptr, B, stream);
    if (r != MPK_OK) return r;
    if (gate) {
        // the plan is judged first; an invalid one finishes its episode (done |= !valid) before the integer rule looks at it
        r = launch_validity(pos, gd.raw_params ? gd.raw_params : params, h->dev.P, h->dev.D, gd.lo, gd.hi, gd.check_td, gd.tau_b, gd.delay_b,
                            gd.valid, gd.penalty, B, h->dev.T, stream);
        if (r != MPK_OK) return r;
    }
    r = launch_replan_advance(rp.traj_steps, rp.plan_steps, rp.seg_len, rp.done, rp.every, rp.max_planning_times,
                              rp.horizon, h->dev.T, B, stream, gate ? gd.valid : nullptr);
    if (r != MPK_OK) return r;
    if (rp.done_out) MPK_HIP(hipMemcpyAsync(rp.done_out, rp.done, (size_t)B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    r = launch_pd_rollout(rd, h->dev.D, pos, vel, q, qd, rp.seg_len, actions, B, h->dev.T, stream, effective_tuning(h), h->d_fault);
    if (r != MPK_OK) return r;
    if (rp.cond_pos) return launch_condition_gather(pos, vel, rp.seg_len, rp.cond_pos, rp.cond_vel, B, h->dev.T, h->dev.D, stream);
    return MPK_OK;
}

int mpk_replan_step(mpk_handle hh, const float* params, const float* init_pos, const float* init_vel,
                    double init_time_shared, const mpk_rollout_cfg* rc, double* q, double* qd,
                    const mpk_replan_state* st, float* pos, float* vel, float* actions, int32_t B, void* stream) {
    return mpk_replan_step_gated(hh, params, init_pos, init_vel, init_time_shared, rc, q, qd, st, nullptr, pos, vel, actions, B, stream);
}

int mpk_episode_return_gated(mpk_handle hh, const float* params, const float* init_pos, const float* init_vel,
                             double init_time_shared, const mpk_rollout_cfg* rc, double* q, double* qd, const mpk_replan_state* st,
                             const mpk_validity_gate* gate, const int32_t* n_steps, int32_t* seg_out, int32_t reward,
                             const double* goal, const int32_t* step0, int32_t steps_before_reward, int32_t agg, double* ret,
                             int32_t B, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (B < 0) { set_error("B must be >= 0"); return MPK_EINVAL; }
    if (B == 0 || h->dev.D == 0) return MPK_OK;
    if (!params || !init_pos || !init_vel || !q || !qd || !ret) { set_error("NULL buffer"); return MPK_EINVAL; }
    if (reward != MPK_REWARD_NONE && reward != MPK_REWARD_SIMPLE_REACHER) { set_error("unknown reward"); return MPK_EINVAL; }
    if (agg < MPK_AGG_SUM || agg > MPK_AGG_LAST) { set_error("unknown reward aggregation"); return MPK_EINVAL; }
    if (reward == MPK_REWARD_SIMPLE_REACHER && !goal) { set_error("the reacher reward needs goal [B, 2]"); return MPK_EINVAL; }
    ReplanDev rp;
    if (st) {
        if (!st->traj_steps || !st->plan_steps || !st->done || !st->seg_len) { set_error("NULL replanning state"); return MPK_EINVAL; }
        if ((st->cond_pos == nullptr) != (st->cond_vel == nullptr)) { set_error("cond_pos and cond_vel go together"); return MPK_EINVAL; }
        if (st->every < 1 || st->horizon < 1) { set_error("every and horizon must be >= 1"); return MPK_EINVAL; }
        if (st->cond_pos && (st->cond_pos == init_pos || st->cond_vel == init_vel)) { set_error("cond_pos / cond_vel must not alias init_pos / init_vel"); return MPK_EINVAL; }
        rp.traj_steps = st->traj_steps; rp.plan_steps = st->plan_steps; rp.done = st->done; rp.seg_len = st->seg_len;
        rp.done_out = st->done_out; rp.cond_pos = st->cond_pos; rp.cond_vel = st->cond_vel;
        rp.every = st->every; rp.max_planning_times = st->max_planning_times; rp.horizon = st->horizon;
    }
    RolloutDev rd;
    int r = fill_rollout(h, rc, &rd);
    if (r != MPK_OK) return r;
    if (rd.plant_type != MPK_PLANT_DOUBLE_INTEGRATOR) { set_error("mpk_episode_return integrates MPK_PLANT_DOUBLE_INTEGRATOR"); return MPK_EINVAL; }
    GateDev gd;
    if (gate) {
        r = fill_gate(h, gate, &gd);
        if (r != MPK_OK) return r;
    }
    MPK_ON_DEVICE(h->cfg.device);
    {
        const int fr = pending_ring_fault(h);
        if (fr != MPK_OK) return fr;
    }
    if (fused_phase_capable(h)) {
        // learned tau / delay: k_phase_fused without stores (no device reward for these families: include/mpk.h)
        if (reward != MPK_REWARD_NONE) { set_error("mpk_episode_return: a per-episode phase has no device reward"); return MPK_ENOTIMPL; }
        return phase_fused_common(h, params, init_pos, init_vel, init_time_shared, nullptr, nullptr, nullptr, rd, q, qd,
                                  st ? nullptr : n_steps, st ? &rp : nullptr, gate ? &gd : nullptr, ret, seg_out, B, stream);
    }
    if (!fused_capable(h)) { set_error("mpk_episode_return needs a shared phase with <= 16 contraction columns and DoF, or a learned phase with <= 8 columns"); return MPK_ENOTIMPL; }
    const Tuning tune = effective_tuning(h);
    if (h->cfg.mp_type == MPK_MP_DMP && h->cfg.dmp_first_sample == MPK_DMP_FIRST_IS_STEP) {
        set_error("mpk_episode_return: MPK_DMP_FIRST_IS_STEP handles take the separate launches");
        return MPK_ENOTIMPL;
    }
    const bool resp = h->cfg.mp_type == MPK_MP_DMP;
    SharedTables stt;
    r = get_shared(h, (float)init_time_shared, stream, &stt, resp);
    if (r != MPK_OK) return r;
    const char* name = "";
    r = launch_episode_return(resp ? h->dev_resp : h->dev, stt, params, init_pos, init_vel, rd, q, qd, st ? nullptr : n_steps,
                              st ? &rp : nullptr, reward, goal, step0, steps_before_reward, agg, ret, seg_out, B, h->num_cu, stream,
                              &name, tune, gate ? &gd : nullptr);
    if (r == MPK_OK) {
        h->kernel_name_buf = name;
        if (resp) {
            const size_t at = h->kernel_name_buf.find("prodmp");
            if (at != std::string::npos) h->kernel_name_buf.replace(at, 6, "dmp_resp");
        }
        h->last_kernel = h->kernel_name_buf.c_str();
    }
    return r;
}

int mpk_episode_return(mpk_handle hh, const float* params, const float* init_pos, const float* init_vel,
                       double init_time_shared, const mpk_rollout_cfg* rc, double* q, double* qd, const mpk_replan_state* st,
                       const int32_t* n_steps, int32_t* seg_out, int32_t reward, const double* goal, const int32_t* step0,
                       int32_t steps_before_reward, int32_t agg, double* ret, int32_t B, void* stream) {
    return mpk_episode_return_gated(hh, params, init_pos, init_vel, init_time_shared, rc, q, qd, st, nullptr, n_steps, seg_out, reward,
                                    goal, step0, steps_before_reward, agg, ret, B, stream);
}

int mpk_reward_aggregate(mpk_handle hh, const double* rewards, const int32_t* seg_len, int32_t agg, double* out, int32_t B,
                         int32_t T, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (B < 0 || T < 0) { set_error("B and T must be >= 0"); return MPK_EINVAL; }
    if (agg < MPK_AGG_SUM || agg > MPK_AGG_LAST) { set_error("unknown reward aggregation"); return MPK_EINVAL; }
    if (B == 0) return MPK_OK;
    if (!rewards || !seg_len || !out) { set_error("NULL buffer"); return MPK_EINVAL; }
    MPK_ON_DEVICE(h->cfg.device);
    return launch_reward_aggregate(rewards, seg_len, agg, out, B, T, stream);
}

int mpk_pd_rollout(mpk_handle hh, const mpk_rollout_cfg* rc, const float* des_pos, const float* des_vel, double* q,
                   double* qd, const int32_t* n_steps, float* actions, int32_t B, int32_t T, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!des_pos || !des_vel || !q || !qd) { set_error("NULL buffer"); return MPK_EINVAL; }
    if (B < 0 || T < 0) { set_error("B and T must be >= 0"); return MPK_EINVAL; }
    RolloutDev rd;
    int r = fill_rollout(h, rc, &rd);
    if (r != MPK_OK) return r;
    if (B == 0 || T == 0 || h->dev.D == 0) return MPK_OK;
    MPK_ON_DEVICE(h->cfg.device);
    {
        const int fr = pending_ring_fault(h);       // (k_pd_rollout_pipe reports through the handle's fault word)
        if (fr != MPK_OK) return fr;
    }
    return launch_pd_rollout(rd, h->dev.D, des_pos, des_vel, q, qd, n_steps, actions, B, T, stream, effective_tuning(h), h->d_fault);
}

int mpk_reacher_rollout(mpk_handle hh, const mpk_rollout_cfg* rc, const float* des_pos, const float* des_vel,
                        double* q, double* qd, const int32_t* n_steps, const int32_t* step0, const double* goal,
                        int32_t steps_before_reward, float* actions, double* rewards, int32_t B, int32_t T,
                        void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (B < 0 || T < 0) { set_error("B and T must be >= 0"); return MPK_EINVAL; }
    RolloutDev rd;
    int r = fill_rollout(h, rc, &rd);
    if (r != MPK_OK) return r;
    if (rd.plant_type != MPK_PLANT_DOUBLE_INTEGRATOR) {
        set_error("the reacher reward needs the torque double-integrator plant");
        return MPK_EINVAL;
    }
    if (B == 0 || T == 0 || h->dev.D == 0) return MPK_OK;
    if (!des_pos || !des_vel || !q || !qd || !goal || !rewards) { set_error("NULL buffer"); return MPK_EINVAL; }
    MPK_ON_DEVICE(h->cfg.device);
    {
        const int fr = pending_ring_fault(h);       // (its helper-wave variant reports through the handle's fault word, like k_traj_ring)
        if (fr != MPK_OK) return fr;
    }
    return launch_reacher_rollout(rd, h->dev.D, des_pos, des_vel, q, qd, n_steps, step0, goal, steps_before_reward,
                                  actions, rewards, B, T, stream, effective_tuning(h), h->d_fault);
}

int mpk_episode_reset(mpk_handle hh, const double* init_q, const double* init_qd, double* q, double* qd,
                      float* cond_pos, float* cond_vel, int32_t* traj_steps, int32_t* plan_steps, uint8_t* done,
                      int32_t B, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (B < 0) { set_error("B must be >= 0"); return MPK_EINVAL; }
    if (B == 0) return MPK_OK;
    if (!q || !qd || !traj_steps || !plan_steps || !done) { set_error("NULL buffer"); return MPK_EINVAL; }
    if ((cond_pos == nullptr) != (cond_vel == nullptr)) { set_error("cond_pos and cond_vel go together"); return MPK_EINVAL; }
    MPK_ON_DEVICE(h->cfg.device);
    return launch_episode_reset(init_q, init_qd, q, qd, cond_pos, cond_vel, traj_steps, plan_steps, done, B, h->dev.D,
                                stream);
}

int mpk_replan_advance(mpk_handle hh, int32_t* traj_steps, int32_t* plan_steps, int32_t* seg_len, uint8_t* done,
                       int32_t every, int32_t max_planning_times, int32_t horizon, int32_t T, int32_t B,
                       void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!traj_steps || !plan_steps || !seg_len || !done) { set_error("NULL buffer"); return MPK_EINVAL; }
    if (every < 1 || horizon < 1 || T < 1 || B < 0) { set_error("every, horizon, T must be >= 1"); return MPK_EINVAL; }
    if (B == 0) return MPK_OK;
    MPK_ON_DEVICE(h->cfg.device);
    return launch_replan_advance(traj_steps, plan_steps, seg_len, done, every, max_planning_times, horizon, T, B,
                                 stream);
}

int mpk_gate_flags(mpk_handle hh, const uint8_t* valid, const uint8_t* was_done, const uint8_t* done, uint8_t* terminated,
                   uint8_t* truncated, int32_t B, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (B < 0) { set_error("B must be >= 0"); return MPK_EINVAL; }
    if (B == 0) return MPK_OK;
    if (!valid || !done || !terminated || !truncated) { set_error("NULL buffer"); return MPK_EINVAL; }
    MPK_ON_DEVICE(h->cfg.device);
    return launch_gate_flags(valid, was_done, done, terminated, truncated, B, stream);
}

int mpk_condition_gather(mpk_handle hh, const float* pos, const float* vel, const int32_t* seg_len, float* cond_pos,
                         float* cond_vel, int32_t B, int32_t T, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (B < 0 || T < 1) { set_error("B must be >= 0 and T >= 1"); return MPK_EINVAL; }
    if (B == 0 || h->dev.D == 0) return MPK_OK;
    if (!pos || !vel || !seg_len || !cond_pos || !cond_vel) { set_error("NULL buffer"); return MPK_EINVAL; }
    MPK_ON_DEVICE(h->cfg.device);
    return launch_condition_gather(pos, vel, seg_len, cond_pos, cond_vel, B, T, h->dev.D, stream);
}

int mpk_traj_validity(mpk_handle hh, const float* pos, const float* params, const double* pos_low,
                      const double* pos_high, int32_t check_tau_delay, const double tau_bound[2],
                      const double delay_bound[2], uint8_t* valid, int32_t B, int32_t T, void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!pos || !pos_low || !pos_high || !valid) { set_error("NULL buffer"); return MPK_EINVAL; }
    if (check_tau_delay && (!params || !tau_bound || !delay_bound)) { set_error("tau/delay check needs params and bounds"); return MPK_EINVAL; }
    if (h->dev.D > kMaxDofArgs) { set_error("num_dof too large"); return MPK_EINVAL; }
    if (B <= 0 || T <= 0) return MPK_OK;
    MPK_ON_DEVICE(h->cfg.device);
    return launch_validity(pos, params, h->dev.P, h->dev.D, pos_low, pos_high, check_tau_delay, tau_bound,
                           delay_bound, valid, nullptr, B, T, stream);
}

int mpk_traj_validity_penalty(mpk_handle hh, const float* pos, const float* params, const double* pos_low,
                              const double* pos_high, int32_t check_tau_delay, const double tau_bound[2],
                              const double delay_bound[2], uint8_t* valid, double* penalty, int32_t B, int32_t T,
                              void* stream) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!pos || !pos_low || !pos_high || !valid || !penalty) { set_error("NULL buffer"); return MPK_EINVAL; }
    if (check_tau_delay && (!params || !tau_bound || !delay_bound)) { set_error("tau/delay check needs params and bounds"); return MPK_EINVAL; }
    if (h->dev.D > kMaxDofArgs) { set_error("num_dof too large"); return MPK_EINVAL; }
    if (B <= 0 || T <= 0) return MPK_OK;
    MPK_ON_DEVICE(h->cfg.device);
    return launch_validity(pos, params, h->dev.P, h->dev.D, pos_low, pos_high, check_tau_delay, tau_bound,
                           delay_bound, valid, penalty, B, T, stream);
}

int mpk_prodmp_tables(mpk_handle hh, double* y1, double* y2, double* dy1, double* dy2, double* pos_basis,
                      double* vel_basis, double* scale) {
    if (!hh) { set_error("NULL handle"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (h->cfg.basis_type != MPK_BASIS_PRODMP) { set_error("not a prodmp handle"); return MPK_EINVAL; }
    const HostTables& t = h->tab;
    auto cp = [](double* dst, const std::vector<double>& src) { if (dst) std::memcpy(dst, src.data(), src.size() * sizeof(double)); };
    cp(y1, t.y1); cp(y2, t.y2); cp(dy1, t.dy1); cp(dy2, t.dy2);
    cp(pos_basis, t.pos_basis); cp(vel_basis, t.vel_basis); cp(scale, t.scale);
    return t.n_pc;
}

int mpk_prodmp_indices(mpk_handle hh, double init_time, int32_t* idx, int32_t* idx_init, void* stream) {
    if (!hh || !idx || !idx_init) { set_error("NULL argument"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (h->cfg.mp_type != MPK_MP_PRODMP) { set_error("not a prodmp handle"); return MPK_EINVAL; }
    if (h->cfg.learn_tau || h->cfg.learn_delay) { set_error("indices are per-episode when tau/delay are learned"); return MPK_EINVAL; }
    if (!mfma_capable(h)) { set_error("configuration exceeds the shared-table kernel limits"); return MPK_EINVAL; }
    MPK_ON_DEVICE(h->cfg.device);
    const int T = h->dev.T;
    if (h->idx_cap < T + 1) {
        if (h->d_idx) (void)hipFree(h->d_idx);
        MPK_HIP(hipMalloc((void**)&h->d_idx, sizeof(int32_t) * (T + 1)));
        h->idx_cap = T + 1;
    }
    int TS = 0, n_out = 0;
    const size_t nf = shared_tables_floats(h->dev, &TS, &n_out);
    SharedTables st;
    MPK_HIP(hipMalloc((void**)&st.A, nf * sizeof(float)));
    MPK_HIP(hipMalloc((void**)&st.aux, (size_t)TS * sizeof(float)));
    st.TS = TS; st.n_out = n_out;
    int rc = launch_build_shared(h->dev, (float)init_time, st, h->d_idx, h->d_flag, stream);
    if (rc == MPK_OK) {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e == hipSuccess) e = hipMemcpy(idx, h->d_idx, sizeof(int32_t) * T, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(idx_init, h->d_idx + T, sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = MPK_EHIP; }
    }
    (void)hipFree(st.A); (void)hipFree(st.aux);
    return rc;
}

int mpk_scaled_basis(mpk_handle hh, const float* times, int32_t n, float* basis, void* stream) {
    if (!hh || !times || !basis) { set_error("NULL argument"); return MPK_EINVAL; }
    if (n < 0) { set_error("n must be >= 0"); return MPK_EINVAL; }
    if (n == 0) return MPK_OK;
    Handle* h = reinterpret_cast<Handle*>(hh);
    MPK_ON_DEVICE(h->cfg.device);
    const int K = h->cfg.mp_type == MPK_MP_PRODMP ? h->cfg.num_basis + 1 : h->cfg.num_basis;
    float *d_t = nullptr, *d_o = nullptr;
    MPK_HIP(hipMalloc((void**)&d_t, sizeof(float) * n));
    if (hipMalloc((void**)&d_o, sizeof(float) * (size_t)n * K) != hipSuccess) { (void)hipFree(d_t); set_error("hipMalloc failed"); return MPK_EHIP; }
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(d_t, times, sizeof(float) * n, hipMemcpyHostToDevice, s);
    int rc = e == hipSuccess ? launch_scaled_basis(h->dev, d_t, n, d_o, stream) : MPK_EHIP;
    if (rc == MPK_OK) {
        e = hipMemcpyAsync(basis, d_o, sizeof(float) * (size_t)n * K, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = MPK_EHIP; }
    }
    (void)hipFree(d_t); (void)hipFree(d_o);
    return rc;
}

int mpk_selftest_division(mpk_handle hh, float divisor, uint32_t first_bits, uint64_t count, uint64_t* mismatches,
                          void* stream) {
    if (!hh || !mismatches) { set_error("NULL argument"); return MPK_EINVAL; }
    Handle* h = reinterpret_cast<Handle*>(hh);
    if (!(divisor > 0.0f) || !std::isfinite(divisor)) { set_error("divisor must be positive and finite"); return MPK_EINVAL; }
    MPK_ON_DEVICE(h->cfg.device);
    unsigned long long* d_bad = nullptr;
    MPK_HIP(hipMalloc((void**)&d_bad, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), (hipStream_t)stream);
    int rc = e == hipSuccess ? launch_div_sweep(divisor, first_bits, count, d_bad, stream) : MPK_EHIP;
    unsigned long long bad = 0;
    if (rc == MPK_OK) {
        e = hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = MPK_EHIP; }
    }
    (void)hipFree(d_bad);
    *mismatches = bad;
    return rc;
}

int mpk_host_prodmp_tables(const mpk_config* cfg, double* y1, double* y2, double* dy1, double* dy2,
                           double* pos_basis, double* vel_basis, double* scale, float* scaled_dt) {
    if (!cfg) { set_error("NULL argument"); return MPK_EINVAL; }
    int rc = check_cfg(*cfg);
    if (rc != MPK_OK) return rc;
    if (cfg->basis_type != MPK_BASIS_PRODMP) { set_error("not a prodmp configuration"); return MPK_EINVAL; }
    mpk_config q = *cfg;
    quantise(q);
    HostTables t;
    build_prodmp(q, t);
    auto cp = [](double* dst, const std::vector<double>& src) { if (dst) std::memcpy(dst, src.data(), src.size() * sizeof(double)); };
    cp(y1, t.y1); cp(y2, t.y2); cp(dy1, t.dy1); cp(dy2, t.dy2);
    cp(pos_basis, t.pos_basis); cp(vel_basis, t.vel_basis); cp(scale, t.scale);
    if (scaled_dt) *scaled_dt = t.scaled_dt;
    return t.n_pc;
}

int mpk_host_rbf(const mpk_config* cfg, double* centers, double* bw) {
    if (!cfg) { set_error("NULL argument"); return MPK_EINVAL; }
    int rc = check_cfg(*cfg);
    if (rc != MPK_OK) return rc;
    mpk_config q = *cfg;
    quantise(q);
    HostTables t;
    build_rbf(q, t);
    if (centers) std::memcpy(centers, t.centers.data(), t.centers.size() * sizeof(double));
    if (bw) std::memcpy(bw, t.bw.data(), t.bw.size() * sizeof(double));
    return t.n_total;
}

int mpk_host_times(double duration, double dt, float* times, int32_t cap) {
    if (!(duration > 0.0) || !(dt > 0.0)) { set_error("dt and duration must be > 0"); return MPK_EINVAL; }
    const int T = steps_for(duration, dt);
    if (T < 1) { set_error("duration/dt gives no time steps"); return MPK_EINVAL; }
    if (times) {
        if (cap < T) { set_error("times buffer too small"); return MPK_EINVAL; }
        const std::vector<float> t = build_times(duration, T);
        std::memcpy(times, t.data(), sizeof(float) * T);
    }
    return T;
}

int mpk_host_num_params(const mpk_config* cfg) {
    if (!cfg) { set_error("NULL argument"); return MPK_EINVAL; }
    int rc = check_cfg(*cfg);
    if (rc != MPK_OK) return rc;
    return (cfg->learn_tau ? 1 : 0) + (cfg->learn_delay ? 1 : 0) + cfg->num_dof * local_per_dof(*cfg);
}

// ---- RCCL all-gather (the one exchange step of the path) ----------------------------------------------------------
// librccl is bound with dlopen on first use: libmpk.so itself carries no dependency on it, and the copy the process
// already has (torch ships one with the same SONAME) is the one that gets used.
}  // extern "C"
namespace {
struct RcclId { char internal[MPK_COMM_ID_BYTES]; };     // = ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.why = std::string("librccl.so.1 not found: ") + dlerror(); return; }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
            r.why = "librccl.so.1 lacks the nccl* entry points";
            r.lib = nullptr;
        }
    });
    return r;
}
constexpr int kNcclFloat = 7;   // rccl.h ncclDataType_t: ncclFloat32 = 7
struct Comm { void* nccl = nullptr; int rank = 0, world = 1, device = 0; };
int rccl_fail(const char* what, int rc) {
    set_error(std::string(what) + ": " + rccl().GetErrorString(rc));
    return MPK_ECOMM;
}
}  // namespace
extern "C" {

int mpk_comm_unique_id(uint8_t* id) {
    if (!id) { set_error("NULL argument"); return MPK_EINVAL; }
    Rccl& r = rccl();
    if (!r.lib) { set_error(r.why); return MPK_ECOMM; }
    RcclId u;
    const int rc = r.GetUniqueId(&u);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id, u.internal, MPK_COMM_ID_BYTES);
    return MPK_OK;
}

int mpk_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device, mpk_comm* out) {
    if (!id || !out) { set_error("NULL argument"); return MPK_EINVAL; }
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { set_error("need 0 <= rank < world"); return MPK_EINVAL; }
    Rccl& r = rccl();
    if (!r.lib) { set_error(r.why); return MPK_ECOMM; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void)hipGetLastError(); set_error("no HIP device"); return MPK_ENODEV; }
    if (device < 0 || device >= n) { set_error("device ordinal out of range"); return MPK_EINVAL; }
    MPK_ON_DEVICE(device);
    RcclId u;
    std::memcpy(u.internal, id, MPK_COMM_ID_BYTES);
    Comm* c = new Comm;
    c->rank = rank; c->world = world; c->device = device;
    const int rc = r.CommInitRank(&c->nccl, world, u, rank);
    if (rc != 0) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    *out = reinterpret_cast<mpk_comm>(c);
    return MPK_OK;
}

int mpk_comm_rank(mpk_comm cc) {
    if (!cc) { set_error("NULL communicator"); return MPK_EINVAL; }
    return reinterpret_cast<Comm*>(cc)->rank;
}

int mpk_comm_world(mpk_comm cc) {
    if (!cc) { set_error("NULL communicator"); return MPK_EINVAL; }
    return reinterpret_cast<Comm*>(cc)->world;
}

int mpk_allgather(mpk_comm cc, const float* send, float* recv, int64_t count, void* stream) {
    if (!cc) { set_error("NULL communicator"); return MPK_EINVAL; }
    if (count < 0) { set_error("count must be >= 0"); return MPK_EINVAL; }
    if (count == 0) return MPK_OK;
    if (!send || !recv) { set_error("NULL buffer"); return MPK_EINVAL; }
    Comm* c = reinterpret_cast<Comm*>(cc);
    MPK_ON_DEVICE(c->device);
    const int rc = rccl().AllGather(send, recv, (size_t)count, kNcclFloat, c->nccl, (hipStream_t)stream);
    if (rc != 0) return rccl_fail("ncclAllGather", rc);
    return MPK_OK;
}

void mpk_comm_destroy(mpk_comm cc) {
    if (!cc) return;
    Comm* c = reinterpret_cast<Comm*>(cc);
    DeviceGuard guard(c->device);
    if (c->nccl) (void)rccl().CommDestroy(c->nccl);
    delete c;
}

const char* mpk_last_kernel(mpk_handle hh) {
    if (!hh) return "";
    return reinterpret_cast<Handle*>(hh)->last_kernel;
}

}  // extern "C"
