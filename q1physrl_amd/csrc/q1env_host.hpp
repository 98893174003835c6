// q1env_host.hpp - host-side internals shared by the translation units of libq1env.so (NOT part of the C ABI: everything here has
// hidden visibility; the exported surface is include/q1env.h and nothing else).
//
//   q1env_core.hip      env kernels (step / autoreset / rollout / reset / observe / decode / phys.apply), the handle, the core ABI
//   q1env_policy.hip    policy-side glue kernels (action sampling, fused sampler tick, episode statistics, GAE, PPO loss gradient)
//                       and the launchers of the matrix-core forward (q1policy.hpp)
//   q1env_server.hip    the resident tick server (q1server.hpp)
//   q1env_resident.hip  the resident sampler (q1resident.hpp)
//   q1env_learner.hip   the native PPO learner step (q1learner.hpp, q1ppo_loss.hpp)
//   q1env_diag.hip      timers, PMC traffic calibration, division self-test
//
// Kernels live in the translation unit that launches them (no relocatable device code: each TU carries its own code object).
#pragma once
#include "q1env_device.hpp"
#pragma GCC visibility push(default)
#include "../../include/q1env.h"
#pragma GCC visibility pop

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define Q1_HIDDEN __attribute__((visibility("hidden")))

// thread-local last-error message (q1env_last_error) + status code pass-through
Q1_HIDDEN int q1_fail(int code, const std::string& msg);
Q1_HIDDEN const char* q1_last_error_cstr();
static inline int fail(int code, const std::string& msg) { return q1_fail(code, msg); }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(Q1ENV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

// Makes the handle's device current for the duration of an entry point and restores the caller's device afterwards
// (a host framework such as torch tracks the thread's current device itself; the library must not change it under it).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

using q1::Params;
using q1::StatePtrs;

struct q1env {
    q1env_config cfg{};
    Params p{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void* arena = nullptr;            // one allocation holding the whole SoA state
    StatePtrs st{};
    // staging for the *_host entry points (grown on demand)
    void* snap = nullptr;             // q1env_snapshot_state: a second arena holding a copy of the whole SoA state
    void* stage = nullptr;
    size_t stage_bytes = 0;
    void* pin = nullptr;              // pinned (page-locked) host staging of the *_host entry points: one DMA each way instead of
    size_t pin_bytes = 0;             // one staged pageable copy per array
    uint64_t tick_count = 0;          // ticks since create: the counter of the counter-based RNG
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cus = 256;                // compute units of the device (MI355X in SPX mode: 256)
    int server_blocks_per_cu[3] = {-1, -1, -1};   // occupancy of the resident tick server at 1/2/4 envs per lane (queried once)
    int pair_blocks_per_cu[3] = {-1, -1, -1};     // ... and of the server + driver pair kernel, per shape (PAIR_SHAPES)
    bool resident_attr_set = false;   // the resident sampler's dynamic-LDS attribute
    bool mlp_attr_set = false;        // dynamic-LDS attribute of the policy kernels (a per-device setting: kept per handle)
    bool learner_attr_set = false;    // ... and of the native learner's forward / backward kernels
    bool plearner_attr_set = false;   // ... and of the persistent learner (q1learner_persist.hpp)
    float pi_upscale = 0.0f, value_downscale = 0.0f;   // the learner's float16 loss scales (q1env_learner_set_loss_scale; 0 = the defaults)
    int plearner_mode = 0;            // exchange mode of the persistent learner (q1env_learner_set_exchange_mode): 0 auto, 1 agent scope, 2 census made to fail
    int learner_step_mode = 0;        // q1env_learner_sgd_step's kernel sequence (q1env_learner_set_step_mode): 0 auto, 1 four launches, 2 fused forward + backward, 3 ... + dW1 products
    int wgrad_variant = 0;            // (measurement only) step mode 3 with an earlier weight-gradient kernel: 1 column quarters, 2 round 4's; 0 = the shared-operand kernel
    int plearner_prof = -1;           // >= 0: the profiling instantiation stamps wave (value >> 3) of workgroup (value & 7) of the policy group
    // cached hipGraphs of step_many, keyed by (ticks, formats, pointers); a handful of entries, oldest evicted
    struct GraphEntry { std::vector<uint64_t> key; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    hipStream_t cap_stream = nullptr;  // private stream used only to CAPTURE (the null stream cannot be captured)
    // completion signal (include/q1env.h "completion signal"): sig_host[0] = start stamp, [1] = end stamp, [2] = sequence number, written
    // by the kernels themselves into host-coherent pinned memory; ticket_dev = device counter of retired waves; sig_seq = last requested
    volatile uint64_t* sig_host = nullptr;
    uint64_t* sig_dev = nullptr;
    uint32_t* ticket_dev = nullptr;
    uint64_t sig_seq = 0;
    double wall_clock_hz = 0.0;
    // host-direct block of the small-batch *_host paths: host-coherent pinned memory the kernels read their inputs from and write their
    // results to (no copy commands, no synchronisation: the completion signal says when the results are there)
    char* direct_host = nullptr;
    char* direct_dev = nullptr;
    size_t direct_bytes = 0;
};
Q1_HIDDEN int ensure_signal(q1env* h);
Q1_HIDDEN int ensure_direct(q1env* h, size_t bytes);
Q1_HIDDEN int signal_wait(q1env* h, double timeout_s);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline dim3 grid_for(int n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

// 64-lane workgroups at every batch size: small batches are latency bound and 64-lane workgroups spread 64 k envs over all 256 CUs
// (1024 waves); for the bandwidth-bound batches round 3 re-measured 64 against 256 lanes three times each (tools/exp_step_large.py,
// profiles/r3_step_large.txt): 512 k envs 15.9 vs 17.0 us per tick, 1 M 33.5-35.7 vs 34.7-37.8, 2 M and 4 M equal within the 5 %
// run-to-run spread - the dispatcher is not the limit, and smaller workgroups retire (and refill their SIMD slots) sooner.
// Q1ENV_BLOCK = 64 / 128 / 256 overrides (measurement knob).
static inline int block_for(int n) {
    static const int forced = [] { const char* e = getenv("Q1ENV_BLOCK"); return e ? atoi(e) : 0; }();
    (void)n;
    if (forced == 64 || forced == 128 || forced == 256) return forced;
    return 64;
}

Q1_HIDDEN int ensure_stage(q1env* h, size_t bytes);
Q1_HIDDEN int ensure_pin(q1env* h, size_t bytes);
Q1_HIDDEN void carve_into(void* arena, size_t n, q1::StatePtrs& st);
Q1_HIDDEN size_t arena_bytes(size_t n);

static inline int check_act(const q1env* h, int fmt, const void* a, const void* b, bool allow_random) {
    if (fmt == Q1ENV_ACT_RANDOM) return allow_random ? 0 : fail(Q1ENV_ERR_INVALID_ARG, "Q1ENV_ACT_RANDOM is rollout-only");
    if (fmt < 0 || fmt > 2) return fail(Q1ENV_ERR_INVALID_ARG, "unknown action_format");
    if (!a) return fail(Q1ENV_ERR_INVALID_ARG, "act_a is NULL");
    if (fmt == Q1ENV_ACT_PACKED && h->p.yaw_mode && !b) return fail(Q1ENV_ERR_INVALID_ARG, "packed actions need act_b (mouse)");
    return 0;
}

static inline size_t act_bytes_a(const q1env* h, int fmt) {
    const size_t n = (size_t)h->p.n;
    if (fmt == Q1ENV_ACT_F64_ROWS) return n * h->p.act_width * 8;
    if (fmt == Q1ENV_ACT_F32_ROWS) return n * h->p.act_width * 4;
    return n;
}

// Width of one row of policy-network outputs (Q1PhysActionDist.required_model_output_shape, action_dist.py:236-241): two logits per
// key, then (mean, log_std) of the continuous mouse or the 2S+1 logits of the discrete one.
static inline int policy_row_width(const Params& p) {
    return 2 * p.num_keys + (p.yaw_mode == 1 ? 2 : (p.yaw_mode == 2 ? 2 * (int)p.yaw_steps + 1 : 0));
}

// The default action/episode structure (4 keys, continuous mouse, jump key, no hover, y reward) runs the SPEC kernels.
// The SPEC tick also takes the square root of |wish_vel|^2 without the scaling of tiny inputs (q1env_device.hpp physics_core
// NORMAL): the move maxima must be 0 or of ordinary magnitude (the reference's are 800 and 1060).
static inline bool move_max_ordinary(double m) {
    const double a = m < 0 ? -m : m;
    return a == 0.0 || (a >= 0x1p-200 && a <= 0x1p+200);
}
// The SPEC tick divides by its two run-time constants (time_limit, yaw_den) with ONE Markstein correction step (q1env_device.hpp
// div_const1): valid when the correctly rounded reciprocal's relative error is at most 2^-54.  c * y - 1 is exact in one fma (it is a
// multiple of 2^-105 or so below 2^-52 in magnitude).  10 (time_limit of get_default and params.yml; action_range 10 of params.yml)
// sits exactly on the bound, float32(10.08) (get_default's action_range) at 0.68 of it; a Config whose constants fail runs the
// generic kernels (two steps).
static inline bool div_one_step_ok(double c, double y) {
    return c > 0.0 && std::fabs(std::fma(c, y, -1.0)) <= 0x1p-54;
}
static inline bool is_spec(const Params& p) {
    return p.num_keys == 4 && p.yaw_mode == 1 && p.jump_mode == 1 && !p.hover && !p.speed_reward &&
           move_max_ordinary(p.fmove_max) && move_max_ordinary(p.smove_max) && move_max_ordinary(p.smooth_scale) &&
           div_one_step_ok(p.time_limit, p.time_limit_rcp) && div_one_step_ok(p.yaw_den, p.yaw_den_rcp);
}
