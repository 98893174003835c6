// q1env_server.hip - the resident tick server of libq1env.so (q1env_step_persistent_*; device code in q1server.hpp).
#include "q1env_host.hpp"
#include "q1server.hpp"

using namespace q1;

extern "C" {

// ---- persistent tick server -----------------------------------------------------------------------------------------------
// Poll pacing of the tick server (see Backoff in q1server.hpp); Q1ENV_SERVER_BACKOFF="first_server,first_driver,between" overrides
// the defaults (measurement knob).
static Backoff server_backoff() {
    static const Backoff bo = [] {
        Backoff b{0, 0, 0};
        if (const char* e = getenv("Q1ENV_SERVER_BACKOFF")) (void)sscanf(e, "%d,%d,%d", &b.first_server, &b.first_driver, &b.between);
        auto clamp = [](int v) { return v < 0 ? 0 : (v > 4096 ? 4096 : v); };
        b.first_server = clamp(b.first_server); b.first_driver = clamp(b.first_driver); b.between = clamp(b.between);
        return b;
    }();
    return bo;
}

// Envs per lane of the resident grid (E in {1, 2, 4}, index e = log2 E): the smallest that makes the whole grid resident (at 8 the
// server needs 416 VGPRs and spills: one wave per SIMD, no more envs resident than at 4).
// A kernel instance per (SPEC, E); the switch keeps every launch a direct call.
#define Q1_FOR_E(e_idx, CALL)          \
    switch (e_idx) {                   \
        case 0: { CALL(1); } break;    \
        case 1: { CALL(2); } break;    \
        default: { CALL(4); } break;   \
    }

extern "C++" {
template <int E> static const void* server_fn(bool spec) { return spec ? (const void*)tick_server_kernel<true, E> : (const void*)tick_server_kernel<false, E>; }
}

// q1env_step_persistent_pair: ES = sub-batches of 64 envs per (server wave, driver wave) workgroup (q1server.hpp, tick_pair_lds_kernel).
// The smallest ES whose grid is resident wins (Q1ENV_SERVER_SHAPE="<ES>" forces one: measurement knob).
static constexpr int MAX_PAIR_ES = 3;
#define Q1_FOR_PAIR_ES(es, CALL)       \
    switch (es) {                      \
        case 1: { CALL(1); } break;    \
        case 2: { CALL(2); } break;    \
        default: { CALL(3); } break;   \
    }

static int server_blocks_per_cu_of(q1env_t* h, int e_idx, int* out) {
    int& slot = h->server_blocks_per_cu[e_idx];
    if (slot < 0) {
        const void* fn = nullptr;
        const bool spec = is_spec(h->p);
#define Q1_FN(E) fn = server_fn<E>(spec)
        Q1_FOR_E(e_idx, Q1_FN)
#undef Q1_FN
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0));
        slot = per_cu;
    }
    *out = slot;
    return Q1ENV_OK;
}

static int pair_blocks_per_cu_of(q1env_t* h, int es, int* out) {
    int& slot = h->pair_blocks_per_cu[es - 1];
    if (slot < 0) {
        const void* fn = nullptr;
        const bool spec = is_spec(h->p);
#define Q1_FN(ES) fn = spec ? (const void*)tick_pair_lds_kernel<true, ES> : (const void*)tick_pair_lds_kernel<false, ES>
        Q1_FOR_PAIR_ES(es, Q1_FN)
#undef Q1_FN
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 128, q1pair::lds_bytes(es)));
        slot = per_cu;
    }
    *out = slot;
    return Q1ENV_OK;
}

// Server on its own stream (an external producer next to it): the whole grid must be resident at once - a wave that is not
// scheduled never polls - AND leave room for the producer's waves on every SIMD (a server that fills the register file starves
// the producer it waits for: both would only time out).  start and drive call this with the same handle, so they agree on E.
static int server_envs_per_lane(q1env_t* h, const char* who, int* e_idx_out) {
    long best = 0;
    for (int e = 0; e < 3; ++e) {
        int per_cu = 0;
        if (int rc = server_blocks_per_cu_of(h, e, &per_cu)) return rc;
        const long max_envs = (long)h->num_cus * (per_cu > 4 ? per_cu - 4 : 0) * 64 * (1L << e);
        if ((long)h->p.n <= max_envs) { *e_idx_out = e; return Q1ENV_OK; }
        if (max_envs > best) best = max_envs;
    }
    return fail(Q1ENV_ERR_INVALID_ARG, std::string(who) + ": too many envs for one resident grid next to its producer (" +
                                       std::to_string(best) + " at most on this device)");
}

int q1env_step_persistent_start(q1env_t* h, int ticks, uint32_t tag0, const uint64_t* mailbox_dev, uint64_t* results_dev,
                                float* obs_final_dev, uint64_t seed, int auto_reset, uint32_t* status_dev, double timeout_s) {
    if (!h || !mailbox_dev || !results_dev || !status_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_start: null argument");
    if (ticks <= 0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_start: ticks must be > 0");
    if (!(timeout_s > 0.0) || timeout_s > 30.0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_start: timeout_s must be in (0, 30]");
    if (h->p.yaw_mode == 2 && h->p.yaw_steps > 8388608.0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_start: step index does not fit the granule");
    DeviceGuard guard(h->device);
    int e_idx = 0;
    if (int rc = server_envs_per_lane(h, "q1env_step_persistent_start", &e_idx)) return rc;
    const unsigned per_block = 64u << e_idx;
    const dim3 g(((unsigned)h->p.n + per_block - 1u) / per_block), b(64);
    const uint64_t timeout_ticks = (uint64_t)(timeout_s * 1.0e8);          // wall_clock64: 100 MHz
#define Q1_LAUNCH(E)                                                                                                                    \
    if (is_spec(h->p))                                                                                                                  \
        hipLaunchKernelGGL((tick_server_kernel<true, E>), g, b, 0, h->stream, h->p, h->st, ticks, tag0, mailbox_dev, results_dev,       \
                           obs_final_dev, seed, h->tick_count, auto_reset, status_dev, timeout_ticks, server_backoff());                \
    else                                                                                                                                \
        hipLaunchKernelGGL((tick_server_kernel<false, E>), g, b, 0, h->stream, h->p, h->st, ticks, tag0, mailbox_dev, results_dev,      \
                           obs_final_dev, seed, h->tick_count, auto_reset, status_dev, timeout_ticks, server_backoff())
    Q1_FOR_E(e_idx, Q1_LAUNCH)
#undef Q1_LAUNCH
    HIP_TRY(hipGetLastError());
    h->tick_count += (uint64_t)ticks;
    return Q1ENV_OK;
}

int q1env_step_persistent_drive(q1env_t* h, void* producer_stream, int ticks, uint32_t tag0, const uint8_t* keys_dev,
                                const float* mouse_dev, uint64_t* mailbox_dev, const uint64_t* results_dev,
                                double* checksum_dev, uint32_t* status_dev, double timeout_s) {
    if (!h || !producer_stream || !keys_dev || !mouse_dev || !mailbox_dev || !results_dev || !status_dev)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_drive: null argument (the producer needs its own stream)");
    if ((hipStream_t)producer_stream == h->stream) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_drive: the producer must run on another stream than the server");
    if (ticks <= 0 || !(timeout_s > 0.0) || timeout_s > 30.0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_drive: bad ticks / timeout_s");
    DeviceGuard guard(h->device);
    int e_idx = 0;
    if (int rc = server_envs_per_lane(h, "q1env_step_persistent_drive", &e_idx)) return rc;
    const unsigned per_block = 64u << e_idx;
    const dim3 g(((unsigned)h->p.n + per_block - 1u) / per_block), b(64);
#define Q1_LAUNCH(E)                                                                                                                 \
    hipLaunchKernelGGL((tick_driver_kernel<E>), g, b, 0, (hipStream_t)producer_stream, h->p.n, ticks, tag0, keys_dev, mouse_dev,     \
                       mailbox_dev, results_dev, checksum_dev, status_dev, (uint64_t)(timeout_s * 1.0e8), server_backoff())
    Q1_FOR_E(e_idx, Q1_LAUNCH)
#undef Q1_LAUNCH
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_step_persistent_publish(q1env_t* h, void* producer_stream, uint32_t tag0, uint32_t tick, const uint8_t* keys_dev,
                                  const float* mouse_dev, uint64_t* mailbox_dev) {
    if (!h || !producer_stream || !keys_dev || !mailbox_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_publish: null argument");
    if (h->p.yaw_mode && !mouse_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_publish: mouse actions required");
    if ((hipStream_t)producer_stream == h->stream) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_publish: the producer must run on another stream than the server");
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(tick_publish_kernel, grid_for(h->p.n, 256), dim3(256), 0, (hipStream_t)producer_stream, h->p.n, tag0, tick, keys_dev,
                       mouse_dev, mailbox_dev);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_step_persistent_collect(q1env_t* h, void* producer_stream, uint32_t tag0, uint32_t tick, const uint64_t* results_dev,
                                  float* obs_dev, float* reward_dev, uint8_t* done_dev, uint8_t* zero_start_dev, uint32_t* status_dev,
                                  double timeout_s) {
    if (!h || !producer_stream || !results_dev || !obs_dev || !status_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_collect: null argument");
    if ((hipStream_t)producer_stream == h->stream) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_collect: the producer must run on another stream than the server");
    if (!(timeout_s > 0.0) || timeout_s > 30.0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_collect: timeout_s must be in (0, 30]");
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(tick_collect_kernel, dim3(((unsigned)h->p.n + 63u) / 64u), dim3(64), 0, (hipStream_t)producer_stream, h->p.n, tag0, tick,
                       results_dev, obs_dev, reward_dev, done_dev, zero_start_dev, status_dev, (uint64_t)(timeout_s * 1.0e8));
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_step_persistent_pair(q1env_t* h, int ticks, uint32_t tag0, const uint8_t* keys_dev, const float* mouse_dev,
                               uint64_t* mailbox_dev, uint64_t* results_dev, float* obs_final_dev, uint64_t seed, int auto_reset,
                               double* checksum_dev, uint32_t* status_dev, double timeout_s) {
    if (!h || !keys_dev || !mouse_dev || !mailbox_dev || !results_dev || !status_dev)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_pair: null argument");
    if (ticks <= 0 || !(timeout_s > 0.0) || timeout_s > 30.0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_pair: bad ticks / timeout_s");
    DeviceGuard guard(h->device);
    // one dispatch of (server wave, driver wave) workgroups: every wait in it is between waves of one workgroup, but the grid is
    // still required to be resident (a workgroup that waits for a slot holds its envs' ticks back, and the launch's time with them)
    int es = 0;
    long best = 0;
    const char* forced = getenv("Q1ENV_SERVER_SHAPE");
    for (int k = 1; k <= MAX_PAIR_ES && !es; ++k) {
        if (forced && forced[0] >= '1' && forced[0] <= '0' + MAX_PAIR_ES && k != forced[0] - '0') continue;
        int per_cu = 0;
        if (int rc = pair_blocks_per_cu_of(h, k, &per_cu)) return rc;
        const long max_envs = (long)h->num_cus * per_cu * 64 * k;
        if (max_envs > best) best = max_envs;
        if ((long)h->p.n <= max_envs) es = k;
    }
    if (!es)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_persistent_pair: too many envs for one resident grid (" + std::to_string(best) +
                                           " at most on this device)");
    const uint64_t timeout_ticks = (uint64_t)(timeout_s * 1.0e8);
    const unsigned per_block = 64u * (unsigned)es;
    const dim3 g(((unsigned)h->p.n + per_block - 1u) / per_block), b(128);
    const bool t_start = (auto_reset & Q1ENV_TIMER_START) != 0, t_stop = (auto_reset & Q1ENV_TIMER_STOP) != 0;
    auto_reset &= 1;
    if (t_start) HIP_TRY(hipEventRecord(h->ev0, h->stream));
#define Q1_LAUNCH(ES)                                                                                                                     \
    if (is_spec(h->p))                                                                                                                    \
        hipLaunchKernelGGL((tick_pair_lds_kernel<true, ES>), g, b, q1pair::lds_bytes(ES), h->stream, h->p, h->st, ticks, tag0, results_dev, \
                           obs_final_dev, seed, h->tick_count, auto_reset, keys_dev, mouse_dev, checksum_dev, status_dev, timeout_ticks); \
    else                                                                                                                                  \
        hipLaunchKernelGGL((tick_pair_lds_kernel<false, ES>), g, b, q1pair::lds_bytes(ES), h->stream, h->p, h->st, ticks, tag0, results_dev, \
                           obs_final_dev, seed, h->tick_count, auto_reset, keys_dev, mouse_dev, checksum_dev, status_dev, timeout_ticks)
    Q1_FOR_PAIR_ES(es, Q1_LAUNCH)
#undef Q1_LAUNCH
    HIP_TRY(hipGetLastError());
    if (t_stop) HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->tick_count += (uint64_t)ticks;
    return Q1ENV_OK;
}

// out4 = {1 if this library was built with -DQ1_CHECK else 0, 16-byte granule-pair stores checked, mismatches found, 0}.
// clear != 0 zeroes the device counters after reading them.
int q1env_debug_counters(q1env_t* h, uint64_t* out4, int clear) {
    if (!h || !out4) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_debug_counters: null argument");
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
#ifdef Q1_CHECK
    DeviceGuard guard(h->device);
    HIP_TRY(hipStreamSynchronize(h->stream));
    unsigned long long v[2] = {0ull, 0ull};
    HIP_TRY(hipMemcpyFromSymbol(&v[0], HIP_SYMBOL(q1_check_pair_stores), sizeof(v[0])));
    HIP_TRY(hipMemcpyFromSymbol(&v[1], HIP_SYMBOL(q1_check_pair_mismatches), sizeof(v[1])));
    out4[0] = 1; out4[1] = v[0]; out4[2] = v[1];
    if (clear) {
        const unsigned long long z = 0ull;
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(q1_check_pair_stores), &z, sizeof(z)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(q1_check_pair_mismatches), &z, sizeof(z)));
    }
#else
    (void)clear;
#endif
    return Q1ENV_OK;
}

}  // extern "C"
