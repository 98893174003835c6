// q1env.hip - kernels, handle and C ABI of libq1env.so (gfx950 only; see include/q1env.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared q1env.hip -o libq1env.so
// (-ffp-contract=off is part of the numerics contract: the reference never fuses multiply-add.)
#include "q1env_device.hpp"
#include "q1policy.hpp"
#include "../../include/q1env.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace q1;

// =========================================================================================== kernels
// One tick of every env (reference VectorPhysEnv.vector_step, env.py:482-510), one lane per env.
// Loads: 85 B of SoA state + the action; stores: the state + obs/reward/done.  The per-tick constants sit in SGPRs.
// LDS is used for one thing only: transposing the wave's 64 float32 observation rows so they leave as 16-B-per-lane
// coalesced stores (write_obs_wave_f32).
//   SPEC: default Config structure baked in (straight-line tick);  FMT: action layout, or FMT_RUNTIME.
template <typename OBS_T, bool SPEC, int FMT>
__global__ void __launch_bounds__(256)
step_kernel(float* pvx, float* pvy, float* pvz, double* ppx, double* ppy, double* pz, double* pyaw, double* ptrem,   // preloaded into SGPRs
            Params p, StatePtrs s, int fmt, const void* act_a, const void* act_b,
            OBS_T* obs, float* reward, uint8_t* done, uint8_t* zero_start) {
    // The eight leading pointers repeat s.vx .. s.trem: leading scalar kernel arguments are preloaded into SGPRs by the command
    // processor (-mllvm -amdgpu-kernarg-preload-count), so the first state loads do not wait for an s_load of the kernarg segment.
    __shared__ float slab[4][384];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = (uint32_t)p.n;
    if (i >= n) return;
    s.vx = pvx; s.vy = pvy; s.vz = pvz; s.px = ppx; s.py = ppy; s.z = pz; s.yaw = pyaw; s.trem = ptrem;
    Env e;
    load_env(s, n, i, e);
    const Env loaded = e;
    double yaw_act;
    const uint32_t keys = fetch_action<SPEC, FMT>(p, fmt, act_a, act_b, (size_t)i, &yaw_act);
    TickOut<OBS_T> o;
    tick<OBS_T, SPEC>(p, e, keys, yaw_act, o);
    store_env_delta(s, n, i, e, loaded);
    if (obs) {
        if constexpr (sizeof(OBS_T) == 4) {
            const uint32_t lane = threadIdx.x & 63u, wave_first = i - lane;
            if (wave_first + 64u <= n) write_obs_wave_f32_nt(obs, wave_first, lane, o.obs, slab[threadIdx.x >> 6]);
            else write_obs<OBS_T>(obs, (size_t)i, o.obs);
        } else {
            write_obs<OBS_T>(obs, (size_t)i, o.obs);
        }
    }
    if (reward) __builtin_nontemporal_store(o.reward, reward + i);
    if (done) __builtin_nontemporal_store((uint8_t)(o.done ? 1 : 0), done + i);
    if (zero_start) zero_start[i] = (e.flags & FLAG_ZERO_START) ? 1 : 0;
}

// One tick WITH in-kernel reset of the envs whose episode ended on it (the "auto-reset" vector-env convention of
// GPU-resident RL loops): reward / done / zero_start describe the finished step; the observation row of a finished env is
// the FIRST observation of its next episode (Philox reset exactly as q1env_reset_philox with counter + 1).  Saves the second
// launch of the step + reset_philox(done_only) pair; bit-identical to that pair.
template <bool SPEC, int FMT>
__global__ void __launch_bounds__(256)
step_autoreset_kernel(Params p, StatePtrs s, int fmt, const void* act_a, const void* act_b, uint64_t seed, uint64_t counter,
                      const uint64_t* counter_dev, float* obs, float* reward, uint8_t* done, uint8_t* zero_start) {
    __shared__ float slab[4][384];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = (uint32_t)p.n;
    if (i >= n) return;
    if (counter_dev) counter += *counter_dev;
    Env e;
    load_env(s, n, i, e);
    const Env loaded = e;
    double yaw_act;
    const uint32_t keys = fetch_action<SPEC, FMT>(p, fmt, act_a, act_b, (size_t)i, &yaw_act);
    TickOut<float> o;
    tick<float, SPEC>(p, e, keys, yaw_act, o);
    if (zero_start) zero_start[i] = (e.flags & FLAG_ZERO_START) ? 1 : 0;      // of the episode the step belonged to
    if (o.done) {
        reset_philox(p, e, seed, (uint64_t)p.env_index_base + (uint64_t)i, counter + 1);
        observe<float>(p, e, o.obs);
    }
    store_env_delta(s, n, i, e, loaded);
    if (obs) {
        const uint32_t lane = threadIdx.x & 63u, wave_first = i - lane;
        if (wave_first + 64u <= n) write_obs_wave_f32(obs, wave_first, lane, o.obs, slab[threadIdx.x >> 6]);
        else write_obs<float>(obs, (size_t)i, o.obs);
    }
    if (reward) reward[i] = o.reward;
    if (done) done[i] = o.done ? 1 : 0;
}

// `ticks` ticks in one launch: the env state lives in registers between ticks, only actions stream in and
// (optional) per-tick outputs stream out.  Tick-major layouts keep every access of a wave contiguous.
// The next tick's action is fetched before the current tick is computed, so its HBM latency hides under the
// tick's float64 arithmetic instead of adding to it (a lone wave per SIMD has nothing else to hide it with).
//   OUT_MODE: 1 = obs, reward and done are all written every tick (no null checks -> static store count),
//             0 = no per-tick output at all, -1 = decided per pointer at run time.
//   FULL:     every lane of the wave owns an env (the ragged tail wave runs its own copy of the loop, so that the
//             number of stores per iteration is a compile-time constant in both).
template <typename OBS_T, bool SPEC, int FMT, bool HAS_RESET, int OUT_MODE, bool FULL>
__device__ __forceinline__ void rollout_loop(const Params& p, Env& e, uint32_t i, uint32_t n, int ticks, int fmt,
                                             const void* act_a, const void* act_b, uint64_t seed, uint64_t tick0,
                                             OBS_T* obs, float* reward, uint8_t* done, int auto_reset, double& ret,
                                             float* slab) {
    const uint64_t genv = (uint64_t)p.env_index_base + (uint64_t)i;
    const bool random = (FMT >= 0 ? FMT : fmt) == FMT_RANDOM;
    const uint32_t lane = threadIdx.x & 63u, wave_first = i - lane;
    // Software prefetch for the packed layout: the RAW bytes of tick t+1's action are requested at the top of the
    // body (unconditionally, index clamped on the last tick, so the loads stay in the body's first basic block),
    // before tick t is computed, and are only decoded one iteration later; other layouts fetch in place.
    constexpr bool PREFETCH = (FMT == FMT_PACKED);
    uint32_t kraw_next = 0;
    float mraw_next = 0.0f;
    if constexpr (PREFETCH) {
        kraw_next = ((const uint8_t*)act_a)[i];
        mraw_next = ((const float*)act_b)[i];
        // Drain every outstanding load (state + first action) once, here: the waitcnt scoreboard then enters the
        // loop clean, so inside the loop the wait for a prefetched action is vmcnt(#younger ops) as seen along the
        // back edge - it no longer has to cover the preheader's load order and does not drain the tick's stores.
        __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0), expcnt/lgkmcnt untouched
    }
    for (int t = 0; t < ticks; ++t) {
        double yaw_act;
        uint32_t keys;
        if constexpr (PREFETCH) {
            const uint32_t kraw = kraw_next;
            const float mraw = mraw_next;
            const size_t nxt = (size_t)(t + 1 < ticks ? t + 1 : t) * n + i;
            kraw_next = ((const uint8_t*)act_a)[nxt];
            mraw_next = ((const float*)act_b)[nxt];
            keys = kraw & 0xFu;
            yaw_act = (double)mraw;
        } else if (random) {
            keys = random_action<SPEC>(p, seed, genv, tick0 + (uint64_t)t, &yaw_act);
        } else {
            keys = fetch_action<SPEC, FMT>(p, fmt, act_a, act_b, (size_t)t * n + i, &yaw_act);
        }
        TickOut<OBS_T> o;
        tick<OBS_T, SPEC>(p, e, keys, yaw_act, o);
        const size_t base = (size_t)t * n;
        if (OUT_MODE == 1 || (OUT_MODE < 0 && obs)) {
            if constexpr (sizeof(OBS_T) == 4 && FULL) write_obs_wave_f32(obs, base + wave_first, lane, o.obs, slab);
            else write_obs<OBS_T>(obs, base + i, o.obs);
        }
        if (OUT_MODE == 1 || (OUT_MODE < 0 && reward)) (reward + base)[i] = o.reward;
        if (OUT_MODE == 1 || (OUT_MODE < 0 && done)) (done + base)[i] = o.done ? 1 : 0;
        ret += (double)o.reward;
        if constexpr (HAS_RESET) {
            if (auto_reset && o.done) reset_philox(p, e, seed, genv, tick0 + (uint64_t)t + 1);
        }
    }
}

template <typename OBS_T, bool SPEC, int FMT, bool HAS_RESET, int OUT_MODE>
__global__ void __launch_bounds__(256)
rollout_kernel(Params p, StatePtrs s, int ticks, int fmt, const void* act_a, const void* act_b,
               uint64_t seed, uint64_t tick0, OBS_T* obs, float* reward, uint8_t* done,
               int auto_reset, double* return_sum) {
    __shared__ float slab[4][384];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = (uint32_t)p.n;
    if (i >= n) return;
    Env e;
    load_env(s, n, i, e);
    double ret = 0.0;
    float* my_slab = slab[threadIdx.x >> 6];
    if (i - (threadIdx.x & 63u) + 64u <= n)
        rollout_loop<OBS_T, SPEC, FMT, HAS_RESET, OUT_MODE, true>(p, e, i, n, ticks, fmt, act_a, act_b, seed, tick0, obs, reward,
                                                                   done, auto_reset, ret, my_slab);
    else
        rollout_loop<OBS_T, SPEC, FMT, HAS_RESET, OUT_MODE, false>(p, e, i, n, ticks, fmt, act_a, act_b, seed, tick0, obs, reward,
                                                                    done, auto_reset, ret, my_slab);
    store_env(s, n, i, e);
    if (return_sum) return_sum[i] += ret;
}

template <typename OBS_T>
__global__ void __launch_bounds__(256) observe_kernel(Params p, StatePtrs s, OBS_T* obs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)p.n) return;
    Env e;
    load_env(s, (uint32_t)p.n, i, e);
    OBS_T o[6];
    observe<OBS_T>(p, e, o);
    write_obs<OBS_T>(obs, (size_t)i, o);
}

// Reset from host-supplied raw draws (NumPy-compatible RNG stays on the host, the arithmetic is here).
template <typename OBS_T>
__global__ void __launch_bounds__(256)
reset_draws_kernel(Params p, StatePtrs s, int count, const int32_t* idx, const uint8_t* zero_start,
                   const double* yaw, const double* tm, const double* speed, const double* angle, OBS_T* obs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint32_t i = idx ? (uint32_t)idx[j] : (uint32_t)j;
    Env e;
    reset_from_draws(p, e, zero_start[j] != 0, yaw[j], tm[j], speed[j], angle[j]);
    store_env(s, (uint32_t)p.n, i, e);
    if (obs) {
        OBS_T o[6];
        observe<OBS_T>(p, e, o);
        write_obs<OBS_T>(obs, (size_t)j, o);
    }
}

template <typename OBS_T>
__global__ void __launch_bounds__(256)
reset_philox_kernel(Params p, StatePtrs s, uint64_t seed, uint64_t counter, const uint64_t* counter_dev, const uint8_t* mask,
                    int done_only, OBS_T* obs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)p.n) return;
    if (counter_dev) counter += *counter_dev;          // device-resident tick counter (hipGraph-replayable loops)
    Env e;
    load_env(s, (uint32_t)p.n, i, e);
    bool go = mask ? (mask[i] != 0) : true;
    if (done_only) go = go && (e.trem < 0.0);
    if (go) {
        reset_philox(p, e, seed, (uint64_t)p.env_index_base + (uint64_t)i, counter);
        store_env(s, (uint32_t)p.n, i, e);
    }
    if (obs) {
        OBS_T o[6];
        observe<OBS_T>(p, e, o);
        write_obs<OBS_T>(obs, (size_t)i, o);
    }
}

// Stand-alone ActionDecoder.map (env.py:225-269): decoder state from the handle, z_vel / time from the caller.
__global__ void __launch_bounds__(256)
decode_kernel(Params p, StatePtrs s, int fmt, const void* act_a, const void* act_b, const float* z_vel,
              const double* trem, double* yaw, int64_t* smove, int64_t* fmove, uint8_t* jump) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)p.n) return;
    Env e;
    load_env(s, (uint32_t)p.n, i, e);
    double yaw_act;
    const uint32_t keys = fetch_action<false, FMT_RUNTIME>(p, fmt, act_a, act_b, (size_t)i, &yaw_act);
    const Cmd c = decode<false>(p, e, keys, yaw_act, z_vel[i], trem[i]);
    s.yaw[i] = e.yaw;
#pragma unroll
    for (int k = 0; k < 4; ++k) s.lk[(size_t)k * p.n + i] = e.lk[k];
    s.flags[i] = (uint8_t)e.flags;
    yaw[i] = e.yaw;
    smove[i] = (int64_t)c.smove;
    fmove[i] = (int64_t)c.fmove;
    jump[i] = c.jump ? 1 : 0;
}

__global__ void __launch_bounds__(256)
decoder_reset_kernel(Params p, StatePtrs s, int count, const int32_t* idx, const double* yaw) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const int i = idx ? idx[j] : j;
#pragma unroll
    for (int k = 0; k < 4; ++k) s.lk[(size_t)k * p.n + i] = -p.key_press_delay;   // env.py:277-278 / 289
    s.flags[i] = s.flags[i] & 0x7u;                                               // env.py:279 / 290
    s.yaw[i] = yaw[j];                                                            // env.py:281 / 291
}

// Policy-side glue of the sampler loop (counterpart of reference q1physrl/action_dist.py:46-243, the TF
// `Q1PhysActionDist`): turn one row of policy-network outputs into a sampled action, already in the packed layout
// step_kernel consumes, plus its log-probability - one launch instead of ~20 elementwise torch ops per tick.
// Row layout (action_dist.py:207-226 + RLlib's MultiActionDistribution): num_keys x [logit0, logit1] (Discrete(2)
// categorical per key), then [mean, log_std] of the mouse Gaussian.  float32 arithmetic like the TF original.
//   key k:   P(1) = softmax(logits)[1];  logp = log softmax[chosen]
//   mouse:   mean clipped to +-3, log_std to [-20, 2] (action_dist.py:68-72); u = mean + std*eps;
//            x = clip(NormalCDF(u / S), 1e-6, 1 - 1e-6) * (high - low) + low,  S = 0.5 * 1.8137  (action_dist.py:151,186-192)
//            logp = N(mean,std).logpdf(u') - N(0,S).logpdf(u') - log(high - low), u' = S * ndtri((x - low)/(high - low))
//            (action_dist.py:91-96,180-184,194-196)
// Discrete mouse (Config.discrete_yaw_steps = S > 0, env.py:216-219): the Tuple's last child is Discrete(2S+1), which the reference
// takes through ModelCatalog.get_action_dist (action_dist.py:221-222) -> RLlib's Categorical over M = 2S+1 logits that follow the
// key pairs in the row.  Sampling: inverse CDF of softmax(logits) on one uniform; deterministic: first arg-max (tf.argmax);
// logp = logit[choice] - logsumexp.  The action leaves as the step index in the packed layout's float mouse slot, which is what
// the decoder's discrete branch consumes ((a - S) * max_yaw_delta / S, env.py:238).
// Randomness: Philox stream 3 keyed by (seed, global env, counter): r[0] low bits -> one uniform per key, r[2],r[3] -> Box-Muller
// (r[2] alone -> the categorical's uniform).
constexpr uint32_t STREAM_POLICY = 3;

__device__ __forceinline__ void sample_categorical(const float* __restrict__ lg, int m, uint32_t rnd, int deterministic,
                                                   int& choice, float& logp) {
    float mx = lg[0];
    int arg = 0;
    for (int j = 1; j < m; ++j) {
        const float v = lg[j];
        if (v > mx) { mx = v; arg = j; }
    }
    float sum = 0.0f;
    for (int j = 0; j < m; ++j) sum += expf(lg[j] - mx);
    choice = arg;
    if (!deterministic) {
        const float target = ((float)(rnd >> 8) * (1.0f / 16777216.0f)) * sum;     // u in [0, 1) scaled to the unnormalised mass
        float acc = 0.0f;
        choice = m - 1;
        for (int j = 0; j < m; ++j) {
            acc += expf(lg[j] - mx);
            if (acc > target) { choice = j; break; }
        }
    }
    logp = (lg[choice] - mx) - logf(sum);
}

// log(S) of the squashing scale S = 0.5f * 1.8137f as float32 (shared by the sampling and the loss kernels, so that the constant
// terms of a log-probability cancel exactly in a ratio)
constexpr float SQUASH_SCALE = 0.5f * 1.8137f;
constexpr float LOG_SQUASH_SCALE = -0.097778246f;     // float32(log(float32(0.90685)))

// The arithmetic of one env's action from its (up to ten) policy outputs in registers; `row` is only read by the discrete-mouse
// branch (2S+1 logits behind the key pairs).
// (the two Philox words of the (env, counter) pair: they do not depend on the policy's outputs, so a caller with idle time before
// the logits arrive - the resident sampler - draws them ahead)
__device__ __forceinline__ void sample_action_draws(uint64_t seed, uint64_t genv, uint64_t counter, uint32_t (&r)[4], uint32_t (&r2)[4]) {
    philox_draw(seed, genv, counter, STREAM_POLICY, 0, r);
    philox_draw(seed, genv, counter, STREAM_POLICY, 1, r2);
}

__device__ __forceinline__ void sample_action_from_draws(const Params& p, const float (&lg)[10], const float* __restrict__ row,
                                                         const uint32_t (&r)[4], const uint32_t (&r2)[4], int deterministic, uint32_t& keys,
                                                         float& mouse, float& logp) {
    logp = 0.0f;
    keys = 0;
    const uint32_t ku[4] = {r[0], r[1], r2[0], r2[1]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= p.num_keys) break;
        const float l0 = lg[2 * k], l1 = lg[2 * k + 1];
        const float d = l1 - l0;                                  // P(1) = sigmoid(d)
        // one exponential serves both the probability and the log-probability: e = exp(-|d|);
        // sigmoid(d) = 1 / (1 + e) for d >= 0, e / (1 + e) otherwise; softplus(+-d) = max(+-d, 0) + log1p(e)
        const float e = expf(-fabsf(d));
        const float rc = 1.0f / (1.0f + e);
        const float p1 = d >= 0.0f ? rc : e * rc;
        const float u = (float)(ku[k] >> 8) * (1.0f / 16777216.0f);
        const uint32_t bit = deterministic ? (d > 0.0f) : (u < p1);   // deterministic: argmax (RLlib Categorical)
        keys |= bit << k;
        const float z = bit ? -d : d;                             // log softmax[chosen] = -softplus(l_other - l_chosen)
        logp -= (z > 0.0f ? z : 0.0f) + log1pf(e);
    }
    mouse = 0.0f;
    if (p.yaw_mode == 1) {
        const float S = SQUASH_SCALE;
        const float low = -p.action_range_f32, high = p.action_range_f32;
        float mean = 0.0f, log_std = 0.0f;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (j == p.num_keys) { mean = lg[2 * j]; log_std = lg[2 * j + 1]; }
        mean = fminf(fmaxf(mean, -3.0f), 3.0f);
        log_std = fminf(fmaxf(log_std, -20.0f), 2.0f);
        const float std = expf(log_std);
        float eps = 0.0f;
        if (!deterministic) {
            const float u1 = ((float)(r[2] >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
            const float u2 = (float)(r[3] >> 8) * (1.0f / 16777216.0f);
            eps = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853071795865f * u2);
        }
        const float un = mean + std * eps;
        float c = normcdff(un / S);
        c = fminf(fmaxf(c, 1e-6f), 1.0f - 1e-6f);
        mouse = c * (high - low) + low;
        const float ub = S * normcdfinvf((mouse - low) / (high - low));
        const float zs = (ub - mean) / std;
        const float lp_pi = -0.5f * zs * zs - log_std - 0.9189385332046727f;            // N(mean, std).logpdf(ub)
        const float zq = ub / S;
        const float lp_sq = -0.5f * zq * zq - LOG_SQUASH_SCALE - 0.9189385332046727f;   // N(0, S).logpdf(ub)
        logp += lp_pi - (lp_sq + p.log_range_f32);
    } else if (p.yaw_mode == 2) {
        int choice;
        float lpc;
        sample_categorical(row + 2 * p.num_keys, 2 * (int)p.yaw_steps + 1, r[2], deterministic, choice, lpc);
        mouse = (float)choice;
        logp += lpc;
    }
}

__device__ __forceinline__ void sample_action_regs(const Params& p, const float (&lg)[10], const float* __restrict__ row, uint64_t seed,
                                                   uint64_t genv, uint64_t counter, int deterministic, uint32_t& keys, float& mouse,
                                                   float& logp) {
    uint32_t r[4], r2[4];
    sample_action_draws(seed, genv, counter, r, r2);
    sample_action_from_draws(p, lg, row, r, r2, deterministic, keys, mouse, logp);
}

__device__ __forceinline__ void sample_action(const Params& p, const float* __restrict__ row, uint64_t seed, uint64_t genv,
                                              uint64_t counter, int deterministic, uint32_t& keys, float& mouse, float& logp) {
    // all (up to ten) policy outputs of the row are requested before anything is computed: inside the per-key loop each pair
    // would expose its own HBM round trip (one wave per SIMD at sampler batch sizes has nothing to hide it with)
    float lg[10];
    const int pairs = p.num_keys + (p.yaw_mode == 1 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        lg[2 * j] = j < pairs ? row[2 * j] : 0.0f;
        lg[2 * j + 1] = j < pairs ? row[2 * j + 1] : 0.0f;
    }
    sample_action_regs(p, lg, row, seed, genv, counter, deterministic, keys, mouse, logp);
}

__global__ void __launch_bounds__(256)
policy_sample_kernel(Params p, const float* __restrict__ logits, int row_stride, uint64_t seed, uint64_t counter,
                     const uint64_t* counter_dev, int deterministic, uint8_t* keys_out, float* mouse_out, float* logp_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)p.n) return;
    if (counter_dev) counter += *counter_dev;
    uint32_t keys;
    float mouse, logp;
    sample_action(p, logits + (size_t)i * row_stride, seed, (uint64_t)p.env_index_base + i, counter, deterministic, keys, mouse, logp);
    keys_out[i] = (uint8_t)keys;
    if (mouse_out) mouse_out[i] = mouse;
    if (logp_out) logp_out[i] = logp;
}

// Generalised advantage estimation over a tick-major trajectory (learner-side glue; RLlib's
// compute_advantages with use_gae, lambda/gamma of reference data/params.yml:4-7).  One lane per env walks its T ticks
// backwards; every access of a wave is a contiguous 256-B segment of the [T][N] arrays.
//   delta_t = r_t + gamma * V_{t+1} * (1 - done_t) - V_t ;  A_t = delta_t + gamma * lambda * (1 - done_t) * A_{t+1}
//   value has T+1 rows (bootstrap row last); vtarg_t = A_t + V_t.
__global__ void __launch_bounds__(256)
gae_kernel(int n, int ticks, const float* __restrict__ reward, const float* __restrict__ value,
           const uint8_t* __restrict__ done, float gamma, float lam, float* __restrict__ adv, float* __restrict__ vtarg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)n) return;
    float a = 0.0f;
    float v_next = value[(size_t)ticks * n + i];
    for (int t = ticks - 1; t >= 0; --t) {
        const size_t o = (size_t)t * n + i;
        const float nd = done[o] ? 0.0f : 1.0f;
        const float v = value[o];
        const float delta = reward[o] + gamma * v_next * nd - v;
        a = delta + gamma * lam * nd * a;
        adv[o] = a;
        vtarg[o] = a + v;
        v_next = v;
    }
}

// PPO loss of one minibatch and its gradient with respect to the policy outputs (learner-side glue): the
// closed forms of q1physrl_amd/ppo.py::ppo_loss (RLlib 0.8.4 PPOLoss over the reference's Q1PhysActionDist, action_dist.py:46-243)
// differentiated by hand, one lane per sample - replaces ~100 elementwise launches + their autograd twins per SGD step:
//   keys k:  d = l1 - l0, p = sigmoid(d), a = action bit:  logp -= softplus(a ? -d : d)            d logp / dd = a - p
//            H += softplus(d) - d p                                                                   dH / dd = -d p (1 - p)
//            KL(old || new) += p_o (logp_o1 - logp_n1) + (1 - p_o)(logp_o0 - logp_n0)                dKL / dd = p - p_o
//   mouse:   u = S ndtri((x - low) / (high - low)), z = (u - mean) / std  (mean, log_std clamped; the clamp gates the gradient)
//            logp += N(mean, std).logpdf(u) - N(0, S).logpdf(u) - log(high - low)                    d/dmean = z / std, d/dlog_std = z^2 - 1
//            H  += log(high - low) - (log S - log_std + (std^2 + mean^2) / (2 S^2) - 1/2)
//            KL += log_std - log_std_o + (std_o^2 + (mean_o - mean)^2) / (2 std^2) - 1/2
//   ratio = exp(logp - logp_old), surrogate = min(adv ratio, adv clip(ratio, 1 -+ c))              d/dlogp = adv ratio if adv ratio <= adv clip(..)
//   vf = max((v - vt)^2, (v_old + clip(v - v_old, +-vc) - vt)^2)
//   total = mean(-surrogate + kl_coeff KL + vf_coeff vf - ent_coeff H);  dlogits / dvalue are d total / d(logits, value).
// partials[block][5] = sums of (entropy, kl, -surrogate, total, vf) over the block's samples (no atomics; add them up and divide by B).
__global__ void __launch_bounds__(256)
ppo_loss_grad_kernel(Params p, int batch, const float* __restrict__ logits, const float* __restrict__ old_logits, int row_stride,
                     const uint8_t* __restrict__ keys, const float* __restrict__ mouse, const float* __restrict__ logp_old,
                     const float* __restrict__ adv, const float* __restrict__ value, const float* __restrict__ value_old,
                     const float* __restrict__ vtarg, float clip, float vf_clip, float vf_coeff, float ent_coeff,
                     const float* __restrict__ kl_coeff_dev, float* __restrict__ dlogits, float* __restrict__ dvalue,
                     float* __restrict__ partials) {
    __shared__ float red[4][5];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < (uint32_t)batch;
    const float klc = *kl_coeff_dev;
    const float inv_b = 1.0f / (float)batch;
    float st[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
        const float* row = logits + (size_t)i * row_stride;
        const float* old = old_logits + (size_t)i * row_stride;
        float* g = dlogits + (size_t)i * row_stride;
        const int nk = p.num_keys;
        const uint32_t kb = keys[i];
        float logp = 0.0f, ent = 0.0f, kl = 0.0f;
        float dlp[4], dh[4], dk[4];
        auto softplus = [](float z) { return (z > 0.0f ? z : 0.0f) + log1pf(expf(-fabsf(z))); };
        for (int k = 0; k < 4; ++k) {
            if (k >= nk) break;
            const float d = row[2 * k + 1] - row[2 * k], d_o = old[2 * k + 1] - old[2 * k];
            const float pn = 1.0f / (1.0f + expf(-d)), po = 1.0f / (1.0f + expf(-d_o));
            const float a = (float)((kb >> k) & 1u);
            const float sp_pos = softplus(d), sp_neg = softplus(-d);            // -log p(0), -log p(1)
            logp -= a != 0.0f ? sp_neg : sp_pos;
            ent += sp_pos - d * pn;
            kl += po * (sp_neg - softplus(-d_o)) + (1.0f - po) * (sp_pos - softplus(d_o));
            dlp[k] = a - pn; dh[k] = -d * pn * (1.0f - pn); dk[k] = pn - po;
        }
        float dlp_m = 0.0f, dlp_s = 0.0f, dh_m = 0.0f, dh_s = 0.0f, dk_m = 0.0f, dk_s = 0.0f;
        bool in_m = false, in_s = false;
        // discrete mouse: Categorical over M = 2S+1 logits (see sample_categorical):
        //   logp += l_a - lse;  H_c = -sum p_j log p_j;  KL_c = sum po_j (log po_j - log p_j)
        //   d logp / d l_j = [j == a] - p_j;  d H_c / d l_j = -p_j (log p_j + H_c);  d KL_c / d l_j = p_j - po_j
        const int cat_m = p.yaw_mode == 2 ? 2 * (int)p.yaw_steps + 1 : 0;
        float cat_lse = 0.0f, cat_lse_o = 0.0f, cat_h = 0.0f;
        int cat_a = 0;
        if (p.yaw_mode == 2) {
            const float* l = row + 2 * nk;
            const float* lo = old + 2 * nk;
            float mx = l[0], mxo = lo[0];
            for (int j = 1; j < cat_m; ++j) { mx = fmaxf(mx, l[j]); mxo = fmaxf(mxo, lo[j]); }
            float sn = 0.0f, so = 0.0f;
            for (int j = 0; j < cat_m; ++j) { sn += expf(l[j] - mx); so += expf(lo[j] - mxo); }
            cat_lse = mx + logf(sn);
            cat_lse_o = mxo + logf(so);
            cat_a = min(max((int)mouse[i], 0), cat_m - 1);
            logp += l[cat_a] - cat_lse;
            float kc = 0.0f;
            for (int j = 0; j < cat_m; ++j) {
                const float lpn = l[j] - cat_lse, lpo = lo[j] - cat_lse_o;
                cat_h -= expf(lpn) * lpn;
                kc += expf(lpo) * (lpo - lpn);
            }
            ent += cat_h;
            kl += kc;
        }
        if (p.yaw_mode == 1) {
            const float S = SQUASH_SCALE, low = -p.action_range_f32, high = p.action_range_f32;
            const float m_raw = row[2 * nk], s_raw = row[2 * nk + 1];
            in_m = m_raw >= -3.0f && m_raw <= 3.0f;
            in_s = s_raw >= -20.0f && s_raw <= 2.0f;
            const float mean = fminf(fmaxf(m_raw, -3.0f), 3.0f), ls = fminf(fmaxf(s_raw, -20.0f), 2.0f);
            const float mean_o = fminf(fmaxf(old[2 * nk], -3.0f), 3.0f), ls_o = fminf(fmaxf(old[2 * nk + 1], -20.0f), 2.0f);
            const float inv_std = expf(-ls), std = expf(ls), std_o = expf(ls_o);
            const float u = S * normcdfinvf((mouse[i] - low) / (high - low));
            const float z = (u - mean) * inv_std, zq = u / S;
            logp += (-0.5f * z * z - ls - 0.9189385332046727f) - ((-0.5f * zq * zq - LOG_SQUASH_SCALE - 0.9189385332046727f) + p.log_range_f32);
            dlp_m = z * inv_std; dlp_s = z * z - 1.0f;
            ent += p.log_range_f32 - (LOG_SQUASH_SCALE - ls + (std * std + mean * mean) / (2.0f * S * S) - 0.5f);
            dh_m = -mean / (S * S); dh_s = 1.0f - std * std / (S * S);
            const float dm = mean_o - mean, q = (std_o * std_o + dm * dm) * inv_std * inv_std;
            kl += ls - ls_o + 0.5f * q - 0.5f;
            dk_m = -dm * inv_std * inv_std; dk_s = 1.0f - q;
        }
        const float ratio = expf(logp - logp_old[i]), ad = adv[i];
        const float s1 = ad * ratio, s2 = ad * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
        const float surr = fminf(s1, s2);
        const float c_lp = s1 <= s2 ? -s1 : 0.0f;                                    // d(-surrogate) / dlogp
        const float v = value[i], vo = value_old[i], vt = vtarg[i];
        const float dv = v - vo, vc = vo + fminf(fmaxf(dv, -vf_clip), vf_clip);
        const float e1 = (v - vt) * (v - vt), e2 = (vc - vt) * (vc - vt);
        const float vf = fmaxf(e1, e2);
        const float dvf = e1 >= e2 ? 2.0f * (v - vt) : ((dv >= -vf_clip && dv <= vf_clip) ? 2.0f * (vc - vt) : 0.0f);
        for (int k = 0; k < 4; ++k) {
            if (k >= nk) break;
            const float gd = (c_lp * dlp[k] + klc * dk[k] - ent_coeff * dh[k]) * inv_b;
            g[2 * k] = -gd; g[2 * k + 1] = gd;
        }
        if (p.yaw_mode == 1) {
            g[2 * nk] = in_m ? (c_lp * dlp_m + klc * dk_m - ent_coeff * dh_m) * inv_b : 0.0f;
            g[2 * nk + 1] = in_s ? (c_lp * dlp_s + klc * dk_s - ent_coeff * dh_s) * inv_b : 0.0f;
        }
        if (p.yaw_mode == 2) {
            const float* l = row + 2 * nk;
            const float* lo = old + 2 * nk;
            for (int j = 0; j < cat_m; ++j) {
                const float lpn = l[j] - cat_lse, pn = expf(lpn), po = expf(lo[j] - cat_lse_o);
                const float dlp_j = (j == cat_a ? 1.0f : 0.0f) - pn, dh_j = -pn * (lpn + cat_h), dk_j = pn - po;
                g[2 * nk + j] = (c_lp * dlp_j + klc * dk_j - ent_coeff * dh_j) * inv_b;
            }
        }
        for (int c = 2 * nk + (p.yaw_mode == 1 ? 2 : cat_m); c < row_stride; ++c) g[c] = 0.0f;
        dvalue[i] = vf_coeff * dvf * inv_b;
        st[0] = ent; st[1] = kl; st[2] = -surr; st[3] = -surr + klc * kl + vf_coeff * vf - ent_coeff * ent; st[4] = vf;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) st[k] += __shfl_down(st[k], off, 64);
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = st[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) partials[(size_t)blockIdx.x * 5 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Episode bookkeeping of a sampler tick (the reference's on_episode_end hook, train.py:54-57): running return per env,
// and - for the envs whose episode ended on this tick - episode count / return sums, split by zero_start.  One wave
// reduces its 64 envs with cross-lane shuffles and adds into ITS OWN slot of `partials` ([ceil(n/64)][4] doubles:
// episodes, zero-start episodes, return sum, zero-start return sum): no atomics, bit-reproducible; the host sums the slots
// when statistics are asked for.  Replaces ~10 elementwise/reduction launches of the torch formulation.
// (all 64 lanes of the wave must call this: `live` masks the lanes without an env)
__device__ __forceinline__ void episode_stats_lane(bool live, uint32_t i, float reward, bool fin, bool zero_start,
                                                   double* __restrict__ ep_return, double* __restrict__ partials) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (live) {
        const double ret = ep_return[i] + (double)reward;
        const bool zs = fin && zero_start;
        ep_return[i] = fin ? 0.0 : ret;
        v[0] = fin ? 1.0 : 0.0; v[1] = zs ? 1.0 : 0.0; v[2] = fin ? ret : 0.0; v[3] = zs ? ret : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    if ((threadIdx.x & 63u) == 0 && live) {
        double* slot = partials + (size_t)(i >> 6) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) slot[k] += v[k];
    }
}

__global__ void __launch_bounds__(256)
episode_stats_kernel(int n, const float* __restrict__ reward, const uint8_t* __restrict__ done,
                     const uint8_t* __restrict__ zero_start, double* __restrict__ ep_return, double* __restrict__ partials) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < (uint32_t)n;
    episode_stats_lane(live, i, live ? reward[i] : 0.0f, live && done[i] != 0, live && zero_start[i] != 0, ep_return, partials);
}

// One sampler tick after the policy forward, in ONE launch: q1env_policy_sample -> q1env_step_autoreset (packed action) ->
// q1env_episode_stats, bit-identical to that sequence.  The sampled action goes from registers straight into the decoder (and
// to the trajectory arrays); reward / done / zero_start of the step feed the episode statistics without a round trip.
// counter = counter_offset + *counter_dev: a captured horizon bakes the tick index into counter_offset and advances the
// device counter once per horizon, so the tick needs no separate "counter += 1" launch either.
template <bool SPEC>
__global__ void __launch_bounds__(256)
sample_step_kernel(Params p, StatePtrs s, const float* __restrict__ logits, int row_stride, uint64_t seed, uint64_t counter,
                   const uint64_t* counter_dev, int deterministic, uint8_t* keys_out, float* mouse_out, float* logp_out,
                   float* obs, float* reward, uint8_t* done, uint8_t* zero_start, double* ep_return, double* partials) {
    __shared__ float slab[4][384];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = (uint32_t)p.n;
    const bool live = i < n;
    if (counter_dev) counter += *counter_dev;
    TickOut<float> o;
    o.reward = 0.0f; o.done = false;
    bool zs = false;
    if (live) {
        const uint64_t genv = (uint64_t)p.env_index_base + (uint64_t)i;
        const uint32_t lane = threadIdx.x & 63u, wave_first = i - lane;
        const bool full = wave_first + 64u <= n;
        float* my_slab = slab[threadIdx.x >> 6];
        Env e;
        load_env(s, n, i, e);                     // requested first: the state's HBM latency hides under the sampling arithmetic
        const Env loaded = e;
        uint32_t keys;
        float mouse, logp;
        // (an LDS-transposed, fully coalesced read of the wave's 64 logits rows was tried in round 2: 8.85 -> 8.73 us at 32 768 envs,
        // 19.8 -> 19.7 us at 262 144 - the kernel is bound by its float32 / float64 arithmetic and latency chain, not by these loads)
        sample_action(p, logits + (size_t)i * row_stride, seed, genv, counter, deterministic, keys, mouse, logp);
        keys_out[i] = (uint8_t)keys;
        if (mouse_out) mouse_out[i] = mouse;
        if (logp_out) logp_out[i] = logp;
        const double yaw_act = cfg_yaw_mode<SPEC>(p) ? (double)mouse : 0.0;        // the packed action layout: float32 mouse
        tick<float, SPEC>(p, e, keys & ((1u << cfg_num_keys<SPEC>(p)) - 1u), yaw_act, o);
        zs = (e.flags & FLAG_ZERO_START) != 0;                                      // of the episode the step belonged to
        if (zero_start) zero_start[i] = zs ? 1 : 0;
        if (o.done) {
            reset_philox(p, e, seed, genv, counter + 1);
            observe<float>(p, e, o.obs);
        }
        store_env_delta(s, n, i, e, loaded);
        if (full) write_obs_wave_f32(obs, wave_first, lane, o.obs, my_slab);
        else write_obs<float>(obs, (size_t)i, o.obs);
        reward[i] = o.reward;
        done[i] = o.done ? 1 : 0;
    }
    episode_stats_lane(live, i, o.reward, o.done, zs, ep_return, partials);
}

// *counter += by: the one extra node of a replayable run of auto-reset ticks (q1env_step_autoreset_many)
__global__ void counter_add_kernel(uint64_t* counter, uint64_t by) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *counter += by;
}

#include "q1server.hpp"       // tick_server_kernel / tick_driver_kernel / tick_pair_kernel (the resident tick server)
#include "q1resident.hpp"     // sampler_resident_kernel (a sampling horizon as one dispatch)

// Traffic calibration for the PMC counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE must be calibrated on
// a known byte count in the kernel's own access pattern): reads every SoA state array with exactly the loads
// step_kernel uses and writes the same bytes to a scratch arena: 85 B read + 85 B written per env, no arithmetic.
__global__ void __launch_bounds__(256) calib_copy_kernel(Params p, StatePtrs src, StatePtrs dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)p.n) return;
    Env e;
    load_env(src, (uint32_t)p.n, i, e);
    store_env(dst, (uint32_t)p.n, i, e);
}

// Self-test of the exact-division helpers against the hardware IEEE division on random operands drawn over the
// ranges the hot path produces (and well beyond).  counts[0..3] = mismatches of: div_const<double>, div_shared,
// the float32 vel-obs column, the float32 z-obs column.
__global__ void __launch_bounds__(256)
selftest_division_kernel(uint64_t n, uint64_t seed, double c_extra0, double c_extra1, unsigned long long* counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r[4], r2[4];
    philox_draw(seed, i, 0, 7, 0, r);
    philox_draw(seed, i, 1, 7, 0, r2);
    const double u = u53(r[0], r[1]), w = u53(r[2], r[3]);
    // magnitude sweep 1e-12 .. 1e7, both signs
    const double mag = exp(-27.6 + 43.7 * w);
    const double x = (2.0 * u - 1.0) * mag;
    const double cs[6] = {180.0, 90.0, 100.0, 200.0, c_extra0, c_extra1};
    unsigned bad0 = 0, bad1 = 0, bad2 = 0, bad3 = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double c = cs[k];
        const double a = div_const<double>(x, c, 1.0 / c), b = x / c;
        bad0 += (__double_as_longlong(a) != __double_as_longlong(b));
    }
    const double den = 0.5 + 4000.0 * u53(r2[0], r2[1]);
    const double num = (2.0 * u53(r2[2], r2[3]) - 1.0) * den;
    {
        const double y = rcp_refined(den);
        const double a = div_shared(num, den, y), b = num / den;
        bad1 += (__double_as_longlong(a) != __double_as_longlong(b));
        const double a2 = div_shared(x, den, y), b2 = x / den;
        bad1 += (__double_as_longlong(a2) != __double_as_longlong(b2));
        // friction quotient (phys.py:88-90): new_speed / speed with a float32 speed and 0 <= new_speed <= speed
        const float spf = (float)(0.001 + 3000.0 * u);
        const double ns = fmax(0.0, (double)spf - w * 60.0);
        const double a3 = div_shared(ns, (double)spf, rcp_refined((double)spf)), b3 = ns / (double)spf;
        bad1 += (__double_as_longlong(a3) != __double_as_longlong(b3));
    }
    {   // vel column: v float32 -> trunc(v/16)*16 / 200, float64 reference vs float32 shortcut
        const float v = (float)((2.0 * u - 1.0) * 40000.0);
        const double ref = (trunc((double)(v / 16.0f)) * 16.0 + 0.0) / 200.0;
        const float fast = div_const<float>(truncf(v * 0.0625f) * 16.0f + 0.0f, 200.0f, 1.0f / 200.0f);
        bad2 += (__float_as_uint((float)ref) != __float_as_uint(fast));
        const double z = 24.03125 + 3000.0 * w;
        const double refz = (rint(z * 8.0) / 8.0) / 100.0;
        const float fastz = div_const<float>((float)(rint(z * 8.0) * 0.125), 100.0f, 1.0f / 100.0f);
        bad3 += (__float_as_uint((float)refz) != __float_as_uint(fastz));
    }
    if (bad0) atomicAdd(&counts[0], (unsigned long long)bad0);
    if (bad1) atomicAdd(&counts[1], (unsigned long long)bad1);
    if (bad2) atomicAdd(&counts[2], (unsigned long long)bad2);
    if (bad3) atomicAdd(&counts[3], (unsigned long long)bad3);
}

// Stateless phys.apply (phys.py:184-197) with general pitch / roll (phys.py:56-66), all float64 trig.  VT = dtype of vel:
// float (the env's storage) or double (PlayerState.from_df, phys.py:168-170: nothing is rounded to float32 then).
template <typename VT>
__global__ void __launch_bounds__(256)
phys_apply_kernel(int n, const double* yaw, const double* pitch, const double* roll, const double* fmove,
                  const double* smove, const uint8_t* button2, const double* time_delta, const double* z_pos,
                  const VT* vel, const uint8_t* on_ground, const uint8_t* jump_released,
                  double* out_z, VT* out_vel, uint8_t* out_og, uint8_t* out_jr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    VT vx = vel[3 * (size_t)i], vy = vel[3 * (size_t)i + 1], vz = vel[3 * (size_t)i + 2];
    double z = z_pos[i];
    uint32_t flags = (on_ground[i] ? FLAG_ON_GROUND : 0u) | (jump_released[i] ? FLAG_JUMP_RELEASED : 0u);
    Cmd c;
    c.fmove = fmove[i]; c.smove = smove[i]; c.jump = button2[i] != 0;
    const double k = 3.141592653589793;
    double sy, cy, sp = 0.0, cp = 1.0, sr = 0.0, cr = 1.0;
    sincos((yaw[i] * k) / 180.0, &sy, &cy);
    if (pitch) sincos((pitch[i] * k) / 180.0, &sp, &cp);
    if (roll) sincos((roll[i] * k) / 180.0, &sr, &cr);
    const double m00 = cp * cy;
    const double m01 = ((-1.0 * sr) * sp) * cy + (-1.0 * cr) * (-sy);
    const double m10 = cp * sy;
    const double m11 = ((-1.0 * sr) * sp) * sy + (-1.0 * cr) * cy;
    const double dt = time_delta[i];
    physics_core<VT>(vx, vy, vz, z, flags, c, m00, m01, m10, m11, dt, 10.0 * dt, 800.0 * dt);
    out_z[i] = z;
    out_vel[3 * (size_t)i] = vx; out_vel[3 * (size_t)i + 1] = vy; out_vel[3 * (size_t)i + 2] = vz;
    out_og[i] = (flags & FLAG_ON_GROUND) ? 1 : 0;
    out_jr[i] = (flags & FLAG_JUMP_RELEASED) ? 1 : 0;
}

// =========================================================================================== host side
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

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
    // cached hipGraphs of step_many, keyed by (ticks, formats, pointers); a handful of entries, oldest evicted
    struct GraphEntry { std::vector<uint64_t> key; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    hipStream_t cap_stream = nullptr;  // private stream used only to CAPTURE (the null stream cannot be captured)
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline dim3 grid_for(int n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

// Small batches are latency bound: 64-lane workgroups spread 64 k envs over all 256 CUs (1024 waves).
// Large batches are bandwidth bound: 256-lane workgroups cut dispatch overhead.
static inline int block_for(int n) {
    static const int forced = [] { const char* e = getenv("Q1ENV_BLOCK"); return e ? atoi(e) : 0; }();   // tuning knob
    if (forced == 64 || forced == 128 || forced == 256) return forced;
    return n >= (1 << 19) ? 256 : 64;
}

static int ensure_stage(q1env* h, size_t bytes) {
    if (bytes <= h->stage_bytes) return 0;
    if (h->stage) (void)hipFree(h->stage);
    h->stage = nullptr;
    h->stage_bytes = 0;
    HIP_TRY(hipMalloc(&h->stage, bytes));
    h->stage_bytes = bytes;
    return 0;
}

static int ensure_pin(q1env* h, size_t bytes) {
    if (bytes <= h->pin_bytes) return 0;
    if (h->pin) (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->pin_bytes = 0;
    HIP_TRY(hipHostMalloc(&h->pin, bytes, hipHostMallocDefault));
    h->pin_bytes = bytes;
    return 0;
}

// Batches up to this many envs go through the handle's pinned staging as ONE block each way (the call count dominates there);
// larger batches copy every array directly (fast when the caller's arrays are pinned - q1env_host_alloc - as the Python layer's are).
constexpr size_t PACK_MAX_ENVS = 16384;

static int make_params(const q1env_config& c, Params& p, std::string& why) {
    if (c.num_envs <= 0) { why = "num_envs must be > 0"; return -1; }
    if (!(c.time_delta > 0)) { why = "time_delta must be > 0"; return -1; }
    if (c.allow_yaw && c.discrete_yaw_steps != -1 && c.discrete_yaw_steps <= 0) {
        why = "discrete_yaw_steps must be -1 or > 0"; return -1;
    }
    p.n = c.num_envs;
    const bool has_jump_action = !c.auto_jump && c.allow_jump;          // env.py:206
    p.num_keys = has_jump_action ? 4 : 3;                               // env.py:207
    p.yaw_mode = !c.allow_yaw ? 0 : (c.discrete_yaw_steps == -1 ? 1 : 2);
    p.act_width = p.num_keys + (p.yaw_mode ? 1 : 0);
    p.jump_mode = c.auto_jump ? 2 : (c.allow_jump ? 1 : 0);             // env.py:262-267
    p.smooth_keys = c.smooth_keys ? 1 : 0;
    p.smooth_prev = c.smooth_keys ? 1.0 : 0.0;                          // env.py:251-254 as exact 0/1 arithmetic
    p.smooth_scale = c.smooth_keys ? 0.5 : 1.0;
    p.hover = c.hover ? 1 : 0;
    p.speed_reward = c.speed_reward ? 1 : 0;
    p.dt = c.time_delta;
    p.time_limit = c.time_limit;
    p.key_press_delay = c.key_press_delay;
    // env.py:230 `_MAX_YAW_SPEED * time_delta` = np.float32(720) * python float: a float32 product under NumPy >= 2 (NEP 50, what
    // the golden fixtures were generated with), a float64 product under the NumPy 1.18.2 the reference pins (requirements.txt:33).
    // Equal for dt = 1/72 (10.0 either way); differs in the 9th digit for dt = 0.014 and the 14th for params.yml's truncated dt.
    p.yaw_num = c.legacy_promotion ? 720.0 * c.time_delta : (double)(720.0f * (float)c.time_delta);
    p.yaw_steps = (double)c.discrete_yaw_steps;
    p.yaw_den = (p.yaw_mode == 2) ? p.yaw_steps : c.action_range;       // env.py:236 / 238
    if (p.yaw_mode && !(p.yaw_den > 0)) { why = "action_range must be > 0"; return -1; }
    if (!(c.time_limit > 0)) { why = "time_limit must be > 0"; return -1; }
    p.yaw_den_rcp = p.yaw_mode ? 1.0 / p.yaw_den : 0.0;                 // correctly rounded reciprocals for div_const
    p.time_limit_rcp = 1.0 / c.time_limit;
    p.fmove_max = (double)(float)c.fmove_max;                           // env.py:261
    p.smove_max = (double)(float)c.smove_max;                           // env.py:260
    p.accel_dt = 10.0 * c.time_delta;                                   // phys.py:78
    p.grav_dt = 800.0 * c.time_delta;                                   // phys.py:122
    p.zero_start_prob = c.zero_start_prob;
    p.yaw_lo = c.initial_yaw_lo;
    p.yaw_hi = c.initial_yaw_hi;
    p.max_initial_speed = c.max_initial_speed;
    p.action_range = c.action_range;
    p.dt_f32 = (float)c.time_delta;                                     // env.py:501/503
    p.action_range_f32 = (float)c.action_range;
    p.log_range_f32 = logf(2.0f * (float)c.action_range);               // log(high - low) of the mouse Box, float32 like the kernels' terms
    p.env_index_base = c.env_index_base;
    return 0;
}

static void carve_into(void* arena, size_t n, StatePtrs& st) {
    char* base = (char*)arena;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* q = base + off; off += align_up(bytes, 256); return q; };
    st.vx = (float*)take(n * 4); st.vy = (float*)take(n * 4); st.vz = (float*)take(n * 4);
    st.px = (double*)take(n * 8); st.py = (double*)take(n * 8); st.z = (double*)take(n * 8);
    st.yaw = (double*)take(n * 8); st.trem = (double*)take(n * 8);
    st.lk = (double*)take(n * 8 * 4);
    st.flags = (uint8_t*)take(n);
}

static void carve(q1env* h) { carve_into(h->arena, (size_t)h->p.n, h->st); }

static size_t arena_bytes(size_t n) {
    return 3 * align_up(n * 4, 256) + 5 * align_up(n * 8, 256) + align_up(n * 32, 256) + align_up(n, 256);
}

// q1phys_apply_host keeps one scratch context per device (stream, device arena, pinned staging, grown on demand) instead of a
// hipMalloc / 15 synchronous copies / hipFree per call: analyse.py-style callers invoke phys.apply hundreds of times
// (hypothetical_delta_speeds, analyse.py:71-118).  Guarded by a mutex: the function is stateless for its callers.
namespace {
struct ApplyCtx { hipStream_t stream = nullptr; char* dev = nullptr; char* pin = nullptr; size_t bytes = 0; };
std::mutex g_apply_mutex;
ApplyCtx g_apply_ctx[64];
}

template <typename VT>
static int phys_apply_host_impl(int device, int64_t n64, const double* yaw, const double* pitch, const double* roll, const double* fmove,
                                const double* smove, const uint8_t* button2, const double* time_delta, const double* z_pos,
                                const VT* vel, const uint8_t* on_ground, const uint8_t* jump_released, double* out_z, VT* out_vel,
                                uint8_t* out_og, uint8_t* out_jr) {
    if (!yaw || !fmove || !smove || !button2 || !time_delta || !z_pos || !vel || !on_ground || !jump_released || !out_z ||
        !out_vel || !out_og || !out_jr)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1phys_apply_host: null argument");
    if (n64 <= 0 || n64 > (int64_t)1 << 30) return fail(Q1ENV_ERR_INVALID_ARG, "q1phys_apply_host: bad n");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(Q1ENV_ERR_NO_DEVICE, "q1phys_apply_host: no HIP device visible (libq1env has no CPU fallback)");
    if (device < 0 || device >= ndev || device >= 64) return fail(Q1ENV_ERR_INVALID_ARG, "q1phys_apply_host: bad device index");
    DeviceGuard guard(device);
    std::lock_guard<std::mutex> lock(g_apply_mutex);
    ApplyCtx& cx = g_apply_ctx[device];
    const size_t n = (size_t)n64;
    const size_t b8 = align_up(n * 8, 256), b1 = align_up(n, 256), bv = align_up(n * 3 * sizeof(VT), 256);
    // block layout, inputs then outputs: yaw pitch roll fmove smove dt z | button2 on_ground jump_released | vel || out_z out_vel out_og out_jr
    const size_t in_bytes = 7 * b8 + 3 * b1 + bv, out_bytes = b8 + bv + 2 * b1, total = in_bytes + out_bytes;
    if (!cx.stream) HIP_TRY(hipStreamCreateWithFlags(&cx.stream, hipStreamNonBlocking));
    if (total > cx.bytes) {
        if (cx.dev) (void)hipFree(cx.dev);
        if (cx.pin) (void)hipHostFree(cx.pin);
        cx.dev = nullptr; cx.pin = nullptr; cx.bytes = 0;
        const size_t want = total + total / 2;
        HIP_TRY(hipMalloc((void**)&cx.dev, want));
        HIP_TRY(hipHostMalloc((void**)&cx.pin, want, hipHostMallocDefault));
        cx.bytes = want;
    }
    char* pin = cx.pin;
    char* d = cx.dev;
    const size_t o_pitch = b8, o_roll = 2 * b8, o_f = 3 * b8, o_s = 4 * b8, o_dt = 5 * b8, o_z = 6 * b8;
    const size_t o_b2 = 7 * b8, o_og = o_b2 + b1, o_jr = o_og + b1, o_v = o_jr + b1;
    const size_t o_oz = in_bytes, o_ov = o_oz + b8, o_oog = o_ov + bv, o_ojr = o_oog + b1;
    memcpy(pin, yaw, n * 8);
    if (pitch) memcpy(pin + o_pitch, pitch, n * 8);
    if (roll) memcpy(pin + o_roll, roll, n * 8);
    memcpy(pin + o_f, fmove, n * 8); memcpy(pin + o_s, smove, n * 8); memcpy(pin + o_dt, time_delta, n * 8);
    memcpy(pin + o_z, z_pos, n * 8);
    memcpy(pin + o_b2, button2, n); memcpy(pin + o_og, on_ground, n); memcpy(pin + o_jr, jump_released, n);
    memcpy(pin + o_v, vel, n * 3 * sizeof(VT));
    HIP_TRY(hipMemcpyAsync(d, pin, in_bytes, hipMemcpyHostToDevice, cx.stream));
    hipLaunchKernelGGL(phys_apply_kernel<VT>, grid_for((int)n, 256), dim3(256), 0, cx.stream, (int)n, (const double*)d,
                       pitch ? (const double*)(d + o_pitch) : (const double*)nullptr,
                       roll ? (const double*)(d + o_roll) : (const double*)nullptr, (const double*)(d + o_f),
                       (const double*)(d + o_s), (const uint8_t*)(d + o_b2), (const double*)(d + o_dt), (const double*)(d + o_z),
                       (const VT*)(d + o_v), (const uint8_t*)(d + o_og), (const uint8_t*)(d + o_jr), (double*)(d + o_oz),
                       (VT*)(d + o_ov), (uint8_t*)(d + o_oog), (uint8_t*)(d + o_ojr));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(pin + in_bytes, d + in_bytes, out_bytes, hipMemcpyDeviceToHost, cx.stream));
    HIP_TRY(hipStreamSynchronize(cx.stream));
    memcpy(out_z, pin + o_oz, n * 8);
    memcpy(out_vel, pin + o_ov, n * 3 * sizeof(VT));
    memcpy(out_og, pin + o_oog, n);
    memcpy(out_jr, pin + o_ojr, n);
    return Q1ENV_OK;
}

extern "C" {

int q1env_abi_version(void) { return Q1ENV_ABI_VERSION; }

const char* q1env_last_error(void) { return g_err.c_str(); }

int q1env_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(Q1ENV_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int q1env_create(const q1env_config* cfg, int device, void* stream, q1env_t** out) {
    if (!cfg || !out) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_create: null argument");
    *out = nullptr;
    Params p{};
    std::string why;
    if (make_params(*cfg, p, why) != 0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_create: " + why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(Q1ENV_ERR_NO_DEVICE, "q1env_create: no HIP device visible (libq1env has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_create: bad device index");
    DeviceGuard guard(device);
    q1env* h = new (std::nothrow) q1env();
    if (!h) return fail(Q1ENV_ERR_ALLOC, "q1env_create: out of host memory");
    h->cfg = *cfg;
    h->p = p;
    h->device = device;
    if (stream) { h->stream = (hipStream_t)stream; h->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return fail(Q1ENV_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
        h->own_stream = true;
    }
    hipError_t e = hipMalloc(&h->arena, arena_bytes((size_t)p.n));
    if (e != hipSuccess) {
        if (h->own_stream) (void)hipStreamDestroy(h->stream);
        delete h;
        return fail(Q1ENV_ERR_ALLOC, std::string("hipMalloc(state): ") + hipGetErrorString(e));
    }
    carve(h);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->num_cus = cus;
    }
    (void)hipEventCreate(&h->ev0);
    (void)hipEventCreate(&h->ev1);
    // zero-start reset of every env: mask NULL, zero_start_prob forced to 1 for this launch
    Params p0 = p;
    p0.zero_start_prob = 2.0;
    const int b = block_for(p.n);
    hipLaunchKernelGGL(reset_philox_kernel<float>, grid_for(p.n, b), dim3(b), 0, h->stream, p0, h->st,
                       (uint64_t)0, (uint64_t)0, (const uint64_t*)nullptr, (const uint8_t*)nullptr, 0, (float*)nullptr);
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { q1env_destroy(h); return fail(Q1ENV_ERR_HIP, std::string("initial reset: ") + hipGetErrorString(e)); }
    *out = h;
    return Q1ENV_OK;
}

int q1env_destroy(q1env_t* h) {
    if (!h) return Q1ENV_OK;
    DeviceGuard guard(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (auto& ge : h->graphs) (void)hipGraphExecDestroy(ge.exec);
    h->graphs.clear();
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stage) (void)hipFree(h->stage);
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->snap) (void)hipFree(h->snap);
    if (h->arena) (void)hipFree(h->arena);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return Q1ENV_OK;
}

int q1env_set_stream(q1env_t* h, void* stream) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_set_stream: null handle");
    DeviceGuard guard(h->device);
    for (auto& ge : h->graphs) (void)hipGraphExecDestroy(ge.exec);
    h->graphs.clear();
    if (h->own_stream) { (void)hipStreamDestroy(h->stream); h->own_stream = false; }
    h->stream = (hipStream_t)stream;          // NULL = the device's default (null) stream
    return Q1ENV_OK;
}

int q1env_sync(q1env_t* h) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sync: null handle");
    DeviceGuard guard(h->device);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

int q1env_num_keys(const q1env_t* h) { return h ? h->p.num_keys : fail(Q1ENV_ERR_INVALID_ARG, "null handle"); }
int q1env_action_width(const q1env_t* h) { return h ? h->p.act_width : fail(Q1ENV_ERR_INVALID_ARG, "null handle"); }

static int check_act(const q1env* h, int fmt, const void* a, const void* b, bool allow_random) {
    if (fmt == Q1ENV_ACT_RANDOM) return allow_random ? 0 : fail(Q1ENV_ERR_INVALID_ARG, "Q1ENV_ACT_RANDOM is rollout-only");
    if (fmt < 0 || fmt > 2) return fail(Q1ENV_ERR_INVALID_ARG, "unknown action_format");
    if (!a) return fail(Q1ENV_ERR_INVALID_ARG, "act_a is NULL");
    if (fmt == Q1ENV_ACT_PACKED && h->p.yaw_mode && !b) return fail(Q1ENV_ERR_INVALID_ARG, "packed actions need act_b (mouse)");
    return 0;
}

static size_t act_bytes_a(const q1env* h, int fmt) {
    const size_t n = (size_t)h->p.n;
    if (fmt == Q1ENV_ACT_F64_ROWS) return n * h->p.act_width * 8;
    if (fmt == Q1ENV_ACT_F32_ROWS) return n * h->p.act_width * 4;
    return n;
}

// Width of one row of policy-network outputs (Q1PhysActionDist.required_model_output_shape, action_dist.py:236-241): two logits per
// key, then (mean, log_std) of the continuous mouse or the 2S+1 logits of the discrete one.
static int policy_row_width(const Params& p) {
    return 2 * p.num_keys + (p.yaw_mode == 1 ? 2 : (p.yaw_mode == 2 ? 2 * (int)p.yaw_steps + 1 : 0));
}

// The default action/episode structure (4 keys, continuous mouse, jump key, no hover, y reward) runs the SPEC kernels.
static bool is_spec(const Params& p) {
    return p.num_keys == 4 && p.yaw_mode == 1 && p.jump_mode == 1 && !p.hover && !p.speed_reward;
}

static void launch_step(q1env* h, int fmt, const void* a, const void* b, int obs_format, void* obs,
                        float* reward, uint8_t* done, uint8_t* zs) {
    const int blk = block_for(h->p.n);
    const dim3 g = grid_for(h->p.n, blk), bs(blk);
    const bool spec = is_spec(h->p);
#define Q1_LAUNCH_STEP(OT, SP, FM) \
    hipLaunchKernelGGL((step_kernel<OT, SP, FM>), g, bs, 0, h->stream, h->st.vx, h->st.vy, h->st.vz, h->st.px, h->st.py, h->st.z, h->st.yaw, \
                       h->st.trem, h->p, h->st, fmt, a, b, (OT*)obs, reward, done, zs)
    if (obs_format == Q1ENV_OBS_F32) {
        if (spec && fmt == Q1ENV_ACT_PACKED) Q1_LAUNCH_STEP(float, true, FMT_PACKED);
        else if (spec && fmt == Q1ENV_ACT_F32_ROWS) Q1_LAUNCH_STEP(float, true, FMT_F32_ROWS);
        else Q1_LAUNCH_STEP(float, false, FMT_RUNTIME);
    } else {
        if (spec && fmt == Q1ENV_ACT_F64_ROWS) Q1_LAUNCH_STEP(double, true, FMT_F64_ROWS);
        else Q1_LAUNCH_STEP(double, false, FMT_RUNTIME);
    }
#undef Q1_LAUNCH_STEP
}

int q1env_step(q1env_t* h, int fmt, const void* a, const void* b, int obs_format, void* obs, float* reward,
               uint8_t* done, uint8_t* zs) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step: null handle");
    DeviceGuard guard(h->device);
    if (int r = check_act(h, fmt, a, b, false)) return r;
    if (obs_format != Q1ENV_OBS_F32 && obs_format != Q1ENV_OBS_F64) return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    launch_step(h, fmt, a, b, obs_format, obs, reward, done, zs);
    HIP_TRY(hipGetLastError());
    h->tick_count += 1;
    return Q1ENV_OK;
}

static void launch_step_autoreset(q1env* h, int fmt, const void* a, const void* b, uint64_t seed, uint64_t counter,
                                  const uint64_t* counter_dev, float* obs, float* reward, uint8_t* done, uint8_t* zs) {
    const int blk = block_for(h->p.n);
    const dim3 g = grid_for(h->p.n, blk), bs(blk);
#define Q1_LAUNCH_AR(SP, FM) \
    hipLaunchKernelGGL((step_autoreset_kernel<SP, FM>), g, bs, 0, h->stream, h->p, h->st, fmt, a, b, seed, counter, counter_dev, obs, reward, done, zs)
    if (is_spec(h->p) && fmt == Q1ENV_ACT_PACKED) Q1_LAUNCH_AR(true, FMT_PACKED);
    else Q1_LAUNCH_AR(false, FMT_RUNTIME);
#undef Q1_LAUNCH_AR
}

int q1env_step_autoreset(q1env_t* h, int fmt, const void* a, const void* b, uint64_t seed, const uint64_t* counter_dev, float* obs,
                         float* reward, uint8_t* done, uint8_t* zs) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_autoreset: null handle");
    DeviceGuard guard(h->device);
    if (int r = check_act(h, fmt, a, b, false)) return r;
    launch_step_autoreset(h, fmt, a, b, seed, counter_dev ? 0 : h->tick_count, counter_dev, obs, reward, done, zs);
    HIP_TRY(hipGetLastError());
    h->tick_count += 1;
    return Q1ENV_OK;
}

// `ticks` auto-reset ticks over tick-major actions as a replayable unit: tick t uses the Philox counter *counter_dev + t, and one
// last node advances *counter_dev by `ticks` - so a cached graph draws fresh reset randomness on every replay.
static void enqueue_many_autoreset(q1env* h, int ticks, int fmt, const void* a, const void* b, uint64_t seed, uint64_t* counter_dev,
                                   float* obs, float* reward, uint8_t* done, uint8_t* zs, int out_stride) {
    const size_t n = (size_t)h->p.n;
    const size_t sa = act_bytes_a(h, fmt), sb = n * 4;
    for (int t = 0; t < ticks; ++t) {
        const size_t ot = out_stride ? (size_t)t : 0;
        launch_step_autoreset(h, fmt, (const char*)a + sa * t, b ? (const char*)b + sb * t : nullptr, seed, (uint64_t)t, counter_dev,
                              obs ? obs + n * 6 * ot : nullptr, reward ? reward + n * ot : nullptr, done ? done + n * ot : nullptr,
                              zs ? zs + n * ot : nullptr);
    }
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, h->stream, counter_dev, (uint64_t)ticks);
}

int q1env_step_autoreset_many(q1env_t* h, int ticks, int fmt, const void* a, const void* b, uint64_t seed, uint64_t* counter_dev,
                              float* obs, float* reward, uint8_t* done, uint8_t* zs, int out_stride, int use_graph) {
    if (!h || !counter_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_autoreset_many: null argument (counter_dev is required)");
    DeviceGuard guard(h->device);
    if (ticks <= 0) return fail(Q1ENV_ERR_INVALID_ARG, "ticks must be > 0");
    if (int r = check_act(h, fmt, a, b, false)) return r;
    if (use_graph < 0 || use_graph > 2) return fail(Q1ENV_ERR_INVALID_ARG, "bad use_graph");
    if (!use_graph) {
        enqueue_many_autoreset(h, ticks, fmt, a, b, seed, counter_dev, obs, reward, done, zs, out_stride);
        HIP_TRY(hipGetLastError());
    } else {
        std::vector<uint64_t> key = {0xA17053E7ull, (uint64_t)ticks, (uint64_t)fmt, (uint64_t)(uintptr_t)a, (uint64_t)(uintptr_t)b, seed,
                                     (uint64_t)(uintptr_t)counter_dev, (uint64_t)(uintptr_t)obs, (uint64_t)(uintptr_t)reward,
                                     (uint64_t)(uintptr_t)done, (uint64_t)(uintptr_t)zs, (uint64_t)out_stride};
        hipGraphExec_t exec = nullptr;
        for (auto& ge : h->graphs)
            if (ge.key == key) { exec = ge.exec; break; }
        if (!exec) {
            hipGraph_t g = nullptr;
            if (!h->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
            hipStream_t launch_stream = h->stream;
            h->stream = h->cap_stream;
            hipError_t ce = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal);
            if (ce == hipSuccess) {
                enqueue_many_autoreset(h, ticks, fmt, a, b, seed, counter_dev, obs, reward, done, zs, out_stride);
                ce = hipStreamEndCapture(h->cap_stream, &g);
            }
            h->stream = launch_stream;
            if (ce != hipSuccess) return fail(Q1ENV_ERR_HIP, std::string("graph capture: ") + hipGetErrorString(ce));
            hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) return fail(Q1ENV_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
            if (h->graphs.size() >= 8) {
                (void)hipGraphExecDestroy(h->graphs.front().exec);
                h->graphs.erase(h->graphs.begin());
            }
            h->graphs.push_back({key, exec});
        }
        if (use_graph == 2) {
            (void)hipGraphUpload(exec, h->stream);
            return Q1ENV_OK;
        }
        HIP_TRY(hipGraphLaunch(exec, h->stream));
    }
    h->tick_count += (uint64_t)ticks;
    return Q1ENV_OK;
}

int q1env_step_host(q1env_t* h, int fmt, const void* a, const void* b, int obs_format, void* obs, float* reward,
                    uint8_t* done, uint8_t* zs) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_host: null handle");
    if (int r = check_act(h, fmt, a, b, false)) return r;
    if (obs_format != Q1ENV_OBS_F32 && obs_format != Q1ENV_OBS_F64) return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    DeviceGuard guard(h->device);
    const size_t n = (size_t)h->p.n;
    const size_t na = act_bytes_a(h, fmt), nb = (fmt == Q1ENV_ACT_PACKED && h->p.yaw_mode) ? n * 4 : 0;
    const size_t no = n * 6 * (obs_format == Q1ENV_OBS_F32 ? 4 : 8);
    const size_t ba = align_up(na, 256), bb = align_up(n * 4, 256), bo = align_up(no, 256);
    const size_t br = align_up(n * 4, 256), bd = align_up(n, 256);
    const size_t in_bytes = ba + bb, out_bytes = bo + br + 2 * bd;
    if (int r = ensure_stage(h, in_bytes + out_bytes)) return r;
    char* d = (char*)h->stage;
    void* d_a = d; void* d_b = d + ba; void* d_o = d + in_bytes;
    float* d_r = (float*)(d + in_bytes + bo); uint8_t* d_d = (uint8_t*)(d + in_bytes + bo + br); uint8_t* d_z = d_d + bd;
    const bool pack = n <= PACK_MAX_ENVS;
    char* pin = nullptr;
    if (pack) {                                   // one H2D block, one D2H block through the handle's pinned staging
        if (int r = ensure_pin(h, in_bytes + out_bytes)) return r;
        pin = (char*)h->pin;
        memcpy(pin, a, na);
        if (nb) memcpy(pin + ba, b, nb);
        HIP_TRY(hipMemcpyAsync(d, pin, nb ? ba + nb : na, hipMemcpyHostToDevice, h->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(d_a, a, na, hipMemcpyHostToDevice, h->stream));
        if (nb) HIP_TRY(hipMemcpyAsync(d_b, b, nb, hipMemcpyHostToDevice, h->stream));
    }
    launch_step(h, fmt, d_a, d_b, obs_format, obs ? d_o : nullptr, reward ? d_r : nullptr, done ? d_d : nullptr, zs ? d_z : nullptr);
    HIP_TRY(hipGetLastError());
    h->tick_count += 1;
    if (pack) {
        HIP_TRY(hipMemcpyAsync(pin + in_bytes, d + in_bytes, out_bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        const char* po = pin + in_bytes;
        if (obs) memcpy(obs, po, no);
        if (reward) memcpy(reward, po + bo, n * 4);
        if (done) memcpy(done, po + bo + br, n);
        if (zs) memcpy(zs, po + bo + br + bd, n);
        return Q1ENV_OK;
    }
    if (obs) HIP_TRY(hipMemcpyAsync(obs, d_o, no, hipMemcpyDeviceToHost, h->stream));
    if (reward) HIP_TRY(hipMemcpyAsync(reward, d_r, n * 4, hipMemcpyDeviceToHost, h->stream));
    if (done) HIP_TRY(hipMemcpyAsync(done, d_d, n, hipMemcpyDeviceToHost, h->stream));
    if (zs) HIP_TRY(hipMemcpyAsync(zs, d_z, n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

static void enqueue_many(q1env* h, int ticks, int fmt, const void* a, const void* b, int obs_format, void* obs,
                         float* reward, uint8_t* done, int out_stride) {
    const size_t n = (size_t)h->p.n;
    const size_t sa = act_bytes_a(h, fmt), sb = n * 4;
    const size_t so = n * 6 * (obs_format == Q1ENV_OBS_F32 ? 4 : 8);
    for (int t = 0; t < ticks; ++t) {
        const size_t ot = out_stride ? (size_t)t : 0;
        launch_step(h, fmt, (const char*)a + sa * t, b ? (const char*)b + sb * t : nullptr, obs_format,
                    obs ? (char*)obs + so * ot : nullptr, reward ? reward + n * ot : nullptr,
                    done ? done + n * ot : nullptr, nullptr);
    }
}

int q1env_step_many(q1env_t* h, int ticks, int fmt, const void* a, const void* b, int obs_format, void* obs,
                    float* reward, uint8_t* done, int out_stride, int use_graph) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_step_many: null handle");
    DeviceGuard guard(h->device);
    if (ticks <= 0) return fail(Q1ENV_ERR_INVALID_ARG, "ticks must be > 0");
    if (int r = check_act(h, fmt, a, b, false)) return r;
    if (obs_format != Q1ENV_OBS_F32 && obs_format != Q1ENV_OBS_F64) return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    const bool t_start = (use_graph & Q1ENV_TIMER_START) != 0;    // record the handle's timer events around the launches
    const bool t_stop = (use_graph & Q1ENV_TIMER_STOP) != 0;
    use_graph &= ~(Q1ENV_TIMER_START | Q1ENV_TIMER_STOP);
    if (use_graph < 0 || use_graph > 2) return fail(Q1ENV_ERR_INVALID_ARG, "bad use_graph");
    if (!use_graph) {
        if (t_start) HIP_TRY(hipEventRecord(h->ev0, h->stream));
        enqueue_many(h, ticks, fmt, a, b, obs_format, obs, reward, done, out_stride);
        HIP_TRY(hipGetLastError());
        if (t_stop) HIP_TRY(hipEventRecord(h->ev1, h->stream));
    } else {
        std::vector<uint64_t> key = {(uint64_t)ticks, (uint64_t)fmt, (uint64_t)(uintptr_t)a, (uint64_t)(uintptr_t)b,
                                     (uint64_t)obs_format, (uint64_t)(uintptr_t)obs, (uint64_t)(uintptr_t)reward,
                                     (uint64_t)(uintptr_t)done, (uint64_t)out_stride};
        hipGraphExec_t exec = nullptr;
        for (auto& ge : h->graphs)
            if (ge.key == key) { exec = ge.exec; break; }
        if (!exec) {
            hipGraph_t g = nullptr;
            if (!h->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
            hipStream_t launch_stream = h->stream;
            h->stream = h->cap_stream;                      // record the launches on the capture stream ...
            hipError_t ce = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal);
            if (ce == hipSuccess) {
                enqueue_many(h, ticks, fmt, a, b, obs_format, obs, reward, done, out_stride);
                ce = hipStreamEndCapture(h->cap_stream, &g);
            }
            h->stream = launch_stream;                      // ... and replay them on the handle's own stream
            if (ce != hipSuccess) return fail(Q1ENV_ERR_HIP, std::string("graph capture: ") + hipGetErrorString(ce));
            hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) return fail(Q1ENV_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
            if (h->graphs.size() >= 8) {                    // small cache: evict the oldest entry
                (void)hipGraphExecDestroy(h->graphs.front().exec);
                h->graphs.erase(h->graphs.begin());
            }
            h->graphs.push_back({key, exec});
        }
        if (use_graph == 2) {                               // prepare only: capture + instantiate + upload, no launch, no tick
            (void)hipGraphUpload(exec, h->stream);          // the executable graph's packets are resident before the first replay
            return Q1ENV_OK;
        }
        if (t_start) HIP_TRY(hipEventRecord(h->ev0, h->stream));
        HIP_TRY(hipGraphLaunch(exec, h->stream));
        if (t_stop) HIP_TRY(hipEventRecord(h->ev1, h->stream));
    }
    h->tick_count += (uint64_t)ticks;
    return Q1ENV_OK;
}

int q1env_rollout(q1env_t* h, int ticks, int fmt, const void* a, const void* b, uint64_t seed, int obs_format,
                  void* obs, float* reward, uint8_t* done, int auto_reset, double* return_sum) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_rollout: null handle");
    DeviceGuard guard(h->device);
    if (ticks <= 0) return fail(Q1ENV_ERR_INVALID_ARG, "ticks must be > 0");
    if (int r = check_act(h, fmt, a, b, true)) return r;
    if (obs_format != Q1ENV_OBS_F32 && obs_format != Q1ENV_OBS_F64) return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    const int blk = block_for(h->p.n);
    const dim3 g = grid_for(h->p.n, blk), bs(blk);
    const bool spec = is_spec(h->p);
#define Q1_LAUNCH_ROLL(OT, SP, FM, HR, OM)                                                                           \
    hipLaunchKernelGGL((rollout_kernel<OT, SP, FM, HR, OM>), g, bs, 0, h->stream, h->p, h->st, ticks, fmt, a, b, seed, \
                       h->tick_count, (OT*)obs, reward, done, auto_reset, return_sum)
    const bool all_out = obs && reward && done, no_out = !obs && !reward && !done;
    if (obs_format == Q1ENV_OBS_F32 && spec && (all_out || no_out) &&
        (fmt == Q1ENV_ACT_PACKED || fmt == Q1ENV_ACT_RANDOM)) {
        const int which = (fmt == Q1ENV_ACT_RANDOM ? 4 : 0) + (auto_reset ? 2 : 0) + (all_out ? 1 : 0);
        switch (which) {
            case 0: Q1_LAUNCH_ROLL(float, true, FMT_PACKED, false, 0); break;
            case 1: Q1_LAUNCH_ROLL(float, true, FMT_PACKED, false, 1); break;
            case 2: Q1_LAUNCH_ROLL(float, true, FMT_PACKED, true, 0); break;
            case 3: Q1_LAUNCH_ROLL(float, true, FMT_PACKED, true, 1); break;
            case 4: Q1_LAUNCH_ROLL(float, true, FMT_RANDOM, false, 0); break;
            case 5: Q1_LAUNCH_ROLL(float, true, FMT_RANDOM, false, 1); break;
            case 6: Q1_LAUNCH_ROLL(float, true, FMT_RANDOM, true, 0); break;
            default: Q1_LAUNCH_ROLL(float, true, FMT_RANDOM, true, 1); break;
        }
    } else if (obs_format == Q1ENV_OBS_F32) {
        if (spec && fmt == Q1ENV_ACT_F32_ROWS) Q1_LAUNCH_ROLL(float, true, FMT_F32_ROWS, true, -1);
        else Q1_LAUNCH_ROLL(float, false, FMT_RUNTIME, true, -1);
    } else {
        Q1_LAUNCH_ROLL(double, false, FMT_RUNTIME, true, -1);
    }
#undef Q1_LAUNCH_ROLL
    HIP_TRY(hipGetLastError());
    h->tick_count += (uint64_t)ticks;
    return Q1ENV_OK;
}

int q1env_observe(q1env_t* h, int obs_format, void* obs) {
    if (!h || !obs) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_observe: null argument");
    DeviceGuard guard(h->device);
    const int blk = block_for(h->p.n);
    if (obs_format == Q1ENV_OBS_F32)
        hipLaunchKernelGGL(observe_kernel<float>, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, (float*)obs);
    else if (obs_format == Q1ENV_OBS_F64)
        hipLaunchKernelGGL(observe_kernel<double>, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, (double*)obs);
    else return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_observe_host(q1env_t* h, int obs_format, void* obs) {
    if (!h || !obs) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_observe_host: null argument");
    DeviceGuard guard(h->device);
    const size_t bytes = (size_t)h->p.n * 6 * (obs_format == Q1ENV_OBS_F32 ? 4 : 8);
    if (int r = ensure_stage(h, bytes)) return r;
    if (int r = q1env_observe(h, obs_format, h->stage)) return r;
    HIP_TRY(hipMemcpyAsync(obs, h->stage, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

int q1env_reset_draws_host(q1env_t* h, int64_t count, const int32_t* idx, const uint8_t* zero_start, const double* yaw,
                           const double* tm, const double* speed, const double* angle, int obs_format, void* obs) {
    if (!h || !zero_start || !yaw || !tm || !speed || !angle) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_reset_draws_host: null argument");
    if (count <= 0 || count > h->p.n) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_reset_draws_host: count must be in 1..num_envs");
    if (obs_format != Q1ENV_OBS_F32 && obs_format != Q1ENV_OBS_F64) return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    if (idx) for (int64_t j = 0; j < count; ++j)
        if (idx[j] < 0 || idx[j] >= h->p.n) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_reset_draws_host: index out of range");
    DeviceGuard guard(h->device);
    const size_t c = (size_t)count;
    const size_t bi = align_up(c * 4, 256), bz = align_up(c, 256), bd = align_up(c * 8, 256);
    const size_t no = c * 6 * (obs_format == Q1ENV_OBS_F32 ? 4 : 8), bo = align_up(no, 256);
    const size_t in_bytes = bi + bz + 4 * bd;
    if (int r = ensure_stage(h, in_bytes + bo)) return r;
    if (int r = ensure_pin(h, in_bytes + (c <= PACK_MAX_ENVS ? bo : 0))) return r;
    char* d = (char*)h->stage;
    char* pin = (char*)h->pin;
    int32_t* d_i = (int32_t*)d; uint8_t* d_z = (uint8_t*)(d + bi);
    double* d_y = (double*)(d + bi + bz); double* d_t = (double*)(d + bi + bz + bd);
    double* d_s = (double*)(d + bi + bz + 2 * bd); double* d_a = (double*)(d + bi + bz + 3 * bd);
    void* d_o = d + in_bytes;
    // the six input arrays travel as ONE block through the pinned staging (a reset_at is 1 env: six tiny copies were six calls)
    if (idx) memcpy(pin, idx, c * 4);
    memcpy(pin + bi, zero_start, c);
    memcpy(pin + bi + bz, yaw, c * 8);
    memcpy(pin + bi + bz + bd, tm, c * 8);
    memcpy(pin + bi + bz + 2 * bd, speed, c * 8);
    memcpy(pin + bi + bz + 3 * bd, angle, c * 8);
    HIP_TRY(hipMemcpyAsync(d, pin, in_bytes, hipMemcpyHostToDevice, h->stream));
    const int blk = 64;
    if (obs_format == Q1ENV_OBS_F32)
        hipLaunchKernelGGL(reset_draws_kernel<float>, grid_for((int)count, blk), dim3(blk), 0, h->stream, h->p, h->st, (int)count,
                           idx ? d_i : nullptr, d_z, d_y, d_t, d_s, d_a, obs ? (float*)d_o : nullptr);
    else
        hipLaunchKernelGGL(reset_draws_kernel<double>, grid_for((int)count, blk), dim3(blk), 0, h->stream, h->p, h->st, (int)count,
                           idx ? d_i : nullptr, d_z, d_y, d_t, d_s, d_a, obs ? (double*)d_o : nullptr);
    HIP_TRY(hipGetLastError());
    if (obs && c <= PACK_MAX_ENVS) {
        HIP_TRY(hipMemcpyAsync(pin + in_bytes, d_o, no, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(obs, pin + in_bytes, no);
        return Q1ENV_OK;
    }
    if (obs) HIP_TRY(hipMemcpyAsync(obs, d_o, no, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

int q1env_reset_philox(q1env_t* h, uint64_t seed, const uint64_t* counter_dev, const uint8_t* mask, int done_only, int obs_format,
                       void* obs) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_reset_philox: null handle");
    DeviceGuard guard(h->device);
    const int blk = block_for(h->p.n);
    const uint64_t counter = counter_dev ? 0 : h->tick_count;
    if (obs_format == Q1ENV_OBS_F32)
        hipLaunchKernelGGL(reset_philox_kernel<float>, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, seed,
                           counter, counter_dev, mask, done_only, (float*)obs);
    else if (obs_format == Q1ENV_OBS_F64)
        hipLaunchKernelGGL(reset_philox_kernel<double>, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, seed,
                           counter, counter_dev, mask, done_only, (double*)obs);
    else return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

static int copy_state(q1env* h, const q1env_state* s, bool to_host) {
    DeviceGuard guard(h->device);
    const size_t n = (size_t)h->p.n;
    struct Item { void* host; void* dev; size_t bytes; };
    const Item items[] = {
        {s->vel_x, h->st.vx, n * 4}, {s->vel_y, h->st.vy, n * 4}, {s->vel_z, h->st.vz, n * 4},
        {s->pos_x, h->st.px, n * 8}, {s->pos_y, h->st.py, n * 8}, {s->z_pos, h->st.z, n * 8},
        {s->yaw, h->st.yaw, n * 8}, {s->time_remaining, h->st.trem, n * 8},
        {s->last_key_press_time, h->st.lk, n * 32}, {s->flags, h->st.flags, n},
    };
    int wanted = 0;
    for (const Item& it : items) wanted += it.host != nullptr;
    if (to_host && n <= PACK_MAX_ENVS && wanted > 2) {
        // the SoA arrays are one contiguous arena: one copy of it through the pinned staging instead of one copy per array
        const size_t bytes = arena_bytes(n);
        if (int r = ensure_pin(h, bytes)) return r;
        HIP_TRY(hipMemcpyAsync(h->pin, h->arena, bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (const Item& it : items)
            if (it.host) memcpy(it.host, (const char*)h->pin + ((const char*)it.dev - (const char*)h->arena), it.bytes);
        return Q1ENV_OK;
    }
    for (const Item& it : items) {
        if (!it.host) continue;
        if (to_host) HIP_TRY(hipMemcpyAsync(it.host, it.dev, it.bytes, hipMemcpyDeviceToHost, h->stream));
        else HIP_TRY(hipMemcpyAsync(it.dev, it.host, it.bytes, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

int q1env_get_state_host(q1env_t* h, const q1env_state* dst) {
    if (!h || !dst) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_get_state_host: null argument");
    return copy_state(h, dst, true);
}

int q1env_set_state_host(q1env_t* h, const q1env_state* src) {
    if (!h || !src) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_set_state_host: null argument");
    return copy_state(h, src, false);
}

int q1env_snapshot_state(q1env_t* h) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_snapshot_state: null handle");
    DeviceGuard guard(h->device);
    const size_t bytes = arena_bytes((size_t)h->p.n);
    if (!h->snap) HIP_TRY(hipMalloc(&h->snap, bytes));
    HIP_TRY(hipMemcpyAsync(h->snap, h->arena, bytes, hipMemcpyDeviceToDevice, h->stream));
    return Q1ENV_OK;
}

int q1env_restore_state(q1env_t* h) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_restore_state: null handle");
    if (!h->snap) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_restore_state: no snapshot taken");
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(h->arena, h->snap, arena_bytes((size_t)h->p.n), hipMemcpyDeviceToDevice, h->stream));
    return Q1ENV_OK;
}

int q1env_state_device_ptrs(q1env_t* h, q1env_state* out) {
    if (!h || !out) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_state_device_ptrs: null argument");
    out->vel_x = h->st.vx; out->vel_y = h->st.vy; out->vel_z = h->st.vz;
    out->pos_x = h->st.px; out->pos_y = h->st.py; out->z_pos = h->st.z;
    out->yaw = h->st.yaw; out->time_remaining = h->st.trem;
    out->last_key_press_time = h->st.lk; out->flags = h->st.flags;
    return Q1ENV_OK;
}

int q1env_decode_host(q1env_t* h, int fmt, const void* a, const void* b, const float* z_vel, const double* trem,
                      double* yaw, int64_t* smove, int64_t* fmove, uint8_t* jump) {
    if (!h || !z_vel || !trem || !yaw || !smove || !fmove || !jump) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_decode_host: null argument");
    if (int r = check_act(h, fmt, a, b, false)) return r;
    DeviceGuard guard(h->device);
    const size_t n = (size_t)h->p.n;
    const size_t ba = align_up(act_bytes_a(h, fmt), 256), b4 = align_up(n * 4, 256), b8 = align_up(n * 8, 256), b1 = align_up(n, 256);
    if (int r = ensure_stage(h, ba + 2 * b4 + 4 * b8 + b1)) return r;
    char* d = (char*)h->stage;
    void* d_a = d; void* d_b = d + ba; float* d_zv = (float*)(d + ba + b4);
    double* d_tr = (double*)(d + ba + 2 * b4); double* d_y = d_tr + b8 / 8;
    int64_t* d_sm = (int64_t*)(d_y + b8 / 8); int64_t* d_fm = d_sm + b8 / 8; uint8_t* d_j = (uint8_t*)(d_fm + b8 / 8);
    const bool pack = n <= PACK_MAX_ENVS;                  // mkdemo-style per-frame use is n = 1: one copy each way, not eight
    const size_t in_bytes = ba + 2 * b4 + b8, total = ba + 2 * b4 + 4 * b8 + b1;
    const bool has_b = fmt == Q1ENV_ACT_PACKED && h->p.yaw_mode;
    char* pin = nullptr;
    if (pack) {
        if (int r = ensure_pin(h, total)) return r;
        pin = (char*)h->pin;
        memcpy(pin, a, act_bytes_a(h, fmt));
        if (has_b) memcpy(pin + ba, b, n * 4);
        memcpy(pin + ba + b4, z_vel, n * 4);
        memcpy(pin + ba + 2 * b4, trem, n * 8);
        HIP_TRY(hipMemcpyAsync(d, pin, in_bytes, hipMemcpyHostToDevice, h->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(d_a, a, act_bytes_a(h, fmt), hipMemcpyHostToDevice, h->stream));
        if (has_b) HIP_TRY(hipMemcpyAsync(d_b, b, n * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(d_zv, z_vel, n * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(d_tr, trem, n * 8, hipMemcpyHostToDevice, h->stream));
    }
    const int blk = block_for(h->p.n);
    hipLaunchKernelGGL(decode_kernel, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, fmt, (const void*)d_a,
                       (const void*)d_b, (const float*)d_zv, (const double*)d_tr, d_y, d_sm, d_fm, d_j);
    HIP_TRY(hipGetLastError());
    if (pack) {
        HIP_TRY(hipMemcpyAsync(pin + in_bytes, d + in_bytes, total - in_bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        const char* po = pin + in_bytes;
        memcpy(yaw, po, n * 8); memcpy(smove, po + b8, n * 8); memcpy(fmove, po + 2 * b8, n * 8); memcpy(jump, po + 3 * b8, n);
        return Q1ENV_OK;
    }
    HIP_TRY(hipMemcpyAsync(yaw, d_y, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(smove, d_sm, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(fmove, d_fm, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(jump, d_j, n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

int q1env_decoder_reset_host(q1env_t* h, int64_t count, const int32_t* idx, const double* yaw) {
    if (!h || !yaw) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_decoder_reset_host: null argument");
    if (count <= 0 || count > h->p.n) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_decoder_reset_host: count must be in 1..num_envs");
    if (idx) for (int64_t j = 0; j < count; ++j)
        if (idx[j] < 0 || idx[j] >= h->p.n) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_decoder_reset_host: index out of range");
    DeviceGuard guard(h->device);
    const size_t c = (size_t)count;
    const size_t bi = align_up(c * 4, 256), bd = align_up(c * 8, 256);
    if (int r = ensure_stage(h, bi + bd)) return r;
    int32_t* d_i = (int32_t*)h->stage; double* d_y = (double*)((char*)h->stage + bi);
    if (idx) HIP_TRY(hipMemcpyAsync(d_i, idx, c * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_y, yaw, c * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(decoder_reset_kernel, grid_for((int)count, 64), dim3(64), 0, h->stream, h->p, h->st, (int)count,
                       idx ? (const int32_t*)d_i : (const int32_t*)nullptr, (const double*)d_y);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

int q1phys_apply_host(int device, int64_t n64, const double* yaw, const double* pitch, const double* roll, const double* fmove,
                      const double* smove, const uint8_t* button2, const double* time_delta, const double* z_pos,
                      const float* vel, const uint8_t* on_ground, const uint8_t* jump_released, double* out_z, float* out_vel,
                      uint8_t* out_og, uint8_t* out_jr) {
    return phys_apply_host_impl<float>(device, n64, yaw, pitch, roll, fmove, smove, button2, time_delta, z_pos, vel, on_ground,
                                       jump_released, out_z, out_vel, out_og, out_jr);
}

int q1phys_apply_host_f64(int device, int64_t n64, const double* yaw, const double* pitch, const double* roll, const double* fmove,
                          const double* smove, const uint8_t* button2, const double* time_delta, const double* z_pos,
                          const double* vel, const uint8_t* on_ground, const uint8_t* jump_released, double* out_z, double* out_vel,
                          uint8_t* out_og, uint8_t* out_jr) {
    return phys_apply_host_impl<double>(device, n64, yaw, pitch, roll, fmove, smove, button2, time_delta, z_pos, vel, on_ground,
                                        jump_released, out_z, out_vel, out_og, out_jr);
}

// Page-locked host memory for the arrays handed to the *_host entry points: copies to and from it are direct DMA (hipMemcpyAsync
// recognises the pointer), pageable arrays are staged by the runtime at a fraction of the rate (98 MB per tick at 1 M envs).
void* q1env_host_alloc(uint64_t bytes) {
    void* p = nullptr;
    if (bytes == 0) { (void)fail(Q1ENV_ERR_INVALID_ARG, "q1env_host_alloc: zero bytes"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)fail(Q1ENV_ERR_NO_DEVICE, "q1env_host_alloc: no HIP device visible"); return nullptr; }
    hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable);
    if (e != hipSuccess) { (void)fail(Q1ENV_ERR_ALLOC, std::string("hipHostMalloc: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}

int q1env_host_free(void* p) {
    if (!p) return Q1ENV_OK;
    HIP_TRY(hipHostFree(p));
    return Q1ENV_OK;
}

int q1env_policy_sample(q1env_t* h, const float* logits, int row_stride, uint64_t seed, uint64_t counter,
                        const uint64_t* counter_dev, int deterministic, uint8_t* keys, float* mouse, float* logp) {
    if (!h || !logits || !keys) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_sample: null argument");
    DeviceGuard guard(h->device);
    const int need = policy_row_width(h->p);
    if (row_stride < need) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_sample: row_stride smaller than the policy row (2*num_keys + 2, or + 2*discrete_yaw_steps+1)");
    if (h->p.yaw_mode != 0 && !mouse) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_sample: mouse output required");
    const int blk = block_for(h->p.n);
    hipLaunchKernelGGL(policy_sample_kernel, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, logits, row_stride, seed, counter,
                       counter_dev, deterministic, keys, mouse, logp);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_gae(q1env_t* h, int ticks, const float* reward, const float* value, const uint8_t* done, float gamma, float lam,
              float* adv, float* vtarg) {
    if (!h || !reward || !value || !done || !adv || !vtarg || ticks <= 0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_gae: bad argument");
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(gae_kernel, grid_for(h->p.n, 256), dim3(256), 0, h->stream, h->p.n, ticks, reward, value, done, gamma, lam, adv, vtarg);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_ppo_loss_grad(q1env_t* h, int64_t batch, const float* logits, const float* old_logits, int row_stride, const uint8_t* keys,
                        const float* mouse, const float* logp_old, const float* adv, const float* value, const float* value_old,
                        const float* vtarg, float clip_param, float vf_clip_param, float vf_loss_coeff, float entropy_coeff,
                        const float* kl_coeff_dev, float* dlogits, float* dvalue, float* partials) {
    if (!h || !logits || !old_logits || !keys || !logp_old || !adv || !value || !value_old || !vtarg || !kl_coeff_dev || !dlogits ||
        !dvalue || !partials)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_ppo_loss_grad: null argument");
    if (batch <= 0 || batch > (int64_t)1 << 30) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_ppo_loss_grad: bad batch");
    if (h->p.yaw_mode != 0 && !mouse) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_ppo_loss_grad: mouse actions required");
    if (row_stride < policy_row_width(h->p)) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_ppo_loss_grad: row_stride too small");
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(ppo_loss_grad_kernel, grid_for((int)batch, 256), dim3(256), 0, h->stream, h->p, (int)batch, logits, old_logits,
                       row_stride, keys, mouse, logp_old, adv, value, value_old, vtarg, clip_param, vf_clip_param, vf_loss_coeff,
                       entropy_coeff, kl_coeff_dev, dlogits, dvalue, partials);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_episode_stats(q1env_t* h, const float* reward, const uint8_t* done, const uint8_t* zero_start, double* ep_return,
                        double* partials) {
    if (!h || !reward || !done || !zero_start || !ep_return || !partials) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_episode_stats: null argument");
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(episode_stats_kernel, grid_for(h->p.n, 256), dim3(256), 0, h->stream, h->p.n, reward, done, zero_start, ep_return, partials);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_sample_step(q1env_t* h, const float* logits, int row_stride, uint64_t seed, const uint64_t* counter_dev,
                      uint64_t counter_offset, int deterministic, uint8_t* keys, float* mouse, float* logp, float* obs, float* reward,
                      uint8_t* done, uint8_t* zero_start, double* ep_return, double* partials) {
    if (!h || !logits || !keys || !obs || !reward || !done || !ep_return || !partials)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_step: null argument");
    DeviceGuard guard(h->device);
    const int need = policy_row_width(h->p);
    if (row_stride < need) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_step: row_stride smaller than the policy row (2*num_keys + 2, or + 2*discrete_yaw_steps+1)");
    if (h->p.yaw_mode != 0 && !mouse) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_step: mouse output required");
    const int blk = block_for(h->p.n);
    const dim3 g = grid_for(h->p.n, blk), bs(blk);
    const uint64_t counter = counter_offset + (counter_dev ? 0 : h->tick_count);
#define Q1_LAUNCH_SS(SP) \
    hipLaunchKernelGGL((sample_step_kernel<SP>), g, bs, 0, h->stream, h->p, h->st, logits, row_stride, seed, counter, counter_dev, \
                       deterministic, keys, mouse, logp, obs, reward, done, zero_start, ep_return, partials)
    if (is_spec(h->p)) Q1_LAUNCH_SS(true);
    else Q1_LAUNCH_SS(false);
#undef Q1_LAUNCH_SS
    HIP_TRY(hipGetLastError());
    h->tick_count += 1;
    return Q1ENV_OK;
}

static int launch_mlp(q1env* h, const float* obs, const q1pol::Net& na, const q1pol::Net& nb, int nets, unsigned rows = 0) {
    const unsigned n = rows ? rows : (unsigned)h->p.n;
    if (!h->mlp_attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)q1pol::mlp_forward_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1pol::LDS_TOTAL));
        HIP_TRY(hipFuncSetAttribute((const void*)q1pol::mlp_forward_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1pol::LDS_TOTAL));
        h->mlp_attr_set = true;
    }
    // One workgroup per CU (LDS holds one network's weights).  Up to one 32-env tile per SIMD of a network's share of the CUs:
    // one wave per SIMD; beyond that two waves per SIMD.  Q1ENV_MLP_THREADS overrides (measurement only).
    static const int forced = [] { const char* e = getenv("Q1ENV_MLP_THREADS"); return e ? atoi(e) : 0; }();
    const unsigned cus = (unsigned)(nets == 2 ? (h->num_cus > 1 ? h->num_cus / 2 : 1) : h->num_cus);   // CUs per network
    const int threads = forced == 256 || forced == 512 ? forced : (n <= cus * 4u * 32u ? 256 : 512);
    const unsigned per_block = 32u * (unsigned)(threads / 64);                   // envs one workgroup covers per grid-stride pass
    unsigned blocks = (n + per_block - 1u) / per_block;
    if (blocks > cus) blocks = cus;
    const dim3 g(blocks * (unsigned)nets), b(threads);
    if (threads == 256)
        hipLaunchKernelGGL(q1pol::mlp_forward_kernel<256>, g, b, q1pol::LDS_TOTAL, h->stream, (int)n, obs, na, nb, nets);
    else
        hipLaunchKernelGGL(q1pol::mlp_forward_kernel<512>, g, b, q1pol::LDS_TOTAL, h->stream, (int)n, obs, na, nb, nets);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_policy_forward(q1env_t* h, const float* obs, const float* w1, const float* b1, const uint16_t* w23_image, const float* b2,
                         const float* b3, int out_dim, float* out) {
    if (!h || !obs || !w1 || !b1 || !w23_image || !b2 || !b3 || !out) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_forward: null argument");
    if (out_dim < 1 || out_dim > 32) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_forward: out_dim must be in 1..32");
    DeviceGuard guard(h->device);
    const q1pol::Net net{w1, b1, w23_image, b2, b3, out, out_dim};
    return launch_mlp(h, obs, net, net, 1);
}

int q1env_policy_value_forward(q1env_t* h, const float* obs, const q1env_mlp* pi, const q1env_mlp* vf) {
    if (!h || !obs || !pi || !vf) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_value_forward: null argument");
    for (const q1env_mlp* m : {pi, vf}) {
        if (!m->w1 || !m->b1 || !m->w23_image || !m->b2 || !m->b3 || !m->out) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_value_forward: null pointer in q1env_mlp");
        if (m->out_dim < 1 || m->out_dim > 32) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_value_forward: out_dim must be in 1..32");
    }
    DeviceGuard guard(h->device);
    const q1pol::Net na{pi->w1, pi->b1, pi->w23_image, pi->b2, pi->b3, pi->out, pi->out_dim};
    const q1pol::Net nb{vf->w1, vf->b1, vf->w23_image, vf->b2, vf->b3, vf->out, vf->out_dim};
    return launch_mlp(h, obs, na, nb, 2);
}

int q1env_policy_forward_rows(q1env_t* h, uint64_t rows, const float* obs, const q1env_mlp* m) {
    if (!h || !obs || !m || !m->w1 || !m->b1 || !m->w23_image || !m->b2 || !m->b3 || !m->out)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_forward_rows: null argument");
    if (m->out_dim < 1 || m->out_dim > 32) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_forward_rows: out_dim must be in 1..32");
    if (rows == 0 || rows > 0x7FFFFFFFull / 32u) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_policy_forward_rows: rows out of range");
    DeviceGuard guard(h->device);
    const q1pol::Net net{m->w1, m->b1, m->w23_image, m->b2, m->b3, m->out, m->out_dim};
    return launch_mlp(h, obs, net, net, 1, (unsigned)rows);
}

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

// ---- resident sampler ----------------------------------------------------------------------------------------------------
int q1env_sample_resident(q1env_t* h, const q1env_resident_args* a) {
    if (!h || !a || !a->pi) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: null argument");
    const q1env_mlp* m = a->pi;
    if (!m->w1 || !m->b1 || !m->w23_image || !m->b2 || !m->b3) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: null pointer in q1env_mlp");
    if (!a->keys_dev || !a->logp_dev || !a->obs_dev || !a->reward_dev || !a->done_dev || !a->ep_return_dev || !a->partials_dev || !a->status_dev)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: null trajectory / status pointer");
    if (a->ticks <= 0 || !(a->timeout_s > 0.0) || a->timeout_s > 30.0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: bad ticks / timeout_s");
    const int width = policy_row_width(h->p);
    if (m->out_dim != width) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: pi->out_dim must be " + std::to_string(width));
    if (width > 24) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: more than 24 policy outputs do not fit the resident workgroup's LDS (use q1env_sample_step)");
    if (h->p.yaw_mode == 2 && !m->out) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: a discrete-mouse policy needs the logits trajectory (pi->out)");
    if (h->p.yaw_mode != 0 && !a->mouse_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: mouse trajectory required");
    DeviceGuard guard(h->device);
    if (!h->resident_attr_set) {
#define Q1_ATTR(SP, TP, R3) HIP_TRY(hipFuncSetAttribute((const void*)sampler_resident_kernel<SP, TP, R3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1res::Map<R3>::bytes(TP)))
        Q1_ATTR(true, 1, 16); Q1_ATTR(true, 2, 16); Q1_ATTR(true, 1, 24);
        Q1_ATTR(false, 1, 16); Q1_ATTR(false, 2, 16); Q1_ATTR(false, 1, 24);
#undef Q1_ATTR
        h->resident_attr_set = true;
    }
    // one workgroup per CU (its LDS holds the network): 128 envs per workgroup at one tile per policy wave, 256 at two (heads of up to 10
    // outputs only: the 24-row W3 tile of a discrete-mouse head leaves no room for the second tile's hand-off area)
    const unsigned n = (unsigned)h->p.n, cus = (unsigned)h->num_cus;
    const bool wide = h->p.yaw_mode == 2 || width > 10;   // a discrete-mouse head: the 24-row variant (gathers all of an env's logits, writes the row before it samples)
    int tp = (n + 127u) / 128u <= cus ? 1 : (!wide && (n + 255u) / 256u <= cus ? 2 : 0);
    if (const char* f = getenv("Q1ENV_RESIDENT_TP")) { if (f[0] == '2' && tp == 1 && !wide) tp = 2; }      // measurement knob
    if (!tp)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: too many envs for one resident grid (" + std::to_string(cus * (wide ? 128u : 256u)) +
                                           " at most on this device)");
    ResidentArgs k{};
    k.ticks = a->ticks;
    k.pi = q1pol::Net{m->w1, m->b1, m->w23_image, m->b2, m->b3, m->out, m->out_dim};
    k.seed = a->seed; k.counter_offset = a->counter_offset + (a->counter_dev ? 0 : h->tick_count); k.counter_dev = a->counter_dev;
    k.deterministic = a->deterministic;
    k.keys = a->keys_dev; k.mouse = a->mouse_dev; k.logp = a->logp_dev; k.obs = a->obs_dev; k.reward = a->reward_dev; k.done = a->done_dev;
    k.zero_start = a->zero_start_dev; k.ep_return = a->ep_return_dev; k.partials = a->partials_dev;
    k.status = a->status_dev;
    k.timeout_ticks = (uint64_t)(a->timeout_s * 1.0e8);
    const unsigned per_block = 128u * (unsigned)tp;
    const dim3 g((n + per_block - 1u) / per_block), b(512);
#define Q1_LAUNCH_RS(SP, TP, R3) hipLaunchKernelGGL((sampler_resident_kernel<SP, TP, R3>), g, b, q1res::Map<R3>::bytes(TP), h->stream, h->p, h->st, k)
    if (is_spec(h->p)) { if (wide) Q1_LAUNCH_RS(true, 1, 24); else if (tp == 1) Q1_LAUNCH_RS(true, 1, 16); else Q1_LAUNCH_RS(true, 2, 16); }
    else { if (wide) Q1_LAUNCH_RS(false, 1, 24); else if (tp == 1) Q1_LAUNCH_RS(false, 1, 16); else Q1_LAUNCH_RS(false, 2, 16); }
#undef Q1_LAUNCH_RS
    HIP_TRY(hipGetLastError());
    h->tick_count += (uint64_t)a->ticks;
    return Q1ENV_OK;
}

int q1env_selftest_division(int device, uint64_t n, uint64_t seed, double c0, double c1, uint64_t* mismatches4) {
    if (!mismatches4 || n == 0 || !(c0 > 0) || !(c1 > 0)) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_selftest_division: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(Q1ENV_ERR_NO_DEVICE, "q1env_selftest_division: no HIP device visible");
    if (device < 0 || device >= ndev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_selftest_division: bad device index");
    DeviceGuard guard(device);
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d, 0, 4 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(selftest_division_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, seed, c0, c1, d);
    unsigned long long hcounts[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpy(hcounts, d, sizeof(hcounts), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(Q1ENV_ERR_HIP, std::string("selftest: ") + hipGetErrorString(e));
    for (int k = 0; k < 4; ++k) mismatches4[k] = hcounts[k];
    return Q1ENV_OK;
}

int q1env_calibrate_traffic(q1env_t* h, int launches) {
    if (!h || launches <= 0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_calibrate_traffic: bad argument");
    DeviceGuard guard(h->device);
    if (int r = ensure_stage(h, arena_bytes((size_t)h->p.n))) return r;
    StatePtrs dst{};
    carve_into(h->stage, (size_t)h->p.n, dst);
    const int blk = block_for(h->p.n);
    for (int l = 0; l < launches; ++l)
        hipLaunchKernelGGL(calib_copy_kernel, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, dst);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

int q1env_timer_start(q1env_t* h) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_timer_start: null handle");
    DeviceGuard guard(h->device);
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return Q1ENV_OK;
}

int q1env_timer_mark(q1env_t* h) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_timer_mark: null handle");
    DeviceGuard guard(h->device);
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    return Q1ENV_OK;
}

int q1env_timer_elapsed(q1env_t* h, float* ms) {
    if (!h || !ms) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_timer_elapsed: null argument");
    DeviceGuard guard(h->device);
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return Q1ENV_OK;
}

int q1env_timer_stop(q1env_t* h, float* ms) {
    if (!h || !ms) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_timer_stop: null argument");
    DeviceGuard guard(h->device);
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return Q1ENV_OK;
}

}  // extern "C"
