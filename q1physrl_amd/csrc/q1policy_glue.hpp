// q1policy_glue.hpp - device-side policy glue shared by the sampler kernels of libq1env (q1env_policy.hip: policy_sample_kernel,
// sample_step_kernel; q1resident.hpp: the resident sampler): action sampling from a row of policy outputs, and the per-wave
// episode bookkeeping.  Device code only; included after q1env_device.hpp.
#pragma once
#include "q1env_device.hpp"

using namespace q1;

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
