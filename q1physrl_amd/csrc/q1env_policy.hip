// q1env_policy.hip - policy-side glue of libq1env.so: action sampling, the fused sampler tick, episode
// statistics, GAE, the PPO loss gradient, and the launchers of the matrix-core policy / value forward (q1policy.hpp).
#include "q1env_host.hpp"
#include "q1policy.hpp"
#include "q1policy_glue.hpp"
#include "q1ppo_loss.hpp"

using namespace q1;

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

extern "C" {

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
    hipLaunchKernelGGL(ppo_loss_grad_kernel<false>, grid_for((int)batch, 256), dim3(256), 0, h->stream, h->p, (int)batch, logits, old_logits,
                       row_stride, row_stride, keys, mouse, logp_old, adv, value, value_old, vtarg, (const int64_t*)nullptr, (const int64_t*)nullptr, clip_param,
                       vf_clip_param, vf_loss_coeff, entropy_coeff, kl_coeff_dev, 1.0f, 1.0f, dlogits, dvalue, partials);
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

}  // extern "C"
