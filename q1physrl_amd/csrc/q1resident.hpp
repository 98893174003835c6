// q1resident.hpp - the resident sampler of libq1env (q1env_sample_resident, include/q1env.h): a whole sampling horizon as ONE
// dispatch.  Included by q1env.hip after q1server.hpp (hand-off protocol) and q1policy.hpp (matrix-core forward).
//
// The two-launch sampler tick (q1env_policy_value_forward + q1env_sample_step) costs ~26 us at 32 768 envs - two dispatch
// boundaries, 157 KB of weights staged into every CU's LDS again, ~4 us of arithmetic.  Here the policy network's weights are
// staged ONCE, the env state lives in registers for the whole horizon, and the two halves talk through the tick server's tagged
// granules (q1server.hpp): XCD-local when a pair of waves found each other on one XCD, agent-scope otherwise.
//
//   blocks [0, Be)            ENV blocks: 8 waves x 64 envs.  Per tick: wait for the action granule, tick (+ in-kernel reset of
//                             finished episodes, Philox exactly as q1env_sample_step), publish the observation as result granules,
//                             write reward / done / next observation into the tick-major trajectory, episode statistics in registers
//   blocks [Be, Be + Bp)      POLICY blocks: 8 waves, the policy network in LDS.  Per tick and 32-env tile: wait for the tile's
//                             observation granules (tick 0: the trajectory's row 0), forward (q1pol::mlp_tile), gather the logits
//                             of an env into one lane, sample the action + log-probability (sample_action_regs: same Philox
//                             draws as q1env_sample_step), write logits / action / log-probability into the trajectory, publish
//                             the action granule
// TP = tiles per policy wave per tick: 1 (two policy blocks per env block: the policy's forward of a tick is one tile deep) or
// 2 (one policy block per env block).  The value network is NOT in the loop: its forward over the (T + 1) N stored observations
// runs afterwards as one batched launch at full efficiency (q1env_policy_forward_rows) - nothing in the loop depends on it.
// Bit-identical trajectories to the two-launch sampler (same forward arithmetic per tile, same draws, same env arithmetic).
// Every wait is bounded exactly like the tick server's; status words as there ([0..2] env side, [3..4] policy side).
#pragma once
#include "q1policy.hpp"
#include "q1server.hpp"

struct ResidentArgs {
    int ticks;
    uint32_t tag0;
    uint32_t env_blocks;            // Be (a multiple of 8 when the grid was padded for XCD affinity)
    q1pol::Net pi;                  // policy network; pi.out = logits trajectory float[T][N][width] (may be null), pi.out_dim = width
    uint64_t seed;
    uint64_t counter_offset;
    const uint64_t* counter_dev;    // optional: Philox counter base in device memory (added to counter_offset)
    int deterministic;
    uint8_t* keys;                  // [T][N]
    float* mouse;                   // [T][N] (null without a mouse)
    float* logp;                    // [T][N]
    float* obs;                     // [T + 1][N][6]: row 0 is the input, rows 1..T are written
    float* reward;                  // [T][N]
    uint8_t* done;                  // [T][N]
    uint8_t* zero_start;            // [N]: flag of the episode the LAST step belonged to
    double* ep_return;              // [N]
    double* partials;               // [ceil(N/64)][4]
    uint64_t* mailbox;              // agent-scope copies: uint64[N], uint64[4][N][2]
    uint64_t* results;
    NearBufs near;                  // XCD-local copies (null = agent-scope only)
    uint32_t* status;
    uint64_t timeout_ticks;
};

// place bits ride in bits 36..39 of EVERY result granule here (the policy reads the observation pairs only)
__device__ __forceinline__ uint64_t result_granule(uint64_t tag, uint32_t place, uint32_t low_bits, uint32_t flags) {
    return (tag << 40) | ((uint64_t)place << 36) | ((uint64_t)flags << 32) | (uint64_t)low_bits;
}

template <bool SPEC>
__device__ __forceinline__ void resident_env_wave(const Params& p, const StatePtrs& s, const ResidentArgs& a, uint32_t i, float* slab) {
    const uint32_t lane = threadIdx.x & 63u, n = (uint32_t)p.n;
    const bool live = i < n;
    const uint32_t wave_first = i - lane;
    const bool full = wave_first + 64u <= n;
    const uint32_t my_xcc = xcc_id();
    const bool has_near = a.near.mailbox != nullptr;
    const uint64_t genv = (uint64_t)p.env_index_base + (uint64_t)i;
    const uint64_t counter0 = a.counter_offset + (a.counter_dev ? *a.counter_dev : 0ull);
    const Backoff bo{0, 0, 0, 0};
    Env env{};
    double ep_ret = 0.0;
    if (live) { load_env(s, n, i, env); ep_ret = a.ep_return[i]; }
    double slot[4] = {0.0, 0.0, 0.0, 0.0};                      // this wave's statistics slot (lane 0)
    if (lane == 0 && live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) slot[k] = a.partials[(size_t)(i >> 6) * 4 + k];
    }
    bool near_peer = false, peer_known = false, timed_out = false, last_zs = false;
    int completed = 0;
    for (int t = 0; t < a.ticks; ++t) {
        const uint64_t tag = tick_tag(a.tag0, (uint32_t)t);
        uint64_t g = 0;
        // until the policy wave's place is known (tick 0) both copies are polled: near-first, the agent-scope copy every eighth poll
        const bool near_first = has_near && (!peer_known || near_peer);
        if (!wait_for(live, a.timeout_ticks, bo, [&](uint32_t polls) {
                const bool far = !near_first || (polls % NEAR_POLL_PERIOD) == NEAR_POLL_PERIOD - 1u;
                g = granule_load((far ? a.mailbox : a.near.mailbox) + i);
                return (g >> 40) == tag;
            })) {
            timed_out = true;
            break;
        }
        near_peer = has_near && __all(!live || peer_is_near((uint32_t)(g >> 36) & 0xFu, my_xcc));
        peer_known = true;
        TickOut<float> o;
        o.reward = 0.0f; o.done = false;
        bool zs = false;
        if (live) {
            const uint32_t keys = (uint32_t)(g >> 32) & ((1u << cfg_num_keys<SPEC>(p)) - 1u);
            const double yaw_act = cfg_yaw_mode<SPEC>(p) ? (double)__uint_as_float((uint32_t)g) : 0.0;
            tick<float, SPEC>(p, env, keys, yaw_act, o);
            zs = (env.flags & FLAG_ZERO_START) != 0;                              // of the episode the step belonged to
            if (o.done) {
                reset_philox(p, env, a.seed, genv, counter0 + (uint64_t)t + 1);
                observe<float>(p, env, o.obs);
            }
            if (t + 1 < a.ticks) {                                               // (nobody reads the last tick's granules)
                const uint32_t place = PEER_VALID | my_xcc;
                uint64_t* r = near_peer ? a.near.results : a.results;
#pragma unroll
                for (uint32_t q = 0; q < 3u; ++q) {
                    const uint64_t g0 = result_granule(tag, place, __float_as_uint(o.obs[2 * q]), 0u);
                    const uint64_t g1 = result_granule(tag, place, __float_as_uint(o.obs[2 * q + 1]), 0u);
                    if (near_peer) granule_pair_store_near(pair_ptr(r, n, q, i), g0, g1);
                    else granule_pair_store(pair_ptr(r, n, q, i), g0, g1);
                }
            }
            last_zs = zs;
        }
        // the trajectory (plain stores: read after the kernel) and the episode bookkeeping of q1env_episode_stats
        if (live) {
            float* obs_next = a.obs + (size_t)(t + 1) * n * 6u;
            if (full) write_obs_wave_f32(obs_next, wave_first, lane, o.obs, slab);
            else write_obs<float>(obs_next, (size_t)i, o.obs);
            a.reward[(size_t)t * n + i] = o.reward;
            a.done[(size_t)t * n + i] = o.done ? 1 : 0;
        }
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        if (live) {
            const double ret = ep_ret + (double)o.reward;
            const bool zfin = o.done && zs;
            ep_ret = o.done ? 0.0 : ret;
            v[0] = o.done ? 1.0 : 0.0; v[1] = zfin ? 1.0 : 0.0; v[2] = o.done ? ret : 0.0; v[3] = zfin ? ret : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) slot[k] += v[k];
        }
        completed = t + 1;
    }
    if (live) {
        store_env(s, n, i, env);
        a.ep_return[i] = ep_ret;
        if (a.zero_start && completed > 0) a.zero_start[i] = last_zs ? 1 : 0;
    }
    if (lane == 0 && live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a.partials[(size_t)(i >> 6) * 4 + k] = slot[k];
    }
    if (lane == 0 && completed != a.ticks) {
        atomicAdd(&a.status[0], 1u);
        if (timed_out) atomicOr(&a.status[1], 1u);
        atomicMax(&a.status[2], (uint32_t)(a.ticks - completed));
    }
}

// one policy wave: TP tiles of 32 envs (tile j at env index env0 + 32 j)
template <int TP>
__device__ __forceinline__ void resident_policy_wave(const Params& p, const ResidentArgs& a, uint32_t env0, const q1pol::LdsNet& l) {
    const uint32_t lane = threadIdx.x & 63u, n = (uint32_t)p.n;
    const uint32_t col = lane & 31u, half = lane >> 5;
    const uint32_t my_xcc = xcc_id();
    const bool has_near = a.near.mailbox != nullptr;
    const int W = a.pi.out_dim;
    const uint64_t counter0 = a.counter_offset + (a.counter_dev ? *a.counter_dev : 0ull);
    const uint32_t result_bytes = n * 64u;
    const __amdgpu_buffer_rsrc_t far_rsrc = granule_rsrc(a.results, result_bytes);
    const __amdgpu_buffer_rsrc_t near_rsrc = granule_rsrc(has_near ? a.near.results : a.results, result_bytes);
    const unsigned char* w1row = l.w1 + (size_t)col * 32u + half * 16u;
    const unsigned char* wrow = l.w2 + (size_t)col * q1pol::ROW_BYTES + half * 16u;
    const unsigned char* w3row = l.w3 + (size_t)col * q1pol::ROW_BYTES + half * 16u;
    uint32_t env[TP];
    bool live[TP], near_peer[TP], peer_known[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        env[j] = env0 + 32u * (uint32_t)j + col;
        live[j] = env[j] < n;
        near_peer[j] = false; peer_known[j] = false;
    }
    bool timed_out = false;
    int handed = 0;
    for (int t = 0; t < a.ticks && !timed_out; ++t) {
        const uint64_t tag = tick_tag(a.tag0, (uint32_t)t);
        const uint64_t want = tick_tag(a.tag0, t > 0 ? (uint32_t)t - 1u : 0u);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            // ---- the tile's observations: lane (col, half) needs columns half, 2 + half, 4 + half of env `col`
            float x[3] = {0.0f, 0.0f, 0.0f};
            if (t == 0) {
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) x[sx] = live[j] ? a.obs[(size_t)env[j] * 6u + 2u * (uint32_t)sx + half] : 0.0f;
            } else {
                u32x4v v[3];
                uint32_t polls = 0;
                uint64_t t_wait = 0;
                const bool near_first = has_near && (!peer_known[j] || near_peer[j]);
                for (;;) {
                    const bool far = !near_first || (polls % NEAR_POLL_PERIOD) == NEAR_POLL_PERIOD - 1u;
                    const __amdgpu_buffer_rsrc_t r = far ? far_rsrc : near_rsrc;
                    bool ok = true;
                    if (live[j]) {
#pragma unroll
                        for (uint32_t q = 0; q < 3u; ++q) v[q] = granule_pair_load_sc1(r, (q * n + env[j]) * 16u);
#pragma unroll
                        for (uint32_t q = 0; q < 3u; ++q) ok = ok && ((uint64_t)(v[q][1] >> 8) == want) && ((uint64_t)(v[q][3] >> 8) == want);
                    }
                    if (__all(ok)) break;
                    if ((++polls & 255u) == 0u) {
                        const uint64_t now = wall_clock64();
                        if (t_wait == 0) t_wait = now;
                        else if (now - t_wait > a.timeout_ticks) { timed_out = true; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
                if (timed_out) break;
                near_peer[j] = has_near && __all(!live[j] || peer_is_near((v[0][1] >> 4) & 0xFu, my_xcc));       // bits 36..39 of granule 0
                peer_known[j] = true;
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) x[sx] = live[j] ? __uint_as_float(v[sx][2u * half]) : 0.0f;          // low word of granule 2 sx + half
            }
            // ---- forward: Y^T of the tile, then all of an env's logits into its half-0 lane
            const q1pol::f16x8 xb = q1pol::split_inputs(x, half);
            const q1pol::f32x16 y = q1pol::mlp_tile(xb, w1row, wrow, w3row, l.b2, half, nullptr);
            // lane (col, half) holds rows r + 8 g + 4 half; rows 0..3, 8..9 are half 0's, rows 4..7 come over from lane + 32
            float lg[10];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float other = __shfl_xor(y[r], 32, 64);               // half 0 receives rows 4 + r of its column
                lg[r] = y[r];
                lg[4 + r] = other;
            }
            lg[8] = y[4]; lg[9] = y[5];
#pragma unroll
            for (int k = 0; k < 10; ++k) lg[k] = k < W ? lg[k] + a.pi.b3[k] : 0.0f;
            const bool actor = live[j] && half == 0u;
            if (actor) {
                const uint32_t i = env[j];
                if (a.pi.out) {
                    float* row = a.pi.out + ((size_t)t * n + i) * (uint32_t)W;
#pragma unroll
                    for (int k = 0; k < 10; ++k)
                        if (k < W) row[k] = lg[k];
                }
                uint32_t keys;
                float mouse, logp;
                sample_action_regs(p, lg, nullptr, a.seed, (uint64_t)p.env_index_base + (uint64_t)i, counter0 + (uint64_t)t, a.deterministic,
                                   keys, mouse, logp);
                a.keys[(size_t)t * n + i] = (uint8_t)keys;
                if (a.mouse) a.mouse[(size_t)t * n + i] = mouse;
                if (a.logp) a.logp[(size_t)t * n + i] = logp;
                const uint64_t act = (tag << 40) | ((uint64_t)(has_near ? (PEER_VALID | my_xcc) : 0u) << 36) | ((uint64_t)(keys & 0xFu) << 32) |
                                     (uint64_t)__float_as_uint(mouse);
                // (near_peer is wave-uniform: both halves of the tile waited on the same env wave)
                if (has_near && (!peer_known[j] || near_peer[j])) granule_store_near(a.near.mailbox + i, act);
                if (!peer_known[j] || !near_peer[j]) granule_store(a.mailbox + i, act);
            }
        }
        if (!timed_out) handed = t + 1;
    }
    if (lane == 0 && handed != a.ticks) {
        if (timed_out) atomicOr(&a.status[3], 1u);
        atomicMax(&a.status[4], (uint32_t)(a.ticks - handed));
    }
}

template <bool SPEC, int TP>
__global__ void __launch_bounds__(512, 1)
sampler_resident_kernel(Params p, StatePtrs s, ResidentArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    if (blockIdx.x < a.env_blocks) {
        // env block: 8 waves x 64 envs; the (unused) weight area of LDS lends each wave its observation slab
        float* slab = reinterpret_cast<float*>(lds) + wave * 384u;
        resident_env_wave<SPEC>(p, s, a, (blockIdx.x * 8u + wave) * 64u + (tid & 63u), slab);
        return;
    }
    const uint32_t q = blockIdx.x - a.env_blocks;
    q1pol::stage_net<512>(lds, a.pi.w1, a.pi.b1, a.pi.w23, a.pi.b2, tid);
    __syncthreads();
    const q1pol::LdsNet l = q1pol::lds_net(lds);
    uint32_t env0;
    if (TP == 2) {
        env0 = q * 512u + wave * 64u;                                  // policy block q <-> env block q, wave <-> wave
    } else {
        // two policy blocks per env block.  Blocks go round-robin over the XCDs (observed; speed only), so the two policy blocks of
        // env block b = 8 k + x are q = 16 k + x and 16 k + 8 + x: all three on XCD x when Be is a multiple of 8
        const uint32_t b = 8u * (q >> 4) + (q & 7u), sh = (q >> 3) & 1u;
        env0 = b * 512u + (8u * sh + wave) * 32u;
    }
    resident_policy_wave<TP>(p, a, env0, l);
}
