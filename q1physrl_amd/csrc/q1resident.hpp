// q1resident.hpp - the resident sampler of libq1env (q1env_sample_resident, include/q1env.h): a whole sampling horizon as ONE
// dispatch.  Included by q1env_resident.hip after q1policy.hpp (matrix-core forward) and q1policy_glue.hpp (action sampling).
//
// The two-launch sampler tick (q1env_policy_value_forward + q1env_sample_step) costs ~25 us at 32 768 envs - two dispatch
// boundaries, 157 KB of weights staged into every CU's LDS again, a few us of arithmetic.  Here the policy network's weights are
// staged ONCE per workgroup, the env state lives in registers for the whole horizon, and a workgroup is self-contained:
//
//   waves 0..3                POLICY waves, ONE per SIMD (a tile of 32 envs through the three layers is latency-bound on one wave;
//                             a second policy wave on the SIMD would double every tick's latency).  Per tick and tile: wait for
//                             the tile's observations (tick 0: the trajectory's row 0), forward (q1pol::mlp_tile), gather the
//                             logits of an env into one lane, sample the action + log-probability (sample_action_regs: the same
//                             Philox draws as q1env_sample_step), write logits / action / log-probability into the trajectory,
//                             hand the action over
//   waves 4..4 + 2 TP - 1     ENV waves, 64 envs each (two tiles).  Per tick: wait for the two tiles' actions, tick (+ in-kernel reset
//                             of finished episodes, Philox exactly as q1env_sample_step), hand the observations over, write
//                             reward / done / next observation into the tick-major trajectory, episode statistics in registers.
//                             Their float64 arithmetic runs while the policy wave of their SIMD waits.
//   remaining waves           exit at once (the workgroup has 512 threads so that it owns its CU's register file and LDS)
// TP = tiles per policy wave per tick (1 or 2): a workgroup serves 128 TP envs, the grid is ceil(N / (128 TP)) workgroups, one per CU.
// Both sides of every hand-off are waves of ONE workgroup, so the hand-offs go through LDS: the data (observation rows / packed
// actions), a workgroup-scope release, then ONE tag word per env wave (observations) or per tile (actions) that carries the tick
// number; the reader spins on the tag (a broadcast ds_read, ~0.1 us per look - an L2 granule costs ~0.45 us per look), acquires, reads.
// LDS room: only 16 of the 32 rows of the W3 tile are kept (rows >= out_dim are zero anyway; lanes 16..31 re-read rows 0..15
// and produce output rows nobody looks at), which frees 8.4 KB next to the weights; a discrete-mouse head with up to 24 outputs
// keeps 24 rows and runs at one tile per policy wave.
// The value network is NOT in the loop: its forward over the (T + 1) N stored observations runs afterwards as one batched launch
// at full efficiency (q1env_policy_forward_rows) - nothing in the loop depends on it.
// Bit-identical trajectories to the two-launch sampler (same forward arithmetic per tile, same draws, same env arithmetic).
// Every wait is bounded (waves of one workgroup are co-resident by construction; the bound guards against a wave that died);
// status words as the tick server's: [0..2] env side, [3..4] policy side, written on failure only.
#pragma once
#include "q1policy.hpp"

struct ResidentArgs {
    int ticks;
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
    uint32_t* status;
    uint64_t timeout_ticks;
};

namespace q1res {
// LDS map of a workgroup: the network with W3 trimmed to R3 rows (16 for the continuous head's out_dim <= 10, 24 for a discrete-mouse
// head of up to 24 outputs) and the hand-off area for 128 TP envs.  (R3, TP) = (16, 1), (16, 2), (24, 1) fit the CU's 160 KB; (24, 2) does not.
constexpr size_t L_W2 = 0;
constexpr size_t L_W3 = q1pol::LDS_W2;                                                 // 135168
template <int R3> struct Map {
    static constexpr size_t W3_BYTES = (size_t)R3 * q1pol::ROW_BYTES;
    static constexpr size_t L_B2 = L_W3 + W3_BYTES;
    static constexpr size_t L_W1 = L_B2 + q1pol::LDS_B2;
    static constexpr size_t L_TAGS = L_W1 + q1pol::LDS_W1;                             // uint32 obs_tag[4], act_tag[8] (+ pad)
    static constexpr size_t L_ACT = L_TAGS + 64;                                       // uint64 act[128 TP]
    static constexpr uint32_t IMG_VEC16 = (uint32_t)((q1pol::LDS_W2 + W3_BYTES) / 16); // W2 rows + the first R3 W3 rows of the host's image
    static constexpr size_t l_obs(int tp) { return L_ACT + (size_t)128 * tp * 8; }     // float obs[128 TP][6]
    static constexpr size_t bytes(int tp) { return l_obs(tp) + (size_t)128 * tp * 24; }
};
static_assert(Map<16>::bytes(2) <= 163840 && Map<24>::bytes(1) <= 163840, "the resident sampler's LDS map must fit one CU");

__device__ __forceinline__ uint32_t tag_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void tag_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// spin (bounded) until *tag == want; wave-uniform (every lane reads the same word)
__device__ __forceinline__ bool wait_tag(const uint32_t* tag, uint32_t want, uint64_t timeout_ticks) {
    uint32_t polls = 0;
    uint64_t t_wait = 0;
    while (tag_load(tag) != want) {
        if ((++polls & 1023u) == 0u) {                           // (the 100 MHz clock is only looked at every 1024 failed looks)
            const uint64_t now = wall_clock64();
            if (t_wait == 0) t_wait = now;
            else if (now - t_wait > timeout_ticks) return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}
}  // namespace q1res

// u = env wave of the workgroup (0 .. 2 TP - 1): envs i = block_env0 + 64 u + lane
template <bool SPEC, int TP, int R3>
__device__ __forceinline__ void resident_env_wave(const Params& p, const StatePtrs& s, const ResidentArgs& a, uint32_t i, uint32_t u, unsigned char* lds) {
    using M = q1res::Map<R3>;
    const uint32_t lane = threadIdx.x & 63u, n = (uint32_t)p.n;
    const bool live = i < n;
    const uint64_t genv = (uint64_t)p.env_index_base + (uint64_t)i;
    const uint64_t counter0 = a.counter_offset + (a.counter_dev ? *a.counter_dev : 0ull);
    uint32_t* obs_tag = reinterpret_cast<uint32_t*>(lds + M::L_TAGS) + u;
    const uint32_t* act_tag = reinterpret_cast<const uint32_t*>(lds + M::L_TAGS) + 4u + 2u * u;       // this wave's two tiles
    const uint64_t* act = reinterpret_cast<const uint64_t*>(lds + M::L_ACT) + 64u * u + lane;
    float* obs_row = reinterpret_cast<float*>(lds + M::l_obs(TP)) + (size_t)(64u * u + lane) * 6u;
    Env env{};
    double ep_ret = 0.0;
    if (live) { load_env(s, n, i, env); ep_ret = a.ep_return[i]; }
    double slot[4] = {0.0, 0.0, 0.0, 0.0};                      // this wave's statistics slot (lane 0)
    if (lane == 0 && live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) slot[k] = a.partials[(size_t)(i >> 6) * 4 + k];
    }
    bool timed_out = false, last_zs = false;
    int completed = 0;
    const TickConsts tc = tick_consts();
    for (int t = 0; t < a.ticks; ++t) {
        if (!q1res::wait_tag(act_tag, (uint32_t)t + 1u, a.timeout_ticks) || !q1res::wait_tag(act_tag + 1, (uint32_t)t + 1u, a.timeout_ticks)) {
            timed_out = true;
            break;
        }
        const uint64_t g = *act;
        TickOut<float> o;
        o.reward = 0.0f; o.done = false;
#pragma unroll
        for (int k = 0; k < 6; ++k) o.obs[k] = 0.0f;
        bool zs = false;
        if (live) {
            const uint32_t keys = (uint32_t)(g >> 32) & ((1u << cfg_num_keys<SPEC>(p)) - 1u);
            const double yaw_act = cfg_yaw_mode<SPEC>(p) ? (double)__uint_as_float((uint32_t)g) : 0.0;
            tick<float, SPEC>(p, tc, env, keys, yaw_act, o);
            zs = (env.flags & FLAG_ZERO_START) != 0;                              // of the episode the step belonged to
            if (o.done) {
                reset_philox(p, env, a.seed, genv, counter0 + (uint64_t)t + 1);
                observe<float>(p, env, o.obs);
            }
            last_zs = zs;
        }
        // hand the observations over: rows, release, tag = number of ticks served
        {
            float2* w = reinterpret_cast<float2*>(obs_row);
            w[0] = make_float2(o.obs[0], o.obs[1]);
            w[1] = make_float2(o.obs[2], o.obs[3]);
            w[2] = make_float2(o.obs[4], o.obs[5]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) q1res::tag_store(obs_tag, (uint32_t)t + 1u);
        }
        // the trajectory (plain stores: read after the kernel) and the episode bookkeeping of q1env_episode_stats
        if (live) {
            write_obs<float>(a.obs + (size_t)(t + 1) * n * 6u, (size_t)i, o.obs);
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

// v = policy wave of the workgroup (0..3): tiles v TP + j, j < TP (tile = 32 envs; tile k of the workgroup belongs to env wave k / 2)
template <int TP, int R3>
__device__ __forceinline__ void resident_policy_wave(const Params& p, const ResidentArgs& a, uint32_t block_env0, uint32_t v, unsigned char* lds) {
    using M = q1res::Map<R3>;
    constexpr int NLG = R3 == 16 ? 10 : 24;                      // logits an env's lane gathers (rows 0..9, or all 24)
    const uint32_t lane = threadIdx.x & 63u, n = (uint32_t)p.n;
    const uint32_t col = lane & 31u, half = lane >> 5;
    const int W = a.pi.out_dim;
    const uint64_t counter0 = a.counter_offset + (a.counter_dev ? *a.counter_dev : 0ull);
    const unsigned char* w1row = lds + M::L_W1 + (size_t)col * 32u + half * 16u;
    const unsigned char* wrow = lds + q1res::L_W2 + (size_t)col * q1pol::ROW_BYTES + half * 16u;
    const unsigned char* w3row = lds + q1res::L_W3 + (size_t)(col % (uint32_t)R3) * q1pol::ROW_BYTES + half * 16u;   // rows R3..31 alias 0..
    const float* l_b2 = reinterpret_cast<const float*>(lds + M::L_B2);
    const uint32_t* obs_tags = reinterpret_cast<const uint32_t*>(lds + M::L_TAGS);
    uint32_t* act_tags = reinterpret_cast<uint32_t*>(lds + M::L_TAGS) + 4u;
    uint64_t* act = reinterpret_cast<uint64_t*>(lds + M::L_ACT);
    const float* obs_lds = reinterpret_cast<const float*>(lds + M::l_obs(TP));
    uint32_t tile[TP], loc[TP], env[TP];
    bool live[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        tile[j] = v * (uint32_t)TP + (uint32_t)j;
        loc[j] = tile[j] * 32u + col;                           // env index within the workgroup
        env[j] = block_env0 + loc[j];
        live[j] = env[j] < n;
    }
    float b3[NLG];                                                // the output bias: fetched once
#pragma unroll
    for (int k = 0; k < NLG; ++k) b3[k] = k < W ? a.pi.b3[k] : 0.0f;
    bool timed_out = false;
    int handed = 0;
    for (int t = 0; t < a.ticks && !timed_out; ++t) {
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            // the tick's Philox draws need no logits: drawn before the wait for the env wave, i.e. in time that is idle anyway
            uint32_t r[4] = {0u, 0u, 0u, 0u}, r2[4] = {0u, 0u, 0u, 0u};
            if (half == 0u && live[j]) sample_action_draws(a.seed, (uint64_t)p.env_index_base + (uint64_t)env[j], counter0 + (uint64_t)t, r, r2);
            // ---- the tile's observations: lane (col, half) needs columns half, 2 + half, 4 + half of env `col`
            float x[3];
            if (t == 0) {
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) x[sx] = live[j] ? a.obs[(size_t)env[j] * 6u + 2u * (uint32_t)sx + half] : 0.0f;
            } else {
                if (!q1res::wait_tag(obs_tags + (tile[j] >> 1), (uint32_t)t, a.timeout_ticks)) { timed_out = true; break; }
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) x[sx] = obs_lds[(size_t)loc[j] * 6u + 2u * (uint32_t)sx + half];
            }
            // ---- forward: Y^T of the tile, then all of an env's logits into its half-0 lane
            const q1pol::f16x8 xb = q1pol::split_inputs(x, half);
            const q1pol::f32x16 y = q1pol::mlp_tile(xb, w1row, wrow, w3row, l_b2, half, nullptr);
            // lane (col, half) holds rows r + 8 g + 4 half in y[4 g + r]: half 0 owns rows 0..3, 8..11, 16..19 of its column, rows 4..7,
            // 12..15, 20..23 come over from lane + 32
            float lg[NLG];
#pragma unroll
            for (int g = 0; g < (NLG == 10 ? 1 : 3); ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float other = __shfl_xor(y[4 * g + r], 32, 64);
                    lg[8 * g + r] = y[4 * g + r];
                    lg[8 * g + 4 + r] = other;
                }
            if (NLG == 10) { lg[8] = y[4]; lg[9] = y[5]; }
#pragma unroll
            for (int k = 0; k < NLG; ++k) lg[k] = k < W ? lg[k] + b3[k] : 0.0f;
            // sample in the half-0 lanes, hand the tile's actions over (data, release, tag = number of ticks handed over), and only
            // THEN write the trajectory: a release waits for every store issued before it, and these go all the way to HBM
            uint32_t keys = 0;
            float mouse = 0.0f, logp = 0.0f;
            const bool actor = half == 0u && live[j];
            float* row = a.pi.out ? a.pi.out + ((size_t)t * n + env[j]) * (uint32_t)W : nullptr;
            if (actor) {
                float lg10[10];
#pragma unroll
                for (int k = 0; k < 10; ++k) lg10[k] = lg[k];
                if (NLG > 10) {
                    // a discrete mouse: the categorical part of the sampling walks the row in memory (as q1env_sample_step does), so the
                    // logits row of the trajectory is written BEFORE the sampling here (a thread reads its own stores)
#pragma unroll
                    for (int k = 0; k < NLG; ++k)
                        if (k < W) row[k] = lg[k];
                }
                sample_action_from_draws(p, lg10, row, r, r2, a.deterministic, keys, mouse, logp);
            }
            if (half == 0u) act[loc[j]] = ((uint64_t)(keys & 0xFu) << 32) | (uint64_t)__float_as_uint(mouse);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) q1res::tag_store(act_tags + tile[j], (uint32_t)t + 1u);
            if (actor) {
                const uint32_t i = env[j];
                if (NLG == 10 && row) {
#pragma unroll
                    for (int k = 0; k < 10; ++k)
                        if (k < W) row[k] = lg[k];
                }
                a.keys[(size_t)t * n + i] = (uint8_t)keys;
                if (a.mouse) a.mouse[(size_t)t * n + i] = mouse;
                if (a.logp) a.logp[(size_t)t * n + i] = logp;
            }
        }
        if (!timed_out) handed = t + 1;
    }
    if (lane == 0 && handed != a.ticks) {
        if (timed_out) atomicOr(&a.status[3], 1u);
        atomicMax(&a.status[4], (uint32_t)(a.ticks - handed));
    }
}

template <bool SPEC, int TP, int R3>
__global__ void __launch_bounds__(512, 1)
sampler_resident_kernel(Params p, StatePtrs s, ResidentArgs a) {
    using M = q1res::Map<R3>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const uint32_t block_env0 = blockIdx.x * (128u * (uint32_t)TP);
    // stage the network: W2 and the first 16 rows of W3 (one contiguous piece of the host's image), b2 (pre-scaled), the layer-1 image
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.pi.w23);
        uint4* d = reinterpret_cast<uint4*>(lds);
        constexpr uint32_t PER = (M::IMG_VEC16 + 511u) / 512u;
        uint4 v[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t c = k * 512u + tid;
            v[k] = c < M::IMG_VEC16 ? src[c] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t c = k * 512u + tid;
            if (c < M::IMG_VEC16) d[c] = v[k];
        }
        if (tid < (uint32_t)q1pol::HID) {
            reinterpret_cast<float*>(lds + M::L_B2)[tid] = q1pol::TANH_PRESCALE * a.pi.b2[tid];
            q1pol::stage_w1_row(lds + M::L_W1, tid, a.pi.w1, a.pi.b1);
        }
        if (tid < 16u) reinterpret_cast<uint32_t*>(lds + M::L_TAGS)[tid] = 0u;
    }
    __syncthreads();                                                       // (the only barrier: before any wave leaves or loops)
    if (wave < 4u) {
        resident_policy_wave<TP, R3>(p, a, block_env0, wave, lds);
    } else if (wave < 4u + 2u * (uint32_t)TP) {
        const uint32_t u = wave - 4u;
        resident_env_wave<SPEC, TP, R3>(p, s, a, block_env0 + u * 64u + (tid & 63u), u, lds);
    }
}
