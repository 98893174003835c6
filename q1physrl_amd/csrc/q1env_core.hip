// q1env_core.hip - the env hot path of libq1env.so: kernels, handle and core C ABI (gfx950 only; see include/q1env.h).
//
// Build (q1physrl_amd/build.py): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -c <each .hip>,
// then one link into libq1env.so.  (-ffp-contract=off is part of the numerics contract: the reference never fuses multiply-add.)
#include "q1env_host.hpp"

#include <mutex>
#include <new>
#include <time.h>

using namespace q1;

// =========================================================================================== kernels
// One tick of every env (reference VectorPhysEnv.vector_step, env.py:482-510), one lane per env.
// Loads: 85 B of SoA state + the action; stores: the state + obs/reward/done.  The per-tick constants sit in SGPRs.
// LDS is used for one thing only: transposing the wave's 64 float32 observation rows so they leave as 16-B-per-lane
// coalesced stores (write_obs_wave_f32).
//   SPEC: default Config structure baked in (straight-line tick);  FMT: action layout, or FMT_RUNTIME.
// Large batches (round 4, measured and dropped - profiles/r4_step_large.txt): a software-pipelined streaming form of this kernel (a fixed
// grid of one-wave workgroups walking 64-env tiles grid-stride, the NEXT tile's 85-B state + action requested before the current tile is
// computed; 128 VGPRs, 4 waves per SIMD) was bit-identical and no faster - 7.8 vs 7.2 us per tick at 262 144 envs, 35.4-35.6 vs 33.6-35.2
// at 1 M, 135.6-147.7 vs 132.8-143.0 at 4 M: the plain kernel's five resident waves per SIMD already keep the memory system's queues full.
#ifndef Q1_STEP_MINWAVES            // (measurement knobs: minimum waves per SIMD the register allocation must leave room for)
#define Q1_STEP_MINWAVES 1
#endif
template <typename OBS_T, bool SPEC, int FMT>
__global__ void __launch_bounds__(256, Q1_STEP_MINWAVES)
step_kernel(float* pvx, float* pvy, float* pvz, double* ppx, double* ppy, double* pz, double* pyaw, double* ptrem,   // preloaded into SGPRs
            Params p, StatePtrs s, int fmt, const void* act_a, const void* act_b,
            OBS_T* obs, float* reward, uint8_t* done, uint8_t* zero_start, Signal sg) {
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
    signal_done_strict(sg);                            // (host-direct launches only: sg.sig is NULL otherwise)
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
template <typename OBS_T, bool SPEC, int FMT, bool HAS_RESET, int OUT_MODE, bool RET, bool FULL, int DEPTH_>
__device__ __forceinline__ void rollout_loop(const Params& p, const double* move_tab, Env& e, uint32_t i, uint32_t n, int ticks, int fmt,
                                             const void* act_a, const void* act_b, uint64_t seed, uint64_t tick0,
                                             OBS_T* obs, float* reward, uint8_t* done, int auto_reset, double& ret,
                                             float* slab) {
    const uint64_t genv = (uint64_t)p.env_index_base + (uint64_t)i;
    const bool random = (FMT >= 0 ? FMT : fmt) == FMT_RANDOM;
    const uint32_t lane = threadIdx.x & 63u, wave_first = i - lane;
    // Software prefetch for the packed layout: the RAW bytes of tick t+1's action are requested at the top of the
    // body (unconditionally, so the loads stay in the body's first basic block), before tick t is computed, and are only decoded one
    // iteration later; other layouts fetch in place.  The two action pointers advance by one tick per iteration - by ZERO on the last
    // one, which re-reads its own action instead of running past the caller's arrays: one scalar select for both strides instead of a
    // 64-bit multiply-add per array.
    constexpr bool PREFETCH = (FMT == FMT_PACKED);
    constexpr int DEPTH = PREFETCH ? DEPTH_ : 1;
    // DEPTH 2: two ticks ahead, in TWO register sets used by alternate ticks (the loop is unrolled by two so that no value has to be
    // moved between them - a move of a prefetched value would itself have to wait for it): the wait for a prefetched action then only
    // covers memory operations at least two ticks old.  gfx9's vmcnt retires loads and stores in order, so with depth 1 the wait also
    // covers the PREVIOUS tick's output stores - and a write-through store's acknowledgement comes from the memory side, not from L2.
    uint32_t kraw_a = 0, kraw_b = 0;
    float mraw_a = 0.0f, mraw_b = 0.0f;
    const uint8_t* ka = (const uint8_t*)act_a + i;
    const float* ma = (const float*)act_b + i;
    if constexpr (PREFETCH) {
        kraw_a = *ka;
        mraw_a = *ma;
        if constexpr (DEPTH == 2) {
            const size_t stride = (1 < ticks) ? (size_t)n : (size_t)0;
            ka += stride;
            ma += stride;
            kraw_b = *ka;
            mraw_b = *ma;
        }
        // Drain every outstanding load (state + first action) once, here: the waitcnt scoreboard then enters the
        // loop clean, so inside the loop the wait for a prefetched action is vmcnt(#younger ops) as seen along the
        // back edge - it no longer has to cover the preheader's load order and does not drain the tick's stores.
        __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0), expcnt/lgkmcnt untouched
    }
    TickConsts tc = tick_consts();
    tc.move_tab = move_tab;
    tc.has_move_tab = true;
    auto one_tick = [&](const int t, uint32_t& kraw, float& mraw) __attribute__((always_inline)) {
        double yaw_act;
        uint32_t keys;
        if constexpr (PREFETCH) {
            // decode tick t's raw action FIRST, then request tick t+DEPTH's into the same two registers (the empty asm keeps the requests
            // behind the decode: no register copies, and the loop closes with ONE conditional branch)
            keys = kraw & 0xFu;
            yaw_act = (double)mraw;
            asm volatile("" : "+v"(keys), "+v"(yaw_act) : : "memory");
            const size_t stride = (t + DEPTH < ticks) ? (size_t)n : (size_t)0;
            ka += stride;
            ma += stride;
            kraw = *ka;
            mraw = *ma;
        } else if (random) {
            keys = random_action<SPEC>(p, seed, genv, tick0 + (uint64_t)t, &yaw_act);
        } else {
            keys = fetch_action<SPEC, FMT>(p, fmt, act_a, act_b, (size_t)t * n + i, &yaw_act);
        }
        TickOut<OBS_T> o;
        tick<OBS_T, SPEC>(p, tc, e, keys, yaw_act, o);
        const size_t base = (size_t)t * n;
        // The per-tick outputs are written once and read by a later kernel / the host.  Q1_ROLLOUT_OUT_STORES = 2 (product): system-scope
        // write-through stores (store_sys in q1env_device.hpp) - acknowledged by the memory side, so the launch's completion signal means
        // "readable by anyone"; the write-through form of the ragged tail's rows included.  1 = non-temporal stores (round 4: acknowledged
        // by the XCD's L2, written back by the end-of-kernel release), 0 = plain stores.  A/B on MI355X: profiles/r5_visibility.txt.
#ifndef Q1_ROLLOUT_OUT_STORES
#define Q1_ROLLOUT_OUT_STORES 2
#endif
        if (OUT_MODE == 1 || (OUT_MODE < 0 && obs)) {
            if constexpr (sizeof(OBS_T) == 4 && FULL) {
                if constexpr (Q1_ROLLOUT_OUT_STORES == 2) write_obs_wave_f32_sys(obs, base + wave_first, lane, o.obs, slab);
                else if constexpr (Q1_ROLLOUT_OUT_STORES == 1) write_obs_wave_f32_nt(obs, base + wave_first, lane, o.obs, slab);
                else write_obs_wave_f32(obs, base + wave_first, lane, o.obs, slab);
            } else if constexpr (Q1_ROLLOUT_OUT_STORES == 2) {
#pragma unroll
                for (int j = 0; j < 6; ++j) store_sys(obs + (base + i) * 6 + j, o.obs[j]);
            } else write_obs<OBS_T>(obs, base + i, o.obs);
        }
        if constexpr (Q1_ROLLOUT_OUT_STORES == 2) {
            if (OUT_MODE == 1 || (OUT_MODE < 0 && reward)) store_sys(reward + base + i, o.reward);
            if (OUT_MODE == 1 || (OUT_MODE < 0 && done)) store_sys(done + base + i, (uint8_t)(o.done ? 1 : 0));
        } else if constexpr (Q1_ROLLOUT_OUT_STORES == 1) {
            if (OUT_MODE == 1 || (OUT_MODE < 0 && reward)) __builtin_nontemporal_store(o.reward, reward + base + i);
            if (OUT_MODE == 1 || (OUT_MODE < 0 && done)) __builtin_nontemporal_store((uint8_t)(o.done ? 1 : 0), done + base + i);
        } else {
            if (OUT_MODE == 1 || (OUT_MODE < 0 && reward)) (reward + base)[i] = o.reward;
            if (OUT_MODE == 1 || (OUT_MODE < 0 && done)) (done + base)[i] = o.done ? 1 : 0;
        }
        if constexpr (RET) ret += (double)o.reward;
        if constexpr (HAS_RESET) {
            if (auto_reset && o.done) reset_philox(p, e, seed, genv, tick0 + (uint64_t)t + 1);
        }
    };
    if constexpr (DEPTH == 2) {
        int t = 0;
        for (; t + 1 < ticks; t += 2) {
            one_tick(t, kraw_a, mraw_a);
            one_tick(t + 1, kraw_b, mraw_b);
        }
        if (t < ticks) one_tick(t, kraw_a, mraw_a);
    } else {
        for (int t = 0; t < ticks; ++t) one_tick(t, kraw_a, mraw_a);
    }
}

// RET: the launch accumulates each env's reward in float64 for return_sum (a convert and an add per tick that a caller who reads the
// per-tick rewards anyway - the bench, the sampler - does not pay for).
// DEPTH: how many ticks ahead the packed action is requested (rollout_loop): 1 for short launches, 2 from ROLLOUT_DEPTH2_MIN_TICKS ticks.
template <typename OBS_T, bool SPEC, int FMT, bool HAS_RESET, int OUT_MODE, bool RET, int DEPTH = 1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
rollout_kernel(Params p, StatePtrs s, int ticks, int fmt, const void* act_a, const void* act_b,
               uint64_t seed, uint64_t tick0, OBS_T* obs, float* reward, uint8_t* done,
               int auto_reset, double* return_sum, Signal sg) {
    __shared__ float slab[4][384];
    __shared__ double move_tab[128];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = (uint32_t)p.n;
    signal_start(sg);
    const double* mt = fill_move_table(p, move_tab);       // (every thread, before the early exit: it ends in a workgroup barrier)
    if (i >= n) return;
    Env e;
    load_env(s, n, i, e);
    double ret = 0.0;
    float* my_slab = slab[threadIdx.x >> 6];
    if (i - (threadIdx.x & 63u) + 64u <= n)
        rollout_loop<OBS_T, SPEC, FMT, HAS_RESET, OUT_MODE, RET, true, DEPTH>(p, mt, e, i, n, ticks, fmt, act_a, act_b, seed, tick0, obs, reward,
                                                                        done, auto_reset, ret, my_slab);
    else
        rollout_loop<OBS_T, SPEC, FMT, HAS_RESET, OUT_MODE, RET, false, DEPTH>(p, mt, e, i, n, ticks, fmt, act_a, act_b, seed, tick0, obs, reward,
                                                                         done, auto_reset, ret, my_slab);
    // The final stores address the arrays from a laundered copy of the index: the twelve 64-bit addresses of the initial loads
    // would otherwise stay in registers across the whole tick loop (24 VGPRs on top of the state and the tick's constants).
    uint32_t i_st = i;
    asm volatile("" : "+v"(i_st));
#ifndef Q1_ROLLOUT_STATE_NT         // 2 (product) = system-scope write-through (store_sys), 1 = non-temporal (round 4), 0 = plain
#define Q1_ROLLOUT_STATE_NT 2
#endif
    if constexpr (Q1_ROLLOUT_STATE_NT == 2) store_env_sys(s, n, i_st, e);
    else if constexpr (Q1_ROLLOUT_STATE_NT == 1) store_env_nt(s, n, i_st, e);
    else store_env(s, n, i_st, e);
    if constexpr (RET) { if (return_sum) store_sys(return_sum + i_st, return_sum[i_st] + ret); }
    signal_done(sg, i_st >> 6);
}

// the end stamp + sequence number of the completion signal behind whatever the stream holds (q1env_signal_mark)
__global__ void signal_mark_kernel(Signal sg) {
    sg.waves = 1;
    signal_start(sg);
    signal_done(sg, 0u);
}

template <typename OBS_T>
__global__ void __launch_bounds__(256) observe_kernel(Params p, StatePtrs s, OBS_T* obs, Signal sg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)p.n) return;
    Env e;
    load_env(s, (uint32_t)p.n, i, e);
    OBS_T o[6];
    observe<OBS_T>(p, e, o);
    write_obs<OBS_T>(obs, (size_t)i, o);
    signal_done_strict(sg);                            // (host-direct launches only)
}

// Reset from host-supplied raw draws (NumPy-compatible RNG stays on the host, the arithmetic is here).
template <typename OBS_T>
__global__ void __launch_bounds__(256)
reset_draws_kernel(Params p, StatePtrs s, int count, const int32_t* idx, const uint8_t* zero_start,
                   const double* yaw, const double* tm, const double* speed, const double* angle, OBS_T* obs, Signal sg) {
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
    signal_done_strict(sg);
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
              const double* trem, double* yaw, int64_t* smove, int64_t* fmove, uint8_t* jump, Signal sg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)p.n) return;
    Env e;
    load_env(s, (uint32_t)p.n, i, e);
    double yaw_act;
    const uint32_t keys = fetch_action<false, FMT_RUNTIME>(p, fmt, act_a, act_b, (size_t)i, &yaw_act);
    const Cmd c = decode<false>(p, make_tick_consts<false>(), e, keys, yaw_act, z_vel[i], trem[i]);
    s.yaw[i] = e.yaw;
#pragma unroll
    for (int k = 0; k < 4; ++k) s.lk[(size_t)k * p.n + i] = e.lk[k];
    s.flags[i] = (uint8_t)e.flags;
    yaw[i] = e.yaw;
    smove[i] = (int64_t)c.smove;
    fmove[i] = (int64_t)c.fmove;
    jump[i] = c.jump ? 1 : 0;
    signal_done_strict(sg);                            // (host-direct launches only)
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

// *counter += by: the one extra node of a replayable run of auto-reset ticks (q1env_step_autoreset_many)
__global__ void counter_add_kernel(uint64_t* counter, uint64_t by) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *counter += by;
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
    c.fmove = fmove[i]; c.smove = smove[i]; c.jump = button2[i] != 0 ? 1u : 0u;
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
    physics_core<VT, false>(make_tick_consts<false>(), vx, vy, vz, z, flags, c, m00, m01, m10, m11, dt, 10.0 * dt, 800.0 * dt);
    out_z[i] = z;
    out_vel[3 * (size_t)i] = vx; out_vel[3 * (size_t)i + 1] = vy; out_vel[3 * (size_t)i + 2] = vz;
    out_og[i] = (flags & FLAG_ON_GROUND) ? 1 : 0;
    out_jr[i] = (flags & FLAG_JUMP_RELEASED) ? 1 : 0;
}
// =========================================================================================== host side
namespace { thread_local std::string g_err; }

int q1_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
const char* q1_last_error_cstr() { return g_err.c_str(); }

int ensure_stage(q1env* h, size_t bytes) {
    if (bytes <= h->stage_bytes) return 0;
    if (h->stage) (void)hipFree(h->stage);
    h->stage = nullptr;
    h->stage_bytes = 0;
    HIP_TRY(hipMalloc(&h->stage, bytes));
    h->stage_bytes = bytes;
    return 0;
}

int ensure_pin(q1env* h, size_t bytes) {
    if (bytes <= h->pin_bytes) return 0;
    if (h->pin) (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->pin_bytes = 0;
    HIP_TRY(hipHostMalloc(&h->pin, bytes, hipHostMallocDefault));
    h->pin_bytes = bytes;
    return 0;
}

// The completion signal's memory: three 64-bit words of host-coherent pinned memory (one cache line of its own) the kernels write
// over PCIe, and the device-resident ticket counter.
int ensure_signal(q1env* h) {
    if (h->sig_host) return 0;
    void* host = nullptr;
    HIP_TRY(hipHostMalloc(&host, 256, hipHostMallocCoherent | hipHostMallocMapped));
    memset(host, 0, 256);
    void* dev = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dev, host, 0);
    if (e == hipSuccess) e = hipMalloc((void**)&h->ticket_dev, 4 * (SIGNAL_LEAVES + 1) * SIGNAL_LEAF_STRIDE);
    if (e == hipSuccess) e = hipMemsetAsync(h->ticket_dev, 0, 4 * (SIGNAL_LEAVES + 1) * SIGNAL_LEAF_STRIDE, h->stream);
    if (e != hipSuccess) {
        (void)hipHostFree(host);
        if (h->ticket_dev) { (void)hipFree(h->ticket_dev); h->ticket_dev = nullptr; }
        return fail(Q1ENV_ERR_HIP, std::string("completion signal set-up: ") + hipGetErrorString(e));
    }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) == hipSuccess && khz > 0) h->wall_clock_hz = 1e3 * (double)khz;
    else h->wall_clock_hz = 1e8;                       // gfx9: s_memrealtime counts at 100 MHz
    h->sig_dev = (uint64_t*)dev;
    h->sig_host = (volatile uint64_t*)host;
    return 0;
}

// Host-direct block (q1env_host.hpp): grown on demand, mapped + coherent so that a kernel's loads / stores reach it over PCIe uncached.
int ensure_direct(q1env* h, size_t bytes) {
    if (bytes <= h->direct_bytes) return 0;
    // the first use sizes the block for the largest host-direct batch (a step of HOST_DIRECT_MAX_ENVS envs needs <= 128 B per env), so that
    // a reset_at (1 env) followed by a reset_many or a step never re-allocates - a re-allocation is a synchronisation + two runtime calls
    if (bytes < 4096 * 128 + 8192) bytes = 4096 * 128 + 8192;
    (void)hipStreamSynchronize(h->stream);            // a launch may still be reading / writing the old block
    if (h->direct_host) (void)hipHostFree(h->direct_host);
    h->direct_host = nullptr; h->direct_dev = nullptr; h->direct_bytes = 0;
    void* host = nullptr;
    const size_t want = align_up(bytes + bytes / 2, 4096);
    HIP_TRY(hipHostMalloc(&host, want, hipHostMallocCoherent | hipHostMallocMapped));
    void* dev = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&dev, host, 0);
    if (e != hipSuccess) { (void)hipHostFree(host); return fail(Q1ENV_ERR_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e)); }
    h->direct_host = (char*)host; h->direct_dev = (char*)dev; h->direct_bytes = want;
    return 0;
}

// Batches up to this many envs take the host-direct form of q1env_step_host / q1env_reset_draws_host: the kernel reads the caller's
// actions from, and writes obs / reward / done to, host-coherent pinned memory itself and says so with the completion signal - one
// launch and a poll (~ 10 us) instead of a copy command each way and a stream synchronisation (~ 30 us; round 3's reset_at 36 us, gym
// step 33 us).  This is the size RLlib actually runs the reference at (data/params.yml:28 num_envs 100).  Above it the DMA engine's
// rate wins over PCIe loads issued by waves.  Q1ENV_HOST_DIRECT=0 turns it off (A/B).
constexpr size_t HOST_DIRECT_MAX_ENVS = 4096;
static bool host_direct_enabled() {
    const char* e = getenv("Q1ENV_HOST_DIRECT");
    return !(e && e[0] == '0');
}
static Signal direct_signal(q1env* h, size_t items) {
    Signal sg{};
    sg.sig = h->sig_dev;
    sg.ticket = h->ticket_dev;
    sg.seq = ++h->sig_seq;
    sg.waves = (uint32_t)((items + 63) / 64);
    sg.flags = 3u;
    return sg;
}

// Poll the sequence word until the last requested signal has arrived.  No sleep, no yield: the caller asked for latency.
// The load that sees the number is an ACQUIRE: the host's reads of the results that follow it (host-direct block, or the stamps) cannot be
// taken before it.  On a timeout the launch may still be queued or running and will write the signal words / the host-direct block later:
// the stream is drained before the error is returned, so that the caller's next call cannot race with it.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
}
int signal_wait(q1env* h, double timeout_s) {
    const uint64_t want = h->sig_seq;
    const uint64_t* seq = const_cast<const uint64_t*>(h->sig_host + 2);
    if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) >= want) return Q1ENV_OK;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        for (int k = 0; k < 256; ++k) {
            if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) >= want) return Q1ENV_OK;
            cpu_relax();
        }
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s) {
            const hipError_t q = hipStreamQuery(h->stream);
            const std::string state = hipGetErrorString(q);
            (void)hipStreamSynchronize(h->stream);         // nothing of this launch is left in flight when the error is returned
            return fail(Q1ENV_ERR_HIP, "completion signal did not arrive within the timeout (stream state: " + state + "; stream drained)");
        }
    }
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

// (Round 4 measured a skew between consecutive arrays - with power-of-two batch sizes every array starts at the same offset modulo any
// power of two - of 256 B, 4 352 B and 66 304 B: no effect at 262 144 / 1 M / 4 M envs, profiles/r4_exp_skew.txt; the memory system
// hashes addresses onto its channels.)
void carve_into(void* arena, size_t n, StatePtrs& st) {
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

size_t arena_bytes(size_t n) {
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

const char* q1env_last_error(void) { return q1_last_error_cstr(); }

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
        // (a high-priority stream was A/B-tested for the 20-tick launch in round 4: no difference, profiles/r4_bench_driver_steps20.json)
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
    if (h->sig_host) (void)hipHostFree((void*)h->sig_host);
    if (h->direct_host) (void)hipHostFree(h->direct_host);
    if (h->ticket_dev) (void)hipFree(h->ticket_dev);
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

int q1env_tick_count(const q1env_t* h, uint64_t* out) {
    if (!h || !out) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_tick_count: null argument");
    *out = h->tick_count;
    return Q1ENV_OK;
}

static void launch_step(q1env* h, int fmt, const void* a, const void* b, int obs_format, void* obs,
                        float* reward, uint8_t* done, uint8_t* zs, const Signal& sg = Signal{}) {
    const int blk = block_for(h->p.n);
    const dim3 g = grid_for(h->p.n, blk), bs(blk);
    const bool spec = is_spec(h->p);
#define Q1_LAUNCH_STEP(OT, SP, FM) \
    hipLaunchKernelGGL((step_kernel<OT, SP, FM>), g, bs, 0, h->stream, h->st.vx, h->st.vy, h->st.vz, h->st.px, h->st.py, h->st.z, h->st.yaw, \
                       h->st.trem, h->p, h->st, fmt, a, b, (OT*)obs, reward, done, zs, sg)
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
    if (n <= HOST_DIRECT_MAX_ENVS && host_direct_enabled()) {
        // host-direct: one launch, results written by the kernel into host-coherent memory, completion by signal
        if (int r = ensure_direct(h, in_bytes + out_bytes)) return r;
        if (int r = ensure_signal(h)) return r;
        char* hp = h->direct_host;
        char* dp = h->direct_dev;
        memcpy(hp, a, na);
        if (nb) memcpy(hp + ba, b, nb);
        const Signal sg = direct_signal(h, n);
        launch_step(h, fmt, dp, dp + ba, obs_format, obs ? dp + in_bytes : nullptr, reward ? (float*)(dp + in_bytes + bo) : nullptr,
                    done ? (uint8_t*)(dp + in_bytes + bo + br) : nullptr, zs ? (uint8_t*)(dp + in_bytes + bo + br + bd) : nullptr, sg);
        HIP_TRY(hipGetLastError());
        h->tick_count += 1;
        if (int r = signal_wait(h, 30.0)) return r;
        const char* po = hp + in_bytes;
        if (obs) memcpy(obs, po, no);
        if (reward) memcpy(reward, po + bo, n * 4);
        if (done) memcpy(done, po + bo + br, n);
        if (zs) memcpy(zs, po + bo + br + bd, n);
        return Q1ENV_OK;
    }
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

constexpr int ROLLOUT_DEPTH2_MIN_TICKS = 32;
int q1env_rollout(q1env_t* h, int ticks, int fmt, const void* a, const void* b, uint64_t seed, int obs_format,
                  void* obs, float* reward, uint8_t* done, int auto_reset, double* return_sum) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_rollout: null handle");
    DeviceGuard guard(h->device);
    if (ticks <= 0) return fail(Q1ENV_ERR_INVALID_ARG, "ticks must be > 0");
    if (int r = check_act(h, fmt, a, b, true)) return r;
    if (obs_format != Q1ENV_OBS_F32 && obs_format != Q1ENV_OBS_F64) return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    // (Q1ENV_TIMER_START / _STOP in auto_reset: record the handle's timer events around the launch, inside this call - as q1env_step_many)
    const bool t_start = (auto_reset & Q1ENV_TIMER_START) != 0, t_stop = (auto_reset & Q1ENV_TIMER_STOP) != 0;
    // (ABI v4) the completion signal: stamps and sequence number written by the launch's own waves (include/q1env.h)
    const bool s_start = (auto_reset & Q1ENV_STAMP_START) != 0, s_wait = (auto_reset & Q1ENV_SIGNAL_WAIT) != 0;
    const bool s_done = (auto_reset & Q1ENV_SIGNAL) != 0 || s_wait;
    auto_reset &= 1;
    Signal sg{};
    if (s_start || s_done) {
        if (int r = ensure_signal(h)) return r;
        sg.sig = h->sig_dev;
        sg.ticket = h->ticket_dev;
        sg.waves = (uint32_t)((h->p.n + 63) / 64);
        sg.flags = (s_start ? 1u : 0u) | (s_done ? 2u : 0u);
        if (s_done) sg.seq = ++h->sig_seq;
    }
    if (t_start) HIP_TRY(hipEventRecord(h->ev0, h->stream));
    const int blk = block_for(h->p.n);
    const dim3 g = grid_for(h->p.n, blk), bs(blk);
    const bool spec = is_spec(h->p);
#define Q1_LAUNCH_ROLL_D(OT, SP, FM, HR, OM, RT, DP)                                                                         \
    hipLaunchKernelGGL((rollout_kernel<OT, SP, FM, HR, OM, RT, DP>), g, bs, 0, h->stream, h->p, h->st, ticks, fmt, a, b, seed, \
                       h->tick_count, (OT*)obs, reward, done, auto_reset, return_sum, sg)
#define Q1_LAUNCH_ROLL(OT, SP, FM, HR, OM, RT) Q1_LAUNCH_ROLL_D(OT, SP, FM, HR, OM, RT, 1)
    // Packed actions with every output written: from ROLLOUT_DEPTH2_MIN_TICKS ticks per launch the action is requested TWO ticks ahead
    // (rollout_loop DEPTH 2; + 10 % steady state with the write-through output stores, profiles/r5_visibility.txt); shorter launches
    // keep the one-tick form (smaller prologue, no gain to amortise).  Q1ENV_ROLLOUT_DEPTH=1|2 in the environment forces one (A/B).
    static const int depth_forced = [] { const char* e = getenv("Q1ENV_ROLLOUT_DEPTH"); return e ? atoi(e) : 0; }();
    const bool deep = depth_forced ? depth_forced == 2 : ticks >= ROLLOUT_DEPTH2_MIN_TICKS;
    const bool all_out = obs && reward && done, no_out = !obs && !reward && !done;
    if (obs_format == Q1ENV_OBS_F32 && spec && (all_out || no_out) && !return_sum &&
        (fmt == Q1ENV_ACT_PACKED || fmt == Q1ENV_ACT_RANDOM)) {
        const int which = (fmt == Q1ENV_ACT_RANDOM ? 4 : 0) + (auto_reset ? 2 : 0) + (all_out ? 1 : 0);
        switch (which) {
            case 0: Q1_LAUNCH_ROLL(float, true, FMT_PACKED, false, 0, false); break;
            case 1: if (deep) Q1_LAUNCH_ROLL_D(float, true, FMT_PACKED, false, 1, false, 2); else Q1_LAUNCH_ROLL(float, true, FMT_PACKED, false, 1, false); break;
            case 2: Q1_LAUNCH_ROLL(float, true, FMT_PACKED, true, 0, false); break;
            case 3: if (deep) Q1_LAUNCH_ROLL_D(float, true, FMT_PACKED, true, 1, false, 2); else Q1_LAUNCH_ROLL(float, true, FMT_PACKED, true, 1, false); break;
            case 4: Q1_LAUNCH_ROLL(float, true, FMT_RANDOM, false, 0, false); break;
            case 5: Q1_LAUNCH_ROLL(float, true, FMT_RANDOM, false, 1, false); break;
            case 6: Q1_LAUNCH_ROLL(float, true, FMT_RANDOM, true, 0, false); break;
            default: Q1_LAUNCH_ROLL(float, true, FMT_RANDOM, true, 1, false); break;
        }
    } else if (obs_format == Q1ENV_OBS_F32) {
        // (return_sum, partial output sets, row-format actions: the SPEC tick with run-time output pointers, or the generic tick)
        if (spec && fmt == Q1ENV_ACT_PACKED) Q1_LAUNCH_ROLL(float, true, FMT_PACKED, true, -1, true);
        else if (spec && fmt == Q1ENV_ACT_RANDOM) Q1_LAUNCH_ROLL(float, true, FMT_RANDOM, true, -1, true);
        else if (spec && fmt == Q1ENV_ACT_F32_ROWS) Q1_LAUNCH_ROLL(float, true, FMT_F32_ROWS, true, -1, true);
        else Q1_LAUNCH_ROLL(float, false, FMT_RUNTIME, true, -1, true);
    } else {
        Q1_LAUNCH_ROLL(double, false, FMT_RUNTIME, true, -1, true);
    }
#undef Q1_LAUNCH_ROLL
#undef Q1_LAUNCH_ROLL_D
    HIP_TRY(hipGetLastError());
    if (t_stop) HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->tick_count += (uint64_t)ticks;
    if (s_wait) return signal_wait(h, 30.0);
    return Q1ENV_OK;
}

int q1env_signal_mark(q1env_t* h) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_signal_mark: null handle");
    DeviceGuard guard(h->device);
    if (int r = ensure_signal(h)) return r;
    Signal sg{h->sig_dev, h->ticket_dev, ++h->sig_seq, 1u, 2u};
    hipLaunchKernelGGL(signal_mark_kernel, dim3(1), dim3(64), 0, h->stream, sg);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_signal_wait(q1env_t* h, double timeout_s) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_signal_wait: null handle");
    if (!h->sig_host) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_signal_wait: no signal was requested on this handle");
    return signal_wait(h, timeout_s);
}

int q1env_signal_read(q1env_t* h, uint64_t* start_ticks, uint64_t* end_ticks, double* hz) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_signal_read: null handle");
    if (!h->sig_host) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_signal_read: no signal was requested on this handle");
    if (start_ticks) *start_ticks = h->sig_host[0];
    if (end_ticks) *end_ticks = h->sig_host[1];
    if (hz) *hz = h->wall_clock_hz;
    return Q1ENV_OK;
}

#ifndef Q1_BUILD_ID
#define Q1_BUILD_ID "unknown"
#endif
const char* q1env_build_id(void) { return Q1_BUILD_ID; }

int q1env_observe(q1env_t* h, int obs_format, void* obs) {
    if (!h || !obs) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_observe: null argument");
    DeviceGuard guard(h->device);
    const int blk = block_for(h->p.n);
    if (obs_format == Q1ENV_OBS_F32)
        hipLaunchKernelGGL(observe_kernel<float>, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, (float*)obs, Signal{});
    else if (obs_format == Q1ENV_OBS_F64)
        hipLaunchKernelGGL(observe_kernel<double>, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, (double*)obs, Signal{});
    else return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_observe_host(q1env_t* h, int obs_format, void* obs) {
    if (!h || !obs) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_observe_host: null argument");
    DeviceGuard guard(h->device);
    if (obs_format != Q1ENV_OBS_F32 && obs_format != Q1ENV_OBS_F64) return fail(Q1ENV_ERR_INVALID_ARG, "bad obs_format");
    const size_t bytes = (size_t)h->p.n * 6 * (obs_format == Q1ENV_OBS_F32 ? 4 : 8);
    if ((size_t)h->p.n <= HOST_DIRECT_MAX_ENVS && host_direct_enabled()) {      // host-direct: the kernel writes the rows into host memory
        if (int r = ensure_direct(h, bytes)) return r;
        if (int r = ensure_signal(h)) return r;
        const Signal sg = direct_signal(h, (size_t)h->p.n);
        const int blk = block_for(h->p.n);
        if (obs_format == Q1ENV_OBS_F32)
            hipLaunchKernelGGL(observe_kernel<float>, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, (float*)h->direct_dev, sg);
        else
            hipLaunchKernelGGL(observe_kernel<double>, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, (double*)h->direct_dev, sg);
        HIP_TRY(hipGetLastError());
        if (int r = signal_wait(h, 30.0)) return r;
        memcpy(obs, h->direct_host, bytes);
        return Q1ENV_OK;
    }
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
    if (c <= HOST_DIRECT_MAX_ENVS && host_direct_enabled()) {
        if (int r = ensure_direct(h, in_bytes + bo)) return r;
        if (int r = ensure_signal(h)) return r;
        char* hp = h->direct_host;
        char* dp = h->direct_dev;
        if (idx) memcpy(hp, idx, c * 4);
        memcpy(hp + bi, zero_start, c);
        memcpy(hp + bi + bz, yaw, c * 8);
        memcpy(hp + bi + bz + bd, tm, c * 8);
        memcpy(hp + bi + bz + 2 * bd, speed, c * 8);
        memcpy(hp + bi + bz + 3 * bd, angle, c * 8);
        const Signal sg = direct_signal(h, c);
        const int blk = 64;
        if (obs_format == Q1ENV_OBS_F32)
            hipLaunchKernelGGL(reset_draws_kernel<float>, grid_for((int)count, blk), dim3(blk), 0, h->stream, h->p, h->st, (int)count,
                               idx ? (const int32_t*)dp : nullptr, (const uint8_t*)(dp + bi), (const double*)(dp + bi + bz), (const double*)(dp + bi + bz + bd),
                               (const double*)(dp + bi + bz + 2 * bd), (const double*)(dp + bi + bz + 3 * bd), obs ? (float*)(dp + in_bytes) : nullptr, sg);
        else
            hipLaunchKernelGGL(reset_draws_kernel<double>, grid_for((int)count, blk), dim3(blk), 0, h->stream, h->p, h->st, (int)count,
                               idx ? (const int32_t*)dp : nullptr, (const uint8_t*)(dp + bi), (const double*)(dp + bi + bz), (const double*)(dp + bi + bz + bd),
                               (const double*)(dp + bi + bz + 2 * bd), (const double*)(dp + bi + bz + 3 * bd), obs ? (double*)(dp + in_bytes) : nullptr, sg);
        HIP_TRY(hipGetLastError());
        if (int r = signal_wait(h, 30.0)) return r;
        if (obs) memcpy(obs, hp + in_bytes, no);
        return Q1ENV_OK;
    }
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
                           idx ? d_i : nullptr, d_z, d_y, d_t, d_s, d_a, obs ? (float*)d_o : nullptr, Signal{});
    else
        hipLaunchKernelGGL(reset_draws_kernel<double>, grid_for((int)count, blk), dim3(blk), 0, h->stream, h->p, h->st, (int)count,
                           idx ? d_i : nullptr, d_z, d_y, d_t, d_s, d_a, obs ? (double*)d_o : nullptr, Signal{});
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
    if (n <= HOST_DIRECT_MAX_ENVS && host_direct_enabled()) {    // host-direct: inputs read from, outputs written to host-coherent memory
        if (int r = ensure_direct(h, total)) return r;
        if (int r = ensure_signal(h)) return r;
        char* hp = h->direct_host;
        char* dp = h->direct_dev;
        memcpy(hp, a, act_bytes_a(h, fmt));
        if (has_b) memcpy(hp + ba, b, n * 4);
        memcpy(hp + ba + b4, z_vel, n * 4);
        memcpy(hp + ba + 2 * b4, trem, n * 8);
        const Signal sg = direct_signal(h, n);
        const int blk2 = block_for(h->p.n);
        char* po = dp + in_bytes;
        hipLaunchKernelGGL(decode_kernel, grid_for(h->p.n, blk2), dim3(blk2), 0, h->stream, h->p, h->st, fmt, (const void*)dp, (const void*)(dp + ba),
                           (const float*)(dp + ba + b4), (const double*)(dp + ba + 2 * b4), (double*)po, (int64_t*)(po + b8), (int64_t*)(po + 2 * b8),
                           (uint8_t*)(po + 3 * b8), sg);
        HIP_TRY(hipGetLastError());
        if (int r = signal_wait(h, 30.0)) return r;
        const char* ho = hp + in_bytes;
        memcpy(yaw, ho, n * 8); memcpy(smove, ho + b8, n * 8); memcpy(fmove, ho + 2 * b8, n * 8); memcpy(jump, ho + 3 * b8, n);
        return Q1ENV_OK;
    }
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
                       (const void*)d_b, (const float*)d_zv, (const double*)d_tr, d_y, d_sm, d_fm, d_j, Signal{});
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

}  // extern "C"
