/* q1env.h - C ABI of libq1env.so: the MI355X (gfx950) implementation of the q1physrl env hot path.
 *
 * This is the drop-in boundary.  Everything the reference computes per tick in pure NumPy
 *     VectorPhysEnv.vector_step   q1physrl_env/q1physrl_env/env.py:482-510
 *     ActionDecoder.map           q1physrl_env/q1physrl_env/env.py:225-269
 *     phys.apply                  q1physrl_env/q1physrl_env/phys.py:184-197
 *     _get_obs                    q1physrl_env/q1physrl_env/env.py:392-400
 *     vector_reset / reset_at     q1physrl_env/q1physrl_env/env.py:428-480
 * is behind these entry points as hand-written HIP kernels over an SoA env state resident in HBM.
 * The reference has no FFI of its own (it is pure Python); the binding a maintainer adds is the
 * ctypes stub shown in INTEGRATION.md (shipped as q1physrl_amd/_lib.py).
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (a hipStream_t travels as void*).
 *   - every function returns 0 on success or a negative q1env_status; q1env_last_error() gives the
 *     thread-local message.  No exceptions or aborts cross the ABI.
 *   - "dev" pointers are device (HBM) pointers, e.g. torch_tensor.data_ptr(); "host" are host pointers.
 *   - a handle is bound to one device and one stream; it is not thread-safe; distinct handles are
 *     independent (one handle per process per GPU is the multi-GPU model - no collectives).
 *   - all launches are asynchronous on the handle's stream; only *_host functions and q1env_sync block.
 */
#ifndef Q1ENV_H
#define Q1ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q1ENV_ABI_VERSION 6

/* Environment variables the library reads (ALL of them; each is an A/B measurement knob read once per process, none changes a result bit):
 *   Q1ENV_HOST_DIRECT=0           the small-batch *_host paths stage through copy commands instead of the host-coherent block (csrc/q1env_core.hip)
 *   Q1ENV_ROLLOUT_DEPTH=1|2       how many ticks ahead the rollout kernels request their actions (default: by launch length)
 *   Q1ENV_BLOCK=64|128|256        workgroup size of the env kernels (default 256)
 *   Q1ENV_MLP_THREADS=256|512     one or two waves per SIMD in the policy forward (default: by batch size)
 *   Q1ENV_RESIDENT_TP=1|2         envs per lane pair of the resident sampler
 *   Q1ENV_SERVER_SHAPE=<ES>, Q1ENV_SERVER_BACKOFF=a,b,c      shape / poll pacing of the resident tick server
 * Everything that changes WHAT is computed or HOW workgroups communicate is an explicit call: q1env_learner_set_loss_scale,
 * q1env_learner_set_exchange_mode, q1env_learner_set_profiling (the Q1_LEARNER_* variables of ABI v5 are gone).  The binding additionally
 * honours Q1ENV_LIB_PATH (which build of this library to load: the -DQ1_CHECK assertion build, the host-side AddressSanitizer build). */

typedef enum q1env_status {
    Q1ENV_OK = 0,
    Q1ENV_ERR_INVALID_ARG = -1,
    Q1ENV_ERR_HIP = -2,          /* a HIP runtime call failed (message has hipGetErrorString) */
    Q1ENV_ERR_NO_DEVICE = -3,    /* no gfx950 device visible: the library has NO CPU fallback */
    Q1ENV_ERR_ALLOC = -4
} q1env_status;

/* POD mirror of reference `Config` (env.py:94-148).  Doubles carry python floats unchanged; the
 * library applies the reference's own float32 casts (fmove_max/smove_max env.py:260-261,
 * max_yaw_delta env.py:230) internally. */
typedef struct q1env_config {
    int32_t num_envs;             /* envs owned by THIS handle (this shard)                     */
    int32_t allow_yaw;            /* env.py:107                                                  */
    int32_t discrete_yaw_steps;   /* -1 = continuous mouse dimension (env.py:109)                */
    int32_t speed_reward;         /* env.py:111                                                  */
    int32_t hover;                /* env.py:116                                                  */
    int32_t smooth_keys;          /* env.py:122                                                  */
    int32_t auto_jump;            /* env.py:124                                                  */
    int32_t allow_jump;           /* env.py:126                                                  */
    double zero_start_prob;       /* env.py:100                                                  */
    double initial_yaw_lo;        /* env.py:102 initial_yaw_range[0]                             */
    double initial_yaw_hi;
    double max_initial_speed;     /* env.py:104                                                  */
    double time_delta;            /* env.py:105                                                  */
    double time_limit;            /* env.py:106                                                  */
    double action_range;          /* env.py:108                                                  */
    double fmove_max;             /* env.py:112                                                  */
    double smove_max;             /* env.py:114                                                  */
    double key_press_delay;       /* env.py:118                                                  */
    int64_t env_index_base;       /* global index of this shard's env 0; keys the counter RNG so
                                     results do not depend on how the batch is split over GPUs  */
    int32_t legacy_promotion;     /* env.py:230 `np.float32(720) * time_delta`: 0 = float32 product (NumPy >= 2, NEP 50: what
                                     the golden fixtures pin), 1 = float64 product (the NumPy 1.18.2 of requirements.txt:33).
                                     Identical for time_delta = 1/72; 7.6e-9 relative apart for 0.014, 6e-14 for params.yml's dt */
    int32_t reserved0;            /* must be 0 */
} q1env_config;

/* Action layouts accepted by step/rollout/decode.  A = num_keys + (allow_yaw ? 1 : 0),
 * num_keys = 4, or 3 when auto_jump or !allow_jump (env.py:206-207). */
enum {
    Q1ENV_ACT_F64_ROWS = 0,   /* act_a = double[N][A]: exactly what _fix_actions returns (env.py:221-223) */
    Q1ENV_ACT_F32_ROWS = 1,   /* act_a = float[N][A]:  a torch policy's output                            */
    Q1ENV_ACT_PACKED   = 2,   /* act_a = uint8[N] key bitmask (bit k = Key k, env.py:61-73), act_b = float[N] mouse action: 5 B/env */
    Q1ENV_ACT_RANDOM   = 3    /* rollout only: iid Bernoulli(1/2) keys + U(-action_range, action_range) mouse from the counter RNG */
};

enum {
    Q1ENV_OBS_F64 = 0,        /* double[N][6]: what the reference returns (env.py:399-400, Obs order env.py:76-86) */
    Q1ENV_OBS_F32 = 1         /* float[N][6]: the same values rounded to float32 (what observation_space declares, env.py:416-417) */
};

/* bits of the per-env flag byte (SoA `flags`) */
enum {
    Q1ENV_FLAG_ON_GROUND     = 1u << 0,   /* PlayerState.on_ground     phys.py:160 */
    Q1ENV_FLAG_JUMP_RELEASED = 1u << 1,   /* PlayerState.jump_released phys.py:161 */
    Q1ENV_FLAG_ZERO_START    = 1u << 2,   /* VectorPhysEnv._zero_start env.py:379  */
    Q1ENV_FLAG_LAST_KEY0     = 1u << 3    /* ActionDecoder._last_keys[k] = bit 3+k, env.py:201 */
};

/* Host-side view of the SoA env state for get/set (any pointer may be NULL = skip that array).
 * Each array has num_envs elements; last_key_press_time has 4*num_envs, key-major: [k*num_envs + i]. */
typedef struct q1env_state {
    float*   vel_x;  float* vel_y;  float* vel_z;     /* PlayerState.vel[:,0..2] float32, phys.py:159 */
    double*  pos_x;  double* pos_y;                   /* extension: sum of dt*vel_x / dt*vel_y in float64 (the "100 m" distance) */
    double*  z_pos;                                   /* PlayerState.z_pos, phys.py:158 */
    double*  yaw;                                     /* VectorPhysEnv._yaw, env.py:376 (unwrapped degrees) */
    double*  time_remaining;                          /* VectorPhysEnv._time_remaining, env.py:377 */
    double*  last_key_press_time;                     /* ActionDecoder._last_key_press_time, env.py:200 */
    uint8_t* flags;                                   /* see Q1ENV_FLAG_* */
} q1env_state;

typedef struct q1env q1env_t;

int         q1env_abi_version(void);
const char* q1env_last_error(void);
/* number of gfx950 devices visible; negative status on HIP failure */
int         q1env_device_count(void);

/* Lifetime.  `stream` = a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream),
 * or NULL to let the handle create and own a non-blocking stream.  State after create equals a
 * zero-start reset (env.py:54-58) of every env. */
int q1env_create(const q1env_config* cfg, int device, void* stream, q1env_t** out);
int q1env_destroy(q1env_t* env);
/* Rebind the handle to another stream (no synchronisation: ordering between the old and the new stream is the caller's
 * business, as with torch.cuda.stream()); legal while the new stream is being captured.  Here NULL means the device's default
 * (null) stream - what torch.cuda.current_stream().cuda_stream is (0) unless a side stream is current. */
int q1env_set_stream(q1env_t* env, void* stream);
int q1env_sync(q1env_t* env);
int q1env_num_keys(const q1env_t* env);
int q1env_action_width(const q1env_t* env);
/* (ABI v3) Ticks this handle has stepped since create = the counter the handle-side counter RNG uses for the entry points called
 * WITHOUT a device counter (reset_philox, step_autoreset, rollout, persistent_*; the reference has no counterpart: its resets draw
 * from the global NumPy stream, env.py:432-446).  A caller that also drives entry points WITH a device counter (counter_dev) on the
 * same handle seeds / re-synchronises that counter from this value, so the two families never reuse a (seed, env, counter) triple. */
int q1env_tick_count(const q1env_t* env, uint64_t* out);

/* ---- resets (env.py:428-480; decoder env.py:271-291) ------------------------------------------
 * reset_draws: the caller supplies the raw random draws (the reference's RNG is the GLOBAL NumPy
 * MT19937, env.py:432-446/461-471, which only the Python host can reproduce) and the device turns
 * them into state: yaw / time / speed selection on zero_start, hover override, vel = speed*(cos,sin)(angle)
 * rounded to float32, decoder state cleared.  idx == NULL resets envs 0..n-1.  All pointers HOST.
 * obs (n x 6, obs_format) may be NULL. Synchronous. */
int q1env_reset_draws_host(q1env_t* env, int64_t n, const int32_t* idx, const uint8_t* zero_start,
                           const double* yaw, const double* time_remaining, const double* speed,
                           const double* angle, int obs_format, void* obs);
/* reset_philox: device-side counter RNG (Philox4x32-10 keyed by (seed, global env index, episode
 * counter)) drawing the reference's distributions, including the one-argument uniform(x) quirk
 * (= uniform(low=x, high=1), env.py:439-446).  mask_dev: NULL = all envs; else uint8[N], non-zero = reset.
 * done_only != 0 additionally restricts to envs whose time_remaining < 0 (env.py:506). Asynchronous.
 * counter_dev: NULL = the RNG counter is the handle's host-side tick count; else a device uint64 the kernel reads at run time,
 * so that a captured hipGraph of a sampler loop draws fresh numbers on every replay (the caller increments it on the device). */
int q1env_reset_philox(q1env_t* env, uint64_t seed, const uint64_t* counter_dev, const uint8_t* mask_dev, int done_only,
                       int obs_format, void* obs_dev);

/* ---- the hot path: one tick of every env (env.py:482-510) ------------------------------------
 * Any of obs / reward / done / zero_start may be NULL (not written).  reward float[N] (env.py:500-503),
 * done uint8[N] (env.py:506), zero_start uint8[N] (the info dict's only field, env.py:510). */
int q1env_step(q1env_t* env, int action_format, const void* act_a_dev, const void* act_b_dev,
               int obs_format, void* obs_dev, float* reward_dev, uint8_t* done_dev, uint8_t* zero_start_dev);
/* One tick with in-kernel reset of the envs whose episode ended on it (auto-reset convention of GPU-resident RL loops):
 * reward / done / zero_start describe the finished step, the observation row of a finished env is the first observation of
 * its next episode.  Bit-identical to q1env_step followed by q1env_reset_philox(seed, counter_dev, NULL, done_only = 1).
 * Observations are float32.  counter_dev as in q1env_reset_philox. */
int q1env_step_autoreset(q1env_t* env, int action_format, const void* act_a_dev, const void* act_b_dev, uint64_t seed,
                         const uint64_t* counter_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, uint8_t* zero_start_dev);
/* `ticks` consecutive q1env_step_autoreset launches over tick-major actions (outputs tick-major when out_stride_ticks = 1, a ring of
 * one slab when 0; any may be NULL) - RLlib's "step, then reset what finished" loop at one launch per tick.  Tick t draws its
 * reset randomness with the Philox counter *counter_dev + t, and one last node advances *counter_dev by `ticks`, so the cached
 * hipGraph (use_graph as in q1env_step_many: 0 eager, 1 replay, 2 prepare only) draws fresh numbers on every replay.  counter_dev
 * (device uint64, required) belongs to the caller; bit-identical to the same ticks issued one by one with that counter. */
int q1env_step_autoreset_many(q1env_t* env, int ticks, int action_format, const void* act_a_dev, const void* act_b_dev, uint64_t seed,
                              uint64_t* counter_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, uint8_t* zero_start_dev,
                              int out_stride_ticks, int use_graph);
/* Same with HOST pointers (the NumPy-compatible path): stages H2D, steps, copies back, synchronises.  (ABI v4) Up to 4 096 envs the
 * staging is host-direct: the kernel reads the actions from and writes the results to host-coherent pinned memory itself and the call
 * polls the completion signal - one launch, no copy commands, no stream synchronisation (Q1ENV_HOST_DIRECT=0 in the environment selects
 * the staged form; results are identical).  q1env_reset_draws_host, q1env_observe_host and q1env_decode_host do the same. */
int q1env_step_host(q1env_t* env, int action_format, const void* act_a, const void* act_b,
                    int obs_format, void* obs, float* reward, uint8_t* done, uint8_t* zero_start);
/* `ticks` consecutive single-tick launches with tick-major inputs/outputs ([ticks][N]... ; outputs may be
 * NULL).  use_graph = 1 replays them from a cached hipGraph (launch-bound regime; up to 8 graphs are cached per handle,
 * keyed by every argument); use_graph = 2 only captures and instantiates that graph (no launch, state untouched) so that a
 * later use_graph = 1 call with the same arguments pays no instantiation.  Adding Q1ENV_TIMER_START (4) / Q1ENV_TIMER_STOP (8)
 * to use_graph 0 or 1 records the handle's start / stop timer event immediately before / after the launches inside this call
 * (read the time with q1env_timer_elapsed): the same bracket as q1env_timer_start / q1env_timer_mark without two more calls
 * across the ABI.  out_stride_ticks = 0 makes every tick overwrite the same output slab (ring of 1), 1 = tick-major slabs. */
#define Q1ENV_TIMER_START 4
#define Q1ENV_TIMER_STOP 8
int q1env_step_many(q1env_t* env, int ticks, int action_format, const void* act_a_dev, const void* act_b_dev,
                    int obs_format, void* obs_dev, float* reward_dev, uint8_t* done_dev,
                    int out_stride_ticks, int use_graph);
/* Fused multi-tick kernel: state stays in registers for `ticks` ticks (one launch).  Actions are
 * tick-major device arrays in `action_format`, or Q1ENV_ACT_RANDOM (then rng_seed keys the counter RNG).
 * Per-tick outputs are tick-major and optional (NULL).  auto_reset bit 0: an env whose tick set `done`
 * is reset in-kernel with the Philox reset (as q1env_reset_philox) before its next tick; (ABI v3) Q1ENV_TIMER_START (4) /
 * Q1ENV_TIMER_STOP (8) may be added to record the handle's timer events around the launch inside this call (as q1env_step_many).
 * return_sum_dev (optional, double[N]) accumulates reward over the launch in float64.
 * Which kernel serves a call (all give identical results; the cost differs): the SPECIALISED instantiations - straight-line
 * tick, static store count, 237 vector instructions per tick - need (a) the default action / episode structure (4 keys, continuous mouse,
 * jump key, no hover, y reward: Config.get_default() and data/params.yml) with move maxima and constants that pass the host's exactness
 * checks (csrc/q1env_host.hpp is_spec: e.g. time_limit and action_range whose reciprocals satisfy the one-step division bound), (b)
 * float32 observations, (c) packed or on-device random actions, (d) either ALL of obs / reward / done or none of them, (e) no
 * return_sum_dev.  Anything else runs the same tick through run-time output pointers (return_sum, partial output sets, row actions: ~5 %
 * slower) or the generic wave-uniform-branch kernels (other Configs: ~1.3 x).  From 32 ticks per launch the specialised all-outputs
 * kernels request the action two ticks ahead (DEPTH = 2 in the kernel's name; Q1ENV_ROLLOUT_DEPTH=1|2 in the environment forces one).
 * (ABI v4) Q1ENV_STAMP_START (16) / Q1ENV_SIGNAL (32) / Q1ENV_SIGNAL_WAIT (64) in auto_reset: the completion signal below, written by
 * this launch's own waves - no marker packets around the kernel and no runtime synchronisation to learn that it has finished. */
#define Q1ENV_STAMP_START 16
#define Q1ENV_SIGNAL 32
#define Q1ENV_SIGNAL_WAIT 64
int q1env_rollout(q1env_t* env, int ticks, int action_format, const void* act_a_dev, const void* act_b_dev,
                  uint64_t rng_seed, int obs_format, void* obs_dev, float* reward_dev, uint8_t* done_dev,
                  int auto_reset, double* return_sum_dev);

/* ---- completion signal + device time stamps (ABI v4) ---------------------------------------------------------------------------
 * A 20-tick launch of 65 536 envs is ~20 us of GPU work; hipEventRecord x 2 + a runtime synchronisation around it cost more host
 * time than that (the driver's line was 54 % host latency).  The handle therefore owns three words of host-coherent
 * pinned memory the KERNEL writes itself:
 *   start stamp   the device's constant-rate wall clock (wall_clock64 = s_memrealtime) read by the first wave of a launch made with
 *                 Q1ENV_STAMP_START
 *   end stamp     the same clock read by the LAST wave to finish of a launch made with Q1ENV_SIGNAL: every wave, after its final
 *                 stores have been ACKNOWLEDGED (s_waitcnt vmcnt(0)), takes a ticket from a two-level tree of relaxed device
 *                 counters; the wave that draws the last root ticket knows all others' stores have been acknowledged
 *   sequence      that wave then stores the launch's sequence number with system-scope release: the host thread sees it over PCIe
 *                 ~1 us later by polling its own cache line - q1env_signal_wait - instead of waiting for the dispatch's completion
 *                 interrupt to travel through the runtime.
 * What the signal GUARANTEES (ABI v5).  q1env_rollout writes every per-tick output (obs / reward / done), the final
 * env state and return_sum with SYSTEM-scope write-through stores (sc0 sc1 on gfx950): such a store is acknowledged by the memory
 * side (Infinity Cache / HBM), not by the issuing XCD's private L2, so "every wave's stores are acknowledged" = "every result of the
 * launch is readable by any agent" - a DMA copy on another stream, the host through a mapped pointer, a kernel on another XCD or
 * another GPU - with NO further stream synchronisation, event or cache write-back.  The tickets need no release / acquire for this
 * (it is the acknowledgement that carries the data to memory, not a cache write-back ordered by a fence), which is why they are
 * relaxed.  tests/test_hip_signal.py::test_signal_carries_visibility_* copy the outputs out through a second, non-blocking stream the
 * instant q1env_signal_wait returns (no runtime synchronisation in between), 1 000 times at 65 536 envs x 20 ticks, and compare.
 * q1env_signal_mark's signal sits behind the stream's earlier kernels, whose end-of-kernel release has completed when it runs: the
 * same guarantee for regions that end in another kernel.  The host-direct small-batch calls use a stricter per-wave system-scope
 * release before an acquire-release ticket (their results live in host memory).
 * Q1ENV_SIGNAL_WAIT makes the launching call itself poll until the signal arrives (one call across the ABI for launch + wait).
 * q1env_signal_mark: the same end stamp + sequence from a one-wave kernel enqueued behind whatever the stream holds (for regions that
 * do not end in a q1env_rollout launch).  q1env_signal_wait: poll (no sleep; acquire load) until the last signal requested on this
 * handle has arrived, or timeout_s passes (-> Q1ENV_ERR_HIP after draining the stream, message says so).  q1env_signal_read: the two
 * stamps of the last signalled region and the clock's rate in Hz (hipDeviceAttributeWallClockRate).  No reference counterpart (the
 * reference is synchronous NumPy: env.py:507-510 hands the caller arrays it can read - which is what the guarantee above restores). */
int q1env_signal_mark(q1env_t* env);
int q1env_signal_wait(q1env_t* env, double timeout_s);
int q1env_signal_read(q1env_t* env, uint64_t* start_ticks, uint64_t* end_ticks, double* ticks_per_second);
/* (ABI v4) 16 hex digits identifying the sources this library was built from (sha256 over csrc/ and include/q1env.h and the compile
 * flags, computed by q1physrl_amd/build.py): profiles/pmc.json records it, bench.py refuses counters taken from another build. */
const char* q1env_build_id(void);

/* current observation without stepping (env.py:392-400) */
int q1env_observe(q1env_t* env, int obs_format, void* obs_dev);
int q1env_observe_host(q1env_t* env, int obs_format, void* obs);

/* SoA state exchange with HOST arrays (checkpoint / golden-state injection / player_state snapshots) */
int q1env_get_state_host(q1env_t* env, const q1env_state* dst);
int q1env_set_state_host(q1env_t* env, const q1env_state* src);
/* Device-side copy of the whole env state (all SoA arrays, one device-to-device copy on the handle's stream) and its
 * restoration.  No reference counterpart: the host mirror uses it to undo speculative work (q1physrl_amd.env.VectorPhysEnv with
 * speculative_resets=True resets every env that finished on a tick at the first reset_at call and rolls back what the caller
 * did not claim). */
int q1env_snapshot_state(q1env_t* env);
int q1env_restore_state(q1env_t* env);

/* device pointers of the live SoA arrays (zero-copy views for torch); valid until destroy */
int q1env_state_device_ptrs(q1env_t* env, q1env_state* out);

/* ---- stand-alone pieces of the reference's public surface -----------------------------------
 * ActionDecoder.map (env.py:225-269) as used on its own by mkdemo.py:47-55 / analyse.py:199-207: the
 * decoder state (last_key_press_time, last_keys, yaw) is this handle's; z_vel / time_remaining come
 * from the caller.  Outputs HOST: yaw double[N], smove/fmove int64[N], jump uint8[N]. Synchronous. */
int q1env_decode_host(q1env_t* env, int action_format, const void* act_a, const void* act_b,
                      const float* z_vel, const double* time_remaining,
                      double* yaw, int64_t* smove, int64_t* fmove, uint8_t* jump);
/* ActionDecoder.vector_reset / reset_at (env.py:271-291): idx NULL = envs 0..n-1. */
int q1env_decoder_reset_host(q1env_t* env, int64_t n, const int32_t* idx, const double* yaw);
/* phys.apply (phys.py:184-197), stateless, HOST arrays of n elements (vel: float[n][3]).  pitch/roll may
 * be NULL (= 0, what the env passes, env.py:490-491).  fmove/smove are doubles (integers convert exactly). */
int q1phys_apply_host(int device, int64_t n, const double* yaw, const double* pitch, const double* roll,
                      const double* fmove, const double* smove, const uint8_t* button2, const double* time_delta,
                      const double* z_pos, const float* vel, const uint8_t* on_ground, const uint8_t* jump_released,
                      double* out_z_pos, float* out_vel, uint8_t* out_on_ground, uint8_t* out_jump_released);
/* The same for a float64 velocity (vel: double[n][3]).  The reference's arithmetic follows the dtype of PlayerState.vel:
 * the env stores float32 (friction speed, the +270 jump add and the store are float32), but PlayerState.from_df
 * (phys.py:168-170, the demo-analysis path of analyse.py:102-118) yields float64 and then nothing is rounded to float32. */
int q1phys_apply_host_f64(int device, int64_t n, const double* yaw, const double* pitch, const double* roll,
                          const double* fmove, const double* smove, const uint8_t* button2, const double* time_delta,
                          const double* z_pos, const double* vel, const uint8_t* on_ground, const uint8_t* jump_released,
                          double* out_z_pos, double* out_vel, uint8_t* out_on_ground, uint8_t* out_jump_released);

/* Page-locked host memory for arrays passed to the *_host entry points (any host pointer is accepted there; arrays from
 * here are copied by direct DMA instead of the runtime's staged pageable path - at 1 M envs a tick moves 98 MB).  NULL on
 * failure (q1env_last_error says why).  Batches of at most 16 384 envs are packed through the handle's own pinned staging
 * instead (one copy each way), so small callers need not bother. */
void* q1env_host_alloc(uint64_t bytes);
int   q1env_host_free(void* p);

/* ---- policy-side glue for a GPU-resident sampler loop ------------------------------------------
 * Counterpart of the reference's TF action distribution `Q1PhysActionDist` (q1physrl/action_dist.py:46-243):
 * one row of policy-network outputs per env -> a sampled action in the PACKED layout q1env_step consumes, plus
 * its log-probability.  logits: float[N][row_stride] device, row = num_keys x (logit0, logit1) then (mean, log_std)
 * of the CDF-squashed Gaussian over (-action_range, action_range); with Config.discrete_yaw_steps = S > 0 the mouse child is
 * Discrete(2S+1) (env.py:216-219; ModelCatalog's Categorical, action_dist.py:221-222) and the row ends with its 2S+1 logits
 * instead - the sampled step index 0..2S is then what `mouse` holds (as a float: the packed layout's mouse slot).  With
 * allow_yaw = False the row is the key pairs only.  keys uint8[N], mouse float[N], logp float[N]
 * (logp may be NULL).  deterministic != 0: arg-max keys / arg-max step and the squashed mean (action_dist.py:84-89).
 * Randomness: Philox keyed by (seed, global env index, counter + (counter_dev ? *counter_dev : 0)). Asynchronous. */
int q1env_policy_sample(q1env_t* env, const float* logits_dev, int row_stride, uint64_t seed, uint64_t counter,
                        const uint64_t* counter_dev, int deterministic, uint8_t* keys_dev, float* mouse_dev, float* logp_dev);

/* Generalised advantage estimation over tick-major device arrays of this handle's N envs (learner-side glue):
 * reward float[T][N], value float[T+1][N] (bootstrap row last), done uint8[T][N] -> adv, vtarg float[T][N]. */
int q1env_gae(q1env_t* env, int ticks, const float* reward_dev, const float* value_dev, const uint8_t* done_dev,
              float gamma, float lam, float* adv_dev, float* vtarg_dev);

/* PPO loss of one minibatch and its gradient with respect to the policy outputs, in one kernel (learner side; the closed forms
 * of RLlib 0.8.4's PPOLoss, as configured by q1physrl/train.py:60-64 + data/params.yml:4-13, over the reference's Q1PhysActionDist,
 * q1physrl/action_dist.py:46-243, differentiated by hand).  Per sample b of `batch`: logits / old_logits float[batch][row_stride]
 * (current and behaviour policy outputs), keys uint8 bit mask + mouse float (the action taken), logp_old, adv (already
 * standardised), value (current), value_old, vtarg.  Outputs: dlogits float[batch][row_stride] and dvalue float[batch] =
 * d mean(total loss) / d(logits, value) - feed them to autograd.backward of the networks; partials float[ceil(batch/256)][5] =
 * per-block sums of (entropy, kl, policy loss, total loss, vf loss).  kl_coeff_dev: device scalar (it changes between updates
 * while a captured graph replays this launch).  The handle supplies num_keys / action_range and the stream. */
int q1env_ppo_loss_grad(q1env_t* env, int64_t batch, const float* logits_dev, const float* old_logits_dev, int row_stride,
                        const uint8_t* keys_dev, const float* mouse_dev, const float* logp_old_dev, const float* adv_dev,
                        const float* value_dev, const float* value_old_dev, const float* vtarg_dev, float clip_param,
                        float vf_clip_param, float vf_loss_coeff, float entropy_coeff, const float* kl_coeff_dev,
                        float* dlogits_dev, float* dvalue_dev, float* partials_dev);

/* One sampler tick after the policy forward in ONE launch: q1env_policy_sample (counter) -> q1env_step_autoreset with the
 * sampled packed action -> q1env_episode_stats; bit-identical to that sequence of three calls.  Writes the sampled action
 * and its log-probability (keys uint8[N], mouse float[N], logp float[N]: the trajectory the learner needs), the step's reward /
 * done / zero_start and the next observation row (float[N][6], fresh first observation for envs that were reset), and updates
 * ep_return / partials as q1env_episode_stats does.  RNG counter = counter_offset + (*counter_dev if counter_dev else the
 * handle's host-side tick count): a captured horizon passes the tick index as counter_offset and advances *counter_dev once
 * per horizon. */
int q1env_sample_step(q1env_t* env, const float* logits_dev, int row_stride, uint64_t seed, const uint64_t* counter_dev,
                      uint64_t counter_offset, int deterministic, uint8_t* keys_dev, float* mouse_dev, float* logp_dev,
                      float* obs_dev, float* reward_dev, uint8_t* done_dev, uint8_t* zero_start_dev, double* ep_return_dev,
                      double* partials_dev);

/* Fused forward pass of one network of the reference policy's shape (RLlib fcnet of data/checkpoints/wr: 6 -> 256 tanh ->
 * 256 tanh -> out_dim, out_dim = 10 policy logits or 1 value) for this handle's N envs: obs float[N][6] -> out float[N][out_dim].
 * w1 float[256][6], b1 float[256], b2 float[256], b3 float[out_dim] in torch nn.Linear layout.  w23_image: W2 (nn.Linear(256,256)
 * weight, row = output unit) followed by W3 (nn.Linear(256,out_dim) weight in rows 0..out_dim-1 of a 32-row tile) as ONE float16 (IEEE binary16)
 * array of 288 rows x 264 elements: 256 weights + 8 zero pad per row, the columns of every row permuted so that within each
 * group of 16 the four groups of four are stored in the order 0,2,1,3, and the W2 rows (not W3) pre-multiplied by 2*log2(e)
 * before the float16 rounding - the kernel's tanh takes its exp2 argument straight from the accumulator
 * (q1physrl_amd.policy.FusedPolicyForward builds the image).
 * All three layers run on the matrix cores with float16 weights and float32 accumulation; layer 1 takes the observations and its
 * bias split into two float16 each (hi + lo, 22 mantissa bits), b2 / b3 and tanh are float32, hidden activations are rounded to
 * float16.  1 <= out_dim <= 32 (one 32-row output tile: 10 = continuous-mouse policy, 2K + 2S+1 = discrete-mouse policy, 1 = value).  Inference only (sampler loop); the learner keeps its float32 torch modules. */
int q1env_policy_forward(q1env_t* env, const float* obs_dev, const float* w1_dev, const float* b1_dev, const uint16_t* w23_image_dev,
                         const float* b2_dev, const float* b3_dev, int out_dim, float* out_dev);

/* The policy and the value network of one sampler tick (train.py:60-64: RLlib's fcnet with vf_share_layers = False is two
 * such networks over the same observation) in ONE launch: half of the CUs evaluate *pi, the other half *vf.  Same arithmetic
 * as two q1env_policy_forward calls, bit for bit. */
typedef struct q1env_mlp {
    const float* w1;              /* float[256][6]   device */
    const float* b1;              /* float[256]      device */
    const uint16_t* w23_image;    /* float16[288][264] bits, device, layout as above */
    const float* b2;              /* float[256]      device */
    const float* b3;              /* float[out_dim]  device */
    float* out;                   /* float[N][out_dim] device */
    int out_dim;
} q1env_mlp;
int q1env_policy_value_forward(q1env_t* env, const float* obs_dev, const q1env_mlp* pi, const q1env_mlp* vf);

/* ---- native PPO learner step (ABI v3) ---------------------------------------------------------------------------------------
 * Counterpart of the RLlib PPOTrainer's SGD step the reference configures (q1physrl/train.py:60-64, data/params.yml:4-13) for the two
 * fcnet networks of the published checkpoint's shape (obs 6 -> 256 tanh -> 256 tanh -> out; policy and value network separate):
 * forward, PPO loss gradient (q1env_ppo_loss_grad's closed forms), backward and weight gradients as hand-written gfx950 kernels -
 * float16 matrix-core operands, float32 accumulation; the float32 master weights, the gradients and the optimizer state stay with the
 * caller (torch Linear layouts: w1 float[256][6], b1 float[256], w2 float[256][256], b2 float[256], w3 float[out][256], b3 float[out]).
 *   q1env_learner_workspace_bytes   size of the device scratch one (minibatch, splits) shape needs (weight images, float16 activations
 *                                   of the minibatch in two layouts, split-K partial sums); the caller allocates it once
 *   q1env_learner_images            float32 masters -> the float16 weight images of both directions (call after every optimizer step)
 *   q1env_learner_forward           logits float[B][out_pi] / value float[B] of rows idx[0..B) of obs (idx NULL: rows 0..B-1);
 *                                   bit-identical to q1env_policy_value_forward on the gathered rows; keeps the activations
 *   q1env_learner_backward          gradients of sum_i <dlogits_i, logits_i> + dvalue_i value_i w.r.t. all twelve tensors, divided by
 *                                   grad_scale (pass dlogits / dvalue PRE-MULTIPLIED by grad_scale, e.g. B: they travel as float16)
 *   q1env_learner_step              forward + PPO loss gradient + backward of one minibatch: gw* / gb* = d mean-loss / d parameter,
 *                                   stats_partials as q1env_ppo_loss_grad; then run the optimizer and q1env_learner_images
 * `splits` = workgroups per network of the split-K weight-gradient kernel (64 is a good value on an MI355X); it fixes the workspace
 * size, so the same value goes into every call.  All launches are asynchronous on the handle's stream; nothing is allocated. */
typedef struct q1env_learner_net {
    const float* w1; const float* b1; const float* w2; const float* b2; const float* w3; const float* b3;   /* device, float32 masters */
    float* gw1; float* gb1; float* gw2; float* gb2; float* gw3; float* gb3;                                  /* device, gradients out  */
    int out_dim;
} q1env_learner_net;
typedef struct q1env_learner_batch {
    int64_t minibatch;                 /* B */
    const int64_t* idx_dev;            /* int64[B]: rows of the arrays below that form the minibatch (NULL: rows 0..B-1) */
    const int64_t* idx_cursor_dev;     /* optional device scalar: the minibatch is idx_dev[*cursor .. *cursor + B) - a whole epoch's permutation stays in
                                        * idx_dev and the step replays from a captured graph without a copy per minibatch.  q1env_learner_adam keeps such
                                        * a cursor at byte 72 of its adam_state (+= B per call; the caller zeroes it when it loads a new permutation) */
    const float* obs_dev;              /* float[total][6] */
    const float* old_logits_dev;       /* float[total][old_stride]: the behaviour policy's outputs */
    int old_stride;
    const uint8_t* keys_dev;           /* uint8[total] packed key actions */
    const float* mouse_dev;            /* float[total] mouse actions (NULL without a mouse) */
    const float* logp_old_dev; const float* adv_dev; const float* value_old_dev; const float* vtarg_dev;   /* float[total] each */
    float clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff;
    const float* kl_coeff_dev;         /* device scalar */
    float* stats_partials_dev;         /* float[ceil(B/256)][5]: sums of (entropy, kl, -surrogate, total, vf) per block of 256 samples */
    int skip_reduce;                   /* != 0: leave the split-K partial sums in the workspace for q1env_learner_adam (gw* / gb* untouched) */
    uint32_t* saturation_dev;          /* (ABI v4) optional uint32[4], accumulating until the caller zeroes it: [0] += (lane, launch) pairs of the
                                        * POLICY network's backward pass that converted a gradient element (dY, dZ2, dZ1 - per-sample gradients
                                        * times the float16 loss scale) beyond float16's largest finite value 65504 and clamped it - a per-sample
                                        * gradient clip the reference (RLlib, grad_clip None) does not apply, so it is counted, not hidden;
                                        * [1] = max of the float32 BITS of the largest |element| seen before the clamp; [2], [3] the same for the
                                        * value network */
} q1env_learner_batch;
/* (ABI v5) The float16 loss scales of every q1env_learner_* call made on this handle afterwards: per-sample gradients of the policy network
 * travel multiplied by pi_upscale, those of the value network divided by value_downscale, and the sums are divided / multiplied back in
 * float32 (csrc/q1learner.hpp "Gradient scaling") - exact for powers of two, which is all this accepts; 0 = the default (256 and 1, or
 * Q1_LEARNER_PI_UPSCALE / Q1_LEARNER_VALUE_DOWNSCALE from the environment).  For a caller that adapts the scale to the gradient magnitudes
 * q1env_learner_batch.saturation_dev reports (q1physrl_amd/ppo.py: chosen per update from the previous update's largest element, so that
 * nothing saturates: RLlib has grad_clip = None, data/params.yml:2-13).  A q1env_learner_step / q1env_learner_adam pair must run under the
 * same setting; a captured graph bakes the scales in. */
int q1env_learner_set_loss_scale(q1env_t* env, float pi_upscale, float value_downscale);
uint64_t q1env_learner_workspace_bytes(int64_t minibatch, int out_dim_pi, int splits);
int q1env_learner_images(q1env_t* env, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int64_t minibatch, int splits);
int q1env_learner_forward(q1env_t* env, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int64_t minibatch, int splits,
                          const float* obs_dev, const int64_t* idx_dev, float* logits_out_dev, float* value_out_dev);
int q1env_learner_backward(q1env_t* env, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int64_t minibatch, int splits,
                           const float* obs_dev, const int64_t* idx_dev, const float* dlogits_dev, const float* dvalue_dev, float grad_scale);
int q1env_learner_step(q1env_t* env, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int splits,
                       const q1env_learner_batch* batch);
/* The optimizer of a single-process run, fused with the gradient reduction and the weight-image rebuild (one pass over the 138 k
 * parameters after a q1env_learner_step with skip_reduce): torch.optim.Adam's update (no weight decay, no amsgrad) on the float32
 * masters IN PLACE, moments and the step count in the caller's adam_state_dev (q1env_learner_adam_state_bytes bytes, zero-initialised;
 * the step count lives on the device, so the call is replayable from a captured graph); gw* / gb* receive the gradients too.
 * grad_scale = the minibatch size B of the q1env_learner_step (skip_reduce) that left the partial sums.  The step's float16 loss scales
 * are applied INTERNALLY on top of it (policy network: x pi_upscale, default 256; value network: / value_downscale, default 1;
 * csrc/q1env_learner.hip learner_pi_upscale / learner_value_downscale): the partial sums this function consumes are the ones
 * q1env_learner_step leaves - those of q1env_learner_backward carry grad_scale only and are NOT valid input here.  vf->out_dim must be 1
 * (the state block is sized for a scalar value head).  adam_state layout: int64 step count at byte 0,
 * float bias corrections [2] at byte 8, float running statistics [5] at byte 16 (+= the mean of stats_partials_dev - the step's
 * q1env_learner_batch.stats_partials_dev - per call, if not NULL; the caller zeroes them when it starts a new average), an int64 minibatch
 * cursor at byte 72 (+= minibatch per call: q1env_learner_batch.idx_cursor_dev may point at it), moments from 256. */
uint64_t q1env_learner_adam_state_bytes(int out_dim_pi);
int q1env_learner_adam(q1env_t* env, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int64_t minibatch, int splits,
                       float grad_scale, float lr, float beta1, float beta2, float eps, void* adam_state_dev, const float* stats_partials_dev);

/* The whole SGD step of a single-process run as ONE call (round 4): q1env_learner_step with skip_reduce, then q1env_learner_adam with
 * grad_scale = minibatch = batch->minibatch - the same masters, moments, weight images and gw* / gb* to the last bit - in four launches
 * instead of six: for the reference's action structure (4 keys + continuous mouse) the backward kernel computes the PPO loss gradient
 * of its own samples (no loss kernel, no dlogits round trip) and the optimizer's bookkeeping rides in the backward and Adam kernels (no
 * bookkeeping kernel); other action structures run the two calls as they are.  The step's statistics are folded into adam_state's running
 * sums (bytes 16..35) as q1env_learner_adam does - summed per workgroup of the backward kernel, i.e. equal to the two-call path's up
 * to the order of the float32 additions; batch->stats_partials_dev and batch->skip_reduce are not used (stats_partials_dev may be NULL
 * for the reference's action structure).  Replayable from a captured graph. */
int q1env_learner_sgd_step(q1env_t* env, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int splits,
                           const q1env_learner_batch* batch, float lr, float beta1, float beta2, float eps, void* adam_state_dev);
/* (ABI v6) The kernel sequence of q1env_learner_sgd_step for the reference's action structure (csrc/q1learner_fused.hpp).  0 = automatic
 * (default): from 2 048 samples on, forward + PPO loss gradient + data gradients run as ONE kernel - the float16 activations are consumed where
 * they are produced instead of travelling through memory between two launches, and dZ1 and tanh(H2) - which the weight-gradient kernel needs
 * only for the 256 x 7 and out x 256 products dW1 / db1 and dW3 - are replaced by those products per 32-sample tile (mode 3); smaller minibatches
 * keep round 4's four launches (mode 1).  2 = the fused kernel with dZ1 and tanh(H2) stored as before: every result bit-identical to mode 1 (what the
 * tests hold the fused kernel to).  Mode 3 differs from 1 / 2 in dW1, db1 and dW3 only, by float32 summation order (per-tile products added in
 * tile order instead of one accumulation chain).  Three launches per step instead of four.  The fused kernel serves minibatches of up to
 * 262 144 samples: beyond, automatic mode keeps the four launches and modes 2 / 3 are refused by q1env_learner_sgd_step.  The workspace is the
 * same for every mode; a captured graph bakes the mode in. */
int q1env_learner_set_step_mode(q1env_t* env, int mode);

/* ---- persistent learner (ABI v5, extended in v6) -------------------------------------------------------------------------------
 * `steps` SGD steps of 128-sample minibatches - forward, PPO loss gradient, backward, weight gradients, Adam, for both networks - as
 * ONE dispatch of 2 x 8 co-operating workgroups (csrc/q1learner_persist.hpp): what RLlib's PPO does 11 719 times per training
 * iteration under the reference's configuration (sgd_minibatch_size 128 x num_sgd_iter 30 over train_batch_size 50 000:
 * q1physrl/train.py:60-64, data/params.yml:4-13), where a four-launch q1env_learner_sgd_step costs ~43 us per step and a step's
 * arithmetic is ~0.1 GFLOP.  batch: as for q1env_learner_sgd_step with minibatch == 128 and the reference's action structure (4 keys +
 * continuous mouse; anything else -> Q1ENV_ERR_INVALID_ARG, use q1env_learner_sgd_step); idx_dev holds the whole schedule: step n
 * trains on rows idx_dev[(n / steps_per_epoch) * epoch_stride + (n % steps_per_epoch) * 128 + (0..127)] (one permutation of the train
 * batch per epoch, epoch_stride >= 128 steps_per_epoch apart; NULL = the rows themselves); idx_cursor_dev, stats_partials_dev and
 * skip_reduce are not used.  Masters, gw* / gb* (the LAST step's gradients) and the moments / step count of adam_state_dev
 * (q1env_learner_adam's layout) are updated as by `steps` calls of q1env_learner_sgd_step - same float16-operand / float32-accumulate
 * recipe and loss scales, different summation order: equal to float16 operand rounding, not bit for bit.  adam_state's running
 * statistics [0], [1], [2], [4] (entropy, kl, policy loss, vf loss) grow by the sum over the steps of the per-step means; [3] (total) is
 * NOT updated - it is policy + kl_coeff kl + vf_loss_coeff vf - entropy_coeff entropy of the others.  The workspace images of
 * q1env_learner_images are NOT refreshed: call it (and rebuild any sampler-side weight image) after the launch.
 * batch_rows = the number of rows of the train batch arrays (every idx value is below it): a first, tiny launch computes the
 * squashed-Gaussian pre-image of all mouse actions once.  pws_dev: q1env_learner_persistent_bytes(batch_rows) bytes of device memory
 * (exchange buffers, barrier counters, the W2 slices' optimizer state in owner-lane order, status).  The 16 workgroups spin
 * on group barriers and must be co-resident (they are whenever 16 CUs are free); every wait is bounded by timeout_s (<= 0: 5 s) and a
 * timeout is reported by q1env_learner_persistent_status: status4[0] != 0 (1 + index of the barrier within the step), [1] the step.
 * Asynchronous on the handle's stream like every launch.
 * (ABI v6) idx_rows = the number of int64 entries of batch->idx_dev (ignored when it is NULL): a schedule that would read past it - or, without
 * an index list, past batch_rows - is refused with Q1ENV_ERR_INVALID_ARG instead of being read (the VALUES of idx_dev are the caller's
 * contract; the assertion build of the library, -DQ1_CHECK, compares every one of them with batch_rows on the device).  The step count the
 * launch starts from is snapshotted by its first kernel, so that the two networks' workgroup groups - which never synchronise with each other -
 * apply the same Adam bias correction however late one of them starts.
 *
 * Exchange mode (ABI v6; csrc/q1learner_persist.hpp "Exchange protocol").  0 = automatic (default): every launch takes a census of the XCD each
 * of a network's eight workgroups runs on; if all eight share one, they exchange through that XCD's L2 (plain stores, device-scope loads,
 * barrier tickets counted in the L2: ~11.8 us per step), otherwise through agent-scope write-through stores and invalidations (what the
 * language's memory model backs on any placement: ~16.7 us).  1 = agent scope always.  2 = automatic with a census that is MADE to disagree
 * (tests: the fallback is then chosen by the same code path a partitioned device or a different dispatcher would take).  The mode each group
 * actually ran in is status4[2] (policy) / status4[3] (value): 0 = agent scope, otherwise 1 + the XCD the group shared.
 * q1env_learner_set_profiling: -1 = off (default); g + 8 w = launch the instantiation that stamps wave w of workgroup g of the policy group
 * (10-ns ticks per phase, summed over the steps, as uint64[20] at byte 24 of the status line: tools/time_learner_persistent.py).
 * q1env_learner_persistent_layout: byte offsets, inside pws_dev, of network `net`'s (0 policy, 1 value) exchange buffers - {barrier line, b3,
 * H1, H1^T, dZ2, W2^T, partial logits, W2 optimizer state} - and, ninth, the bytes of the group's exchange workspace (inspection / tests:
 * tests/test_hip_learner.py reads back the rows a step actually gathered).
 * q1env_learner_debug_counters: out5 = {1 if built with -DQ1_CHECK else 0, exchange accesses checked, row indices checked, barrier readings
 * checked, assertions failed}; an assertion that fails is also reported as status4[0] = 0x100 + code (0x101 exchange offset outside the
 * group's workspace, 0x102 row index >= batch_rows, 0x103 schedule position outside idx_dev, 0x104 barrier reading outside its range),
 * status4[1] = the offending value - and the access is skipped: a status word, not a memory fault.  Synchronises the stream. */
uint64_t q1env_learner_persistent_bytes(int64_t batch_rows);
int q1env_learner_sgd_epochs(q1env_t* env, const q1env_learner_net* pi, const q1env_learner_net* vf, void* pws_dev, const q1env_learner_batch* batch,
                             int64_t batch_rows, int64_t idx_rows, int64_t steps, int64_t steps_per_epoch, int64_t epoch_stride, float lr, float beta1,
                             float beta2, float eps, void* adam_state_dev, double timeout_s);
/* (ABI v6) The same update in FLOAT32 arithmetic (csrc/q1learner_persist32.hpp): float32 matrix operands (v_mfma_f32_32x32x2_f32) and exchange
 * buffers, IEEE square root / division in the optimizer, no loss scale, nothing saturates - what RLlib / TF PPO computes (float32 end to end,
 * grad_clip = None: q1physrl/train.py:60-64, data/params.yml:4-13), up to summation order: gradients within 2e-5 of torch autograd's.  Same
 * arguments, workspace, status words and exchange modes as q1env_learner_sgd_epochs; q1env_learner_set_loss_scale does not apply and
 * batch->saturation_dev is not written.  About twice the time per step: the control the float16 learner's training results are compared
 * with (STATE.md "fp32 control"), and the learner for anyone who wants the reference's arithmetic. */
int q1env_learner_sgd_epochs_f32(q1env_t* env, const q1env_learner_net* pi, const q1env_learner_net* vf, void* pws_dev, const q1env_learner_batch* batch,
                                 int64_t batch_rows, int64_t idx_rows, int64_t steps, int64_t steps_per_epoch, int64_t epoch_stride, float lr, float beta1,
                                 float beta2, float eps, void* adam_state_dev, double timeout_s);
int q1env_learner_persistent_status(q1env_t* env, const void* pws_dev, uint32_t* status4_host);   /* synchronises the stream */
int q1env_learner_set_exchange_mode(q1env_t* env, int mode);
int q1env_learner_set_profiling(q1env_t* env, int wave_of_group);
int q1env_learner_persistent_layout(int64_t batch_rows, int net, uint64_t* offsets9_host);
int q1env_learner_debug_counters(q1env_t* env, uint64_t* out5_host);

/* Episode bookkeeping of one sampler tick (the reference's on_episode_end metric hook, q1physrl/train.py:54-57):
 * ep_return double[N] += reward; for envs with done != 0 the finished return is added to this wave's slot of
 * partials (double[ceil(N/64)][4] = episodes, zero-start episodes, return sum, zero-start return sum) and ep_return is
 * cleared.  No atomics: sums are bit-reproducible; add the slots up on the host. */
int q1env_episode_stats(q1env_t* env, const float* reward_dev, const uint8_t* done_dev, const uint8_t* zero_start_dev,
                        double* ep_return_dev, double* partials_dev);

/* The same forward for `rows` observation rows that need not be this handle's batch: obs float[rows][6] -> net->out
 * float[rows][out_dim] (e.g. the value network over the (T + 1) N observations a resident sampling horizon stored). */
int q1env_policy_forward_rows(q1env_t* env, uint64_t rows, const float* obs_dev, const q1env_mlp* net);

/* ---- resident sampler (experimental) -------------------------------------------------------------
 * A whole sampling horizon - `ticks` times { policy forward, sample the action, step, reset finished episodes, episode
 * statistics } - as ONE dispatch: the counterpart of `ticks` x (q1env_policy_value_forward + q1env_sample_step) without the
 * value network, which is evaluated afterwards over the stored observations (q1env_policy_forward_rows); trajectories are
 * bit-identical to that loop's.  Every workgroup is self-contained (one per CU): four POLICY waves - one per SIMD, the policy
 * network's weights staged ONCE into LDS, 32-env tiles on the matrix cores, the action sampled in the wave that computed its
 * logits - and two or four ENV waves (64 envs each, state in registers); observations and actions change hands through LDS
 * (data, workgroup-scope release, one tag word per env wave / tile).  Every wait is bounded by timeout_s and reported in
 * status uint32[5] as for q1env_step_persistent_* ([0..2] env side, [3..4] policy side; written on failure only).
 * Tick-major trajectory: pi->out = logits float[T][N][out_dim] (or NULL), keys uint8[T][N], mouse float[T][N] (NULL without a
 * mouse), logp float[T][N], obs float[T + 1][N][6] (row 0 = input: the current observations; rows 1..T written), reward
 * float[T][N], done uint8[T][N]; zero_start uint8[N], ep_return double[N], partials double[ceil(N/64)][4] as q1env_sample_step /
 * q1env_episode_stats.  RNG counter of tick t = counter_offset + (*counter_dev if given, else the handle's tick count) + t.
 * Limits: out_dim <= 24 (a discrete-mouse head with up to 24 outputs, e.g. 4 keys + discrete_yaw_steps <= 7; pi->out is then
 * required - the categorical part of the sampling reads the row it has just written); anything else is refused with
 * Q1ENV_ERR_INVALID_ARG - use the per-tick calls.  A workgroup serves 128 envs at one tile per policy wave and tick (every batch
 * of a discrete-mouse head, and batches up to 128 x the number of CUs: 32 768 envs on an MI355X) or 256 at two; the workgroups share
 * nothing, so a batch of more workgroups than the device has CUs (65 536 envs) runs as successive sets of workgroups, each
 * playing its envs' whole horizon (round 4; before, such a batch was refused). */
typedef struct q1env_resident_args {
    int ticks;
    int deterministic;
    const q1env_mlp* pi;
    uint64_t seed;
    const uint64_t* counter_dev;
    uint64_t counter_offset;
    uint8_t* keys_dev;
    float* mouse_dev;
    float* logp_dev;
    float* obs_dev;
    float* reward_dev;
    uint8_t* done_dev;
    uint8_t* zero_start_dev;
    double* ep_return_dev;
    double* partials_dev;
    uint32_t* status_dev;
    double timeout_s;
} q1env_resident_args;
int q1env_sample_resident(q1env_t* env, const q1env_resident_args* args);

/* ---- persistent tick server (experimental) -----------------------------------------------------
 * One resident grid serves `ticks` ticks without a kernel boundary per tick: the env state stays in registers and tick t's action is
 * handed over by a producer that runs CONCURRENTLY on another stream.  Bit-identical to `ticks` q1env_step_autoreset (auto_reset
 * != 0; Philox counter as there) or q1env_step calls with the packed action layout.  Everything that crosses is an 8-byte
 * data-tagged granule, written by ONE agent-scope (sc1) store and polled with sc1 loads (MI355X_MICROARCH.md, persistent-kernel
 * price list) - no flags, fences or drains, one hop per direction:
 *   mailbox[i]    = (tag << 40) | (key bits << 32) | float32 bits of the mouse action          producer -> server, uint64[N]
 *   G[k][i], k = 0..5 = (tag << 40) | float32 bits of observation column k                   server -> consumer
 *   G[6][i] = (tag << 40) | (zero_start << 33) | (done << 32) | float32 bits of reward;  G[7][i] = tag << 40 (padding)
 *   results = uint64[4][N][2]: pair q of env i = {G[2q][i], G[2q+1][i]}, written as ONE 16-byte sc1 store (an sc1 store is one
 *   fabric write per lane whatever its width); every 8-byte half carries its own tag and can be read and validated alone.
 *   tag of tick t (0-based) of the launch = (tag0 + t) mod (2^24 - 1) + 1, i.e. 1 .. 0xFFFFFF and never 0: zero the mailbox before the
 *   first launch; continue a run with tag0' = (tag0 + ticks) mod (2^24 - 1).
 * obs_final (optional, float[N][6]): the last served tick's observation rows as plain stores at the end of the launch.
 * status uint32[5], ACCUMULATED by the kernels and written ONLY on failure (all zero = every wave served / handed over every tick;
 * zero it before a launch to read that launch alone): [0] += server waves that did not serve every tick, [1] |= 1 when a server
 * wave timed out waiting for an action, [2] = max ticks a server wave left unserved, [3] |= 1 when a driver wave timed out,
 * [4] = max actions a driver wave did not hand over.  Every wait is bounded by timeout_s (of no progress): a missing producer
 * ends the launch with status[1] set and the state of the last completed tick stored - it never hangs the device.  The server's
 * grid must be resident at once AND leave room for its producer's waves: a wave serves 1, 2 or 4 sub-batches of 64 envs (the
 * smallest count whose grid fits the kernel's occupancy; external producers never see it - mailbox and results are indexed by
 * env); batches beyond that (262 144 envs on an MI355X) are refused with Q1ENV_ERR_INVALID_ARG.
 * _start launches the server on the handle's stream (asynchronous; wait with q1env_sync).  _drive launches the reference
 * producer on `producer_stream` (a hipStream_t other than the handle's): a DEPENDENT driver - what a policy is to the env - that
 * hands tick t+1's action (from tick-major packed arrays keys uint8[T][N], mouse float[T][N]) over only after all result
 * granules of tick t of the same env arrived, and adds the rewards / first observation column it received to checksum
 * double[2][N] (optional). */
int q1env_step_persistent_start(q1env_t* env, int ticks, uint32_t tag0, const uint64_t* mailbox_dev, uint64_t* results_dev,
                                float* obs_final_dev, uint64_t seed, int auto_reset, uint32_t* status_dev, double timeout_s);
int q1env_step_persistent_drive(q1env_t* env, void* producer_stream, int ticks, uint32_t tag0, const uint8_t* keys_dev,
                                const float* mouse_dev, uint64_t* mailbox_dev, const uint64_t* results_dev,
                                double* checksum_dev, uint32_t* status_dev, double timeout_s);
/* An EXTERNAL producer (a policy) talks to the server with two ordinary launches on its own stream: _publish hands tick `tick`
 * (0-based within the launch that was started with the same tag0) of packed actions over; _collect waits - bounded by timeout_s,
 * reporting a timeout in status[3] - for that tick's result granules and unpacks them into plain arrays (obs float[N][6]; reward
 * float[N], done / zero_start uint8[N] optional) for the kernels that follow on that stream.  A sampler iteration is then
 * collect(t-1) -> policy forward -> action sampling -> publish(t), with no launch at all on the env side. */
int q1env_step_persistent_publish(q1env_t* env, void* producer_stream, uint32_t tag0, uint32_t tick, const uint8_t* keys_dev,
                                  const float* mouse_dev, uint64_t* mailbox_dev);
int q1env_step_persistent_collect(q1env_t* env, void* producer_stream, uint32_t tag0, uint32_t tick, const uint64_t* results_dev,
                                  float* obs_dev, float* reward_dev, uint8_t* done_dev, uint8_t* zero_start_dev, uint32_t* status_dev,
                                  double timeout_s);
/* The server and the reference dependent producer as ONE dispatch on the handle's stream: a workgroup = one server wave + one driver
 * wave serving 1, 2 or 3 sub-batches of 64 envs (the smallest count whose grid is resident: up to 131 072 / 262 144 / 294 912 envs
 * on an MI355X; more is refused).  Both sides of every hand-off then sit on one CU, so the per-tick hand-offs go through LDS (data,
 * then one tag word per sub-batch carrying the tick count) instead of the granules above - the arrangement the resident sampler has
 * with a real policy - and a server wave with several sub-batches keeps one env state in registers and rotates the others through LDS.
 * Same results as _start + _drive: the state, obs_final, status and checksum; `results` receives the LAST tick's granules (tag of tick
 * ticks - 1), mailbox_dev is not touched.  (Two streams are only concurrent when the runtime maps them to different hardware
 * queues, which HIP does not promise - a producer queued behind the server it feeds can only time out: give an external producer
 * a stream of another PRIORITY, hipStreamCreateWithPriority, or use this entry point.)
 * auto_reset: bit 0 = in-kernel reset of finished episodes; Q1ENV_TIMER_START (4) / Q1ENV_TIMER_STOP (8) may be added to record the
 * handle's timer events right around the launch, as in q1env_step_many. */
int q1env_step_persistent_pair(q1env_t* env, int ticks, uint32_t tag0, const uint8_t* keys_dev, const float* mouse_dev,
                               uint64_t* mailbox_dev, uint64_t* results_dev, float* obs_final_dev, uint64_t seed, int auto_reset,
                               double* checksum_dev, uint32_t* status_dev, double timeout_s);

/* (ABI v3, diagnostics) Counters of the ASSERTION build of this library (python -m q1physrl_amd.build --check -> libq1env_check.so,
 * compiled with -DQ1_CHECK): every hand-rolled 16-byte sc1 granule-pair store of the tick server is read back and compared with the
 * registers it was issued from.  out4 = {1 if built with Q1_CHECK else 0, pair stores checked, mismatches, 0}; clear != 0 zeroes the
 * device counters.  The product build returns {0, 0, 0, 0}.  Synchronises the handle's stream.  (No reference counterpart.) */
int q1env_debug_counters(q1env_t* env, uint64_t* out4, int clear);

/* ---- measurement ------------------------------------------------------------------------------
 * calibrate_traffic: `launches` launches of a pure copy kernel that reads the SoA state with step's own
 * load pattern and writes it to scratch: exactly 85 B read + 85 B written per env, for calibrating the
 * rocprofv3 FETCH_SIZE / WRITE_SIZE counters on a known byte count (MI355X_MICROARCH.md, HBM section).
 * selftest_division: runs the kernels' exact-division shortcuts (Markstein constant division, shared-reciprocal
 * division, float32 obs columns) against the hardware IEEE division on n random operands plus the constants
 * 180, 90, 100, 200, c0, c1; writes 4 mismatch counts (all must be 0).
 * timer_*: HIP events recorded on the handle's stream. */
int q1env_selftest_division(int device, uint64_t n, uint64_t seed, double c0, double c1, uint64_t* mismatches4);
/* (ABI v3) selftest_trig: the tick's own sin / cos of a yaw (radians = yaw*pi/180 with the exact constant division, then the in-line
 * sincos of csrc/q1env_device.hpp that replaces the device library's on the tick: phys.py:56-66) for n HOST yaw values in degrees;
 * writes sin and cos to the host arrays (the caller compares them with its own libm: <= 1 ulp, a few % of values differ) and
 * counts[0] = results that differ from the device library's sincos, counts[1] = the largest such difference in ulps,
 * counts[2] = mismatches of the scaling-free square root (sqrt_normal) against the compiler's sqrt on n random operands in
 * [2^-700, 2^700] (must be 0), counts[3] = 0. */
int q1env_selftest_trig(int device, uint64_t n, const double* yaw_deg, double* sin_out, double* cos_out, uint64_t seed, uint64_t* counts4);
int q1env_calibrate_traffic(q1env_t* env, int launches);
/* (ABI v5) visibility check of the completion signal: enqueue, on reader_stream (a hipStream_t other than the handle's), a kernel that
 * waits for the NEXT signalled launch's sequence number and then compares that launch's tick-major outputs - out_dev: obs f32 [T][N][6]
 * at offset 0, reward f32 [T][N] at off_reward, done u8 [T][N] at off_done; N % 8 == 0 - with expect_dev (same layout), newest tick
 * first, using system-scope loads; result_dev = four zeroed 64-bit device words: [0] differing 8-byte words, [1] 1 on timeout, [2] / [3]
 * device wall clock when the signal was seen / when the newest tick had been checked.  tests/test_hip_signal.py,
 * tools/visibility_probe.py.  No reference counterpart. */
int q1env_diag_signal_reader(q1env_t* env, void* reader_stream, const void* out_dev, const void* expect_dev, uint64_t off_reward,
                             uint64_t off_done, int ticks, int workgroups, double timeout_s, uint64_t* result_dev);
int q1env_timer_start(q1env_t* env);
int q1env_timer_stop(q1env_t* env, float* elapsed_ms);   /* = timer_mark + timer_elapsed: synchronises on the stop event */
/* The two halves of timer_stop, for a timed region that ends in ONE synchronisation of the caller's own: mark records the
 * stop event (asynchronous); elapsed waits for it (immediate once the stream has drained) and returns start -> mark. */
int q1env_timer_mark(q1env_t* env);
int q1env_timer_elapsed(q1env_t* env, float* elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* Q1ENV_H */
