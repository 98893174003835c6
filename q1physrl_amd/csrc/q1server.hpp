// q1server.hpp - the resident tick server of libq1env (q1env_step_persistent_start / _drive / _pair, include/q1env.h): device code.
// Included by q1env_server.hip after q1env_device.hpp; uses tick<>, reset_philox, observe<>, load_env / store_env from there.
#pragma once
#include "q1env_device.hpp"

using namespace q1;

// =========================================================================================== persistent tick server
// q1env_step_persistent_*: ONE resident grid serves ticks for as long as actions keep arriving, the env state lives in registers
// between ticks, and tick t's action is handed over by a producer running CONCURRENTLY on another stream - there is no kernel
// boundary per tick (the ~1.8 us dependent-dispatch boundary + the write-back of the tick's dirty state lines that bound
// q1env_step at 65 536 envs).  Hand-off protocol = the data-tagged granule of MI355X_MICROARCH.md (persistent-kernel price list,
// "handoff-1to1"): every word that crosses is a naturally aligned 8-byte {data, tag} written by ONE sc1 (agent-scope,
// write-through) store and polled with sc1 loads, so no separate flag, no fence, no L2 write-back and no store drain is needed in
// either direction, and a consumer has a whole tick's outputs after ONE hop:
//     action granule     mailbox[i]       = (tag << 40) | (keys << 32) | float_bits(mouse)                   producer -> server
//     result granules    G[k][i], k = 0..5 = (tag << 40) | float_bits(obs[k])                                  server -> consumer
//                        G[6][i]          = (tag << 40) | (zero_start << 33) | (done << 32) | float_bits(reward)
//                        G[7][i]          = (tag << 40)                                              (padding)
//     stored as PAIRS: results = uint64[4][N][2], pair q of env i = {G[2q][i], G[2q+1][i]} (16 bytes, one sc1 access).
// tag = (tag0 + t) mod (2^24 - 1) + 1 for tick t of the launch: 1 .. 0xFFFFFF, never 0 - a zeroed mailbox holds no valid action -
// and consecutive ticks never share a tag across the wrap-around.
// Every wait is bounded: a lane that sees no new tag for `timeout_ticks` of the 100 MHz wall clock gives up, the wave stores its
// state as of the last completed tick and reports status[1] != 0 - a missing or stalled producer ends the launch, not the GPU.
// Bit-identical to `ticks` q1env_step_autoreset / q1env_step calls with the packed action layout.
//
// These granules serve the TWO-STREAM form (q1env_step_persistent_start + _drive / _publish / _collect: the producer is another
// dispatch, possibly an external one).  q1env_step_persistent_pair puts a server wave and its driver wave into ONE workgroup and
// hands over through LDS instead (tick_pair_lds_kernel, end of this file).  In between the round tried granules through the shared
// L2 of an XCD for wave pairs that had verified - by exchanging HW_REG_XCC_ID - that they sit on one XCD (plain stores, L1-bypassing
// loads: round trip 0.79 us against 1.78 us agent-scope, tools/ubench_handoff.hip): 2.7 -> 1.8 us per tick at 65 536 envs, then
// superseded by the LDS form (1.3 us) and removed - two separate dispatches were never observed to land pairwise on one XCD.
constexpr int RESULT_GRANULES = 7;

// Poll pacing, in units of s_sleep(1) (64 clocks): `first_*` before the first poll of a tick - the other side needs at least a hop
// plus its own work before anything new can be there, and thousands of waves polling early only load the fabric the hand-offs
// travel through - and `between` after every failed poll.
struct Backoff { int first_server, first_driver, between; };

__device__ __forceinline__ void nap(int units) {
    for (int k = 0; k < units; ++k) __builtin_amdgcn_s_sleep(1);
}

__device__ __forceinline__ uint64_t tick_tag(uint32_t tag0, uint32_t t) {          // 1 .. 0xFFFFFF
    return (uint64_t)(((uint64_t)tag0 + (uint64_t)t) % 0xFFFFFFull) + 1ull;
}

__device__ __forceinline__ uint64_t granule_load(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Two granules that are neighbours in memory leave / arrive as ONE 16-byte sc1 access: an sc1 store is one fabric write per lane
// whatever its width (MI355X_MICROARCH.md: dwordx2 costs 2.7x the dwordx4 time per byte), so the seven result granules of a tick
// cost four writes instead of seven.  Each 8-byte half still carries its own tag, so a consumer validates every granule by itself
// and a 16-byte access torn at the 8-byte boundary (never observed on gfx950) would be detected, not consumed.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifdef Q1_CHECK
// Assertion build (python -m q1physrl_amd.build --check -> libq1env_check.so; tools/soak_check.py): every hand-rolled 16-byte sc1
// store is read back (sc1: from L2, past this CU's L1) and compared with the operands it was issued from.  Nobody else writes a
// result pair between two ticks of its own lane, so a mismatch can only be this store (a data hazard behind the inline assembly,
// as round 2 once had) - counted, never fatal.  q1env_debug_counters reads the two words.
__device__ unsigned long long q1_check_pair_stores = 0ull, q1_check_pair_mismatches = 0ull;
#endif

__device__ __forceinline__ void granule_pair_store(uint64_t* p, uint64_t a, uint64_t b) {
    const u32x4 v = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    // (the s_nop covers the "VALU overwrites the data registers of a > 64-bit VMEM store" hazard: the compiler's hazard recognizer
    // does not look inside inline assembly, and without it lanes 12-15 of every 16 stored the NEXT pair's first word)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
#ifdef Q1_CHECK
    u32x4 r;
    asm volatile("s_waitcnt vmcnt(0)\n\tglobal_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    const bool bad = r[0] != (uint32_t)a || r[1] != (uint32_t)(a >> 32) || r[2] != (uint32_t)b || r[3] != (uint32_t)(b >> 32);
    const unsigned long long act = __ballot(true), wrong = __ballot(bad);
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (unsigned)__builtin_ctzll(act)) {   // first active lane
        atomicAdd(&q1_check_pair_stores, (unsigned long long)__builtin_popcountll(act));
        if (wrong) atomicAdd(&q1_check_pair_mismatches, (unsigned long long)__builtin_popcountll(wrong));
    }
#endif
}

// four pairs (eight granules) with all four loads in flight together
__device__ __forceinline__ void granule_pairs_load4(const uint64_t* p0, const uint64_t* p1, const uint64_t* p2, const uint64_t* p3,
                                                    uint64_t (&g)[8]) {
    u32x4 a, b, c, d;
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
    g[0] = (uint64_t)a[0] | ((uint64_t)a[1] << 32); g[1] = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
    g[2] = (uint64_t)b[0] | ((uint64_t)b[1] << 32); g[3] = (uint64_t)b[2] | ((uint64_t)b[3] << 32);
    g[4] = (uint64_t)c[0] | ((uint64_t)c[1] << 32); g[5] = (uint64_t)c[2] | ((uint64_t)c[3] << 32);
    g[6] = (uint64_t)d[0] | ((uint64_t)d[1] << 32); g[7] = (uint64_t)d[2] | ((uint64_t)d[3] << 32);
}

// address of granule pair q (granules 2q, 2q+1) of env i: pair-index-major, 16 bytes per env
__device__ __forceinline__ uint64_t* pair_ptr(const uint64_t* results, uint32_t n, uint32_t q, uint32_t i) {
    return const_cast<uint64_t*>(results) + ((size_t)q * n + i) * 2u;
}

// One bounded wait of a wave on one granule set: `probe(poll number)` loads and validates the lane's granules (returns true when
// its tag has arrived); the load's own latency paces the loop, and the 100 MHz clock is only consulted every 256 failed polls (no
// s_memrealtime on the path of a tick that is served promptly) - the timeout counts from the first such look.
template <typename Probe>
__device__ __forceinline__ bool wait_for(bool live, uint64_t timeout_ticks, const Backoff& bo, Probe probe) {
    bool ok = !live;
    uint32_t polls = 0;
    uint64_t t_wait = 0;
    for (;;) {
        if (!ok) ok = probe(polls);
        if (__all(ok)) return true;
        nap(bo.between);
        if ((++polls & 255u) == 0u) {
            const uint64_t now = wall_clock64();
            if (t_wait == 0) t_wait = now;
            else if (now - t_wait > timeout_ticks) return false;
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

// E = envs per lane.  A wave serves E sub-batches of 64 consecutive envs (env = (block * E + e) * 64 + lane), each an independent
// hand-off stream served in order e = 0 .. E-1 within a tick: while the server computes sub-batch e, the producer's hand-off for
// e + 1 is already in flight - so a batch that would not be resident at one env per lane (more than ~100 k envs next to a
// producer) still runs as ONE resident grid, at E x the arithmetic per wave and the same two hops per tick.
template <bool SPEC, int E>
__device__ __forceinline__ void tick_server_body(const Params& p, const StatePtrs& s, uint32_t block, int ticks, uint32_t tag0,
                                                 const uint64_t* mailbox, uint64_t* results, float* obs_final, uint64_t seed,
                                                 uint64_t counter0, int auto_reset, uint32_t* status, uint64_t timeout_ticks,
                                                 Backoff bo) {
    const uint32_t lane = threadIdx.x, n = (uint32_t)p.n;
    uint32_t idx[E];
    bool live[E];
    Env env[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        idx[e] = (block * (uint32_t)E + (uint32_t)e) * 64u + lane;
        live[e] = idx[e] < n;
        env[e] = Env{};
        if (live[e]) load_env(s, n, idx[e], env[e]);
    }
    int completed = 0;
    bool timed_out = false;
    const TickConsts tc = tick_consts();
    for (int t = 0; t < ticks && !timed_out; ++t) {
        const uint64_t tag = tick_tag(tag0, (uint32_t)t);
        if (t > 0) nap(bo.first_server);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t i = idx[e];
            uint64_t g = 0;
            // every lane polls its own granule: one contiguous 512-B sc1 read per wave
            if (!wait_for(live[e], timeout_ticks, bo, [&](uint32_t) { g = granule_load(mailbox + i); return (g >> 40) == tag; })) {
                timed_out = true;
                break;
            }
            if (live[e]) {
                const uint32_t keys = (uint32_t)(g >> 32) & ((1u << cfg_num_keys<SPEC>(p)) - 1u);
                const double yaw_act = cfg_yaw_mode<SPEC>(p) ? (double)__uint_as_float((uint32_t)g) : 0.0;
                TickOut<float> o;
                tick<float, SPEC>(p, tc, env[e], keys, yaw_act, o);
                const bool zs = (env[e].flags & FLAG_ZERO_START) != 0;                  // of the episode the step belonged to
                if (auto_reset && o.done) {
                    reset_philox(p, env[e], seed, (uint64_t)p.env_index_base + (uint64_t)i, counter0 + (uint64_t)t + 1);
                    observe<float>(p, env[e], o.obs);
                }
                const uint64_t hi = tag << 40;
                const uint64_t last = hi | ((uint64_t)(zs ? 1u : 0u) << 33) | ((uint64_t)(o.done ? 1u : 0u) << 32) | (uint64_t)__float_as_uint(o.reward);
#pragma unroll
                for (uint32_t q = 0; q < 3u; ++q)
                    granule_pair_store(pair_ptr(results, n, q, i), hi | (uint64_t)__float_as_uint(o.obs[2 * q]),
                                       hi | (uint64_t)__float_as_uint(o.obs[2 * q + 1]));
                granule_pair_store(pair_ptr(results, n, 3u, i), last, hi);      // granule 7 is padding: tag only
            }
        }
        if (!timed_out) completed = t + 1;
    }
    // (a wave that timed out in the middle of a tick has served that tick for its first sub-batches only: their state is one tick
    // ahead of the others' - reported through status, like every incomplete launch)
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (live[e]) {
            store_env(s, n, idx[e], env[e]);
            if (obs_final && completed > 0) {                    // plain row of the last served tick (a tick ends with observe() of the state it leaves)
                float o[6];
                observe<float>(p, env[e], o);
                write_obs<float>(obs_final, (size_t)idx[e], o);
            }
        }
    if (lane == 0 && completed != ticks) {                       // nothing is written on the success path: thousands of waves ending
        atomicAdd(&status[0], 1u);                               // together would serialise ~12 ns per atomic on these five words
        if (timed_out) atomicOr(&status[1], 1u);
        atomicMax(&status[2], (uint32_t)(ticks - completed));    // ticks the slowest wave left unserved
    }
}

// The reference driver of the tick server: a DEPENDENT producer, i.e. what a policy is to the env - it hands tick t+1's action
// over only after ALL result granules of tick t of the same env have arrived.  Actions come from a resident tick-major packed
// episode (keys uint8[T][N], mouse float[T][N]); checksum (optional, double[2][N]) accumulates the rewards and the first
// observation column it received, so the data really makes the round trip.
// A driver wave feeds the server wave of the same block index (ED sub-batches of 64 envs).  The result polls of up to four
// sub-batches are in flight TOGETHER (a poll is a fabric round trip even when the granule is there): each round requests the
// pending sub-batches of the group, then consumes those that have arrived and hands their next action over at once.
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t granule_rsrc(const uint64_t* base, uint32_t bytes) {
    // (wave-uniform by construction: kernel arguments only)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(base), 0, bytes, 0x00020000);
}
// 16-byte L1-bypassing (sc1) load the compiler can see: several stay in flight, its own s_waitcnt before the first use
__device__ __forceinline__ u32x4v granule_pair_load_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_offset, 0, 16);
}

template <int ED>
__device__ __forceinline__ void tick_driver_body(int n_, uint32_t dblock, int ticks, uint32_t tag0, const uint8_t* keys, const float* mouse,
                                                 uint64_t* mailbox, const uint64_t* results, double* checksum, uint32_t* status,
                                                 uint64_t timeout_ticks, Backoff bo) {
    constexpr int GROUP = ED < 4 ? ED : 4;
    static_assert(ED % GROUP == 0, "sub-batches are polled in full groups");
    const uint32_t lane = threadIdx.x, n = (uint32_t)n_;
    const __amdgpu_buffer_rsrc_t rsrc = granule_rsrc(results, n * 64u);    // uint64[4][N][2]
    uint32_t idx[ED];
    bool live[ED];
    double acc_r[ED], acc_o[ED];
#pragma unroll
    for (int e = 0; e < ED; ++e) {
        idx[e] = (dblock * (uint32_t)ED + (uint32_t)e) * 64u + lane;
        live[e] = idx[e] < n;
        acc_r[e] = 0.0; acc_o[e] = 0.0;
    }
    bool timed_out = false;
    int handed = 0;
    for (int t = 0; t < ticks && !timed_out; ++t) {
        const uint64_t tag = tick_tag(tag0, (uint32_t)t);
        const uint64_t want = tick_tag(tag0, t > 0 ? (uint32_t)t - 1u : 0u);          // results of tick t-1
        // tick t's actions are fetched before the waits: their latency hides under the server's tick
        uint32_t k[ED];
        float m[ED];
#pragma unroll
        for (int e = 0; e < ED; ++e) {
            k[e] = live[e] ? keys[(size_t)t * n + idx[e]] : 0u;
            m[e] = live[e] ? mouse[(size_t)t * n + idx[e]] : 0.0f;
        }
        if (t > 0) nap(bo.first_driver);
#pragma unroll
        for (int base = 0; base < ED; base += GROUP) {
            uint32_t pending = t > 0 ? (1u << GROUP) - 1u : 0u;           // sub-batches of the group whose tick t-1 results are awaited
            uint32_t polls = 0;
            uint64_t t_wait = 0;
            while (pending != 0u) {
                u32x4v v[GROUP][4];
#pragma unroll
                for (int q = 0; q < GROUP; ++q) {
                    const int e = base + q;
#pragma unroll
                    for (uint32_t w = 0; w < 4u; ++w) v[q][w] = u32x4v{0u, 0u, 0u, 0u};
                    if (((pending >> q) & 1u) != 0u && live[e]) {
#pragma unroll
                        for (uint32_t w = 0; w < 4u; ++w) v[q][w] = granule_pair_load_sc1(rsrc, (w * n + idx[e]) * 16u);
                    }
                }
#pragma unroll
                for (int q = 0; q < GROUP; ++q) {
                    const int e = base + q;
                    if (((pending >> q) & 1u) == 0u) continue;
                    bool ok = true;
#pragma unroll
                    for (uint32_t w = 0; w < 4u; ++w)
                        ok = ok && ((uint64_t)(v[q][w][1] >> 8) == want) && ((uint64_t)(v[q][w][3] >> 8) == want);      // tag = bits 40..63 of each granule
                    if (!__all(!live[e] || ok)) continue;
                    pending &= ~(1u << q);
                    if (live[e]) {
                        acc_r[e] += (double)__uint_as_float(v[q][3][0]);                                                  // granule 6: reward
                        acc_o[e] += (double)__uint_as_float(v[q][0][0]);                                                  // granule 0: obs[0]
                        granule_store(mailbox + idx[e], (tag << 40) | ((uint64_t)(k[e] & 0xFu) << 32) | (uint64_t)__float_as_uint(m[e]));
                    }
                }
                if (pending != 0u) {
                    nap(bo.between);
                    if ((++polls & 255u) == 0u) {                        // (the 100 MHz clock is only looked at every 256 failed rounds)
                        const uint64_t now = wall_clock64();
                        if (t_wait == 0) t_wait = now;
                        else if (now - t_wait > timeout_ticks) { timed_out = true; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
            }
            if (timed_out) break;
        }
        if (timed_out) break;
        if (t == 0) {
#pragma unroll
            for (int e = 0; e < ED; ++e)
                if (live[e]) granule_store(mailbox + idx[e], (tag << 40) | ((uint64_t)(k[e] & 0xFu) << 32) | (uint64_t)__float_as_uint(m[e]));
        }
        handed = t + 1;
    }
#pragma unroll
    for (int e = 0; e < ED; ++e)
        if (live[e] && checksum) { checksum[idx[e]] += acc_r[e]; checksum[(size_t)n + idx[e]] += acc_o[e]; }
    if (lane == 0 && handed != ticks) {
        if (timed_out) atomicOr(&status[3], 1u);
        atomicMax(&status[4], (uint32_t)(ticks - handed));       // actions the slowest wave did not hand over
    }
}

// An EXTERNAL producer's two halves of the protocol, as ordinary launches on the producer's own stream (a torch policy sits
// between them): publish hands one tick's packed actions over, collect waits - bounded - for one tick's result granules and unpacks
// them into plain tensors (obs float[N][6], reward, done, zero_start) for the kernels that follow on that stream.
__global__ void __launch_bounds__(256)
tick_publish_kernel(int n, uint32_t tag0, uint32_t t, const uint8_t* keys, const float* mouse, uint64_t* mailbox) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)n) return;
    const uint64_t tag = tick_tag(tag0, t);
    const float m = mouse ? mouse[i] : 0.0f;
    granule_store(mailbox + i, (tag << 40) | ((uint64_t)(keys[i] & 0xFu) << 32) | (uint64_t)__float_as_uint(m));
}

__global__ void __launch_bounds__(64)
tick_collect_kernel(int n, uint32_t tag0, uint32_t t, const uint64_t* results, float* obs, float* reward, uint8_t* done,
                    uint8_t* zero_start, uint32_t* status, uint64_t timeout_ticks) {
    const uint32_t lane = threadIdx.x, i = blockIdx.x * 64u + lane;
    const bool live = i < (uint32_t)n;
    const uint64_t want = tick_tag(tag0, t);
    uint64_t g[8];
    bool ok = !live, timed_out = false;
    uint32_t polls = 0;
    uint64_t t_wait = 0;
    for (;;) {
        if (!ok) {
            granule_pairs_load4(pair_ptr(results, (uint32_t)n, 0u, i), pair_ptr(results, (uint32_t)n, 1u, i),
                                pair_ptr(results, (uint32_t)n, 2u, i), pair_ptr(results, (uint32_t)n, 3u, i), g);
            ok = true;
#pragma unroll
            for (int q = 0; q < RESULT_GRANULES; ++q) ok = ok && ((g[q] >> 40) == want);
        }
        if (__all(ok)) break;
        if ((++polls & 255u) == 0u) {
            const uint64_t now = wall_clock64();
            if (t_wait == 0) t_wait = now;
            else if (now - t_wait > timeout_ticks) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (timed_out) {
        if (lane == 0) { atomicOr(&status[3], 1u); atomicMax(&status[4], 1u); }
        return;
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < 6; ++k) obs[(size_t)i * 6 + k] = __uint_as_float((uint32_t)g[k]);
        if (reward) reward[i] = __uint_as_float((uint32_t)g[6]);
        if (done) done[i] = (uint8_t)((g[6] >> 32) & 1u);
        if (zero_start) zero_start[i] = (uint8_t)((g[6] >> 33) & 1u);
    }
}

template <bool SPEC, int E>
__global__ void __launch_bounds__(64)
tick_server_kernel(Params p, StatePtrs s, int ticks, uint32_t tag0, const uint64_t* mailbox, uint64_t* results, float* obs_final,
                   uint64_t seed, uint64_t counter0, int auto_reset, uint32_t* status, uint64_t timeout_ticks, Backoff bo) {
    tick_server_body<SPEC, E>(p, s, blockIdx.x, ticks, tag0, mailbox, results, obs_final, seed, counter0, auto_reset, status, timeout_ticks, bo);
}

template <int E>
__global__ void __launch_bounds__(64)
tick_driver_kernel(int n, int ticks, uint32_t tag0, const uint8_t* keys, const float* mouse, uint64_t* mailbox,
                   const uint64_t* results, double* checksum, uint32_t* status, uint64_t timeout_ticks, Backoff bo) {
    tick_driver_body<E>(n, blockIdx.x, ticks, tag0, keys, mouse, mailbox, results, checksum, status, timeout_ticks, bo);
}

// ============================================================================ server + driver as ONE WORKGROUP (hand-offs through LDS)
// q1env_step_persistent_pair's kernel: a workgroup = one SERVER wave + one DRIVER wave (128 threads) serving ES sub-batches of 64
// envs.  Both sides of every hand-off sit on one CU, so it goes through LDS - the data words, a workgroup-scope release, then ONE
// tag word per sub-batch that counts the ticks handed over; the reader spins on the tag (a broadcast ds_read, ~0.1 us per look
// against ~0.45 us for an L2 granule and ~0.9 us agent-scope), acquires, reads.  This is the arrangement the resident sampler has
// (q1resident.hpp: the policy waves sit in the env waves' workgroup); the driver is its stand-in - a DEPENDENT producer that hands
// tick t + 1's action over only after it has read all eight result words of tick t.  With ES > 1 the server keeps ONE env state in
// registers and rotates the sub-batches' states through LDS (11 x 8 bytes per env): one copy of the tick code for any ES, 128
// registers, four waves per SIMD - two server waves per SIMD overlap their float64 chains.  The LAST tick of a launch is also
// stored as agent-scope granules into `results` (and obs_final), exactly as the two-stream form leaves it.
namespace q1pair {
constexpr uint32_t STATE_WORDS = 11;        // uint64 words of one Env: {vx,vy} {vz,flags} px py z yaw trem lk[4]
constexpr size_t lds_bytes(int es) {          // tags | act[ES][64] u64 | res[ES][8][64] u32 | state[ES][11][64] u64 (ES > 1 only)
    return (size_t)es * 64u * (8u + 32u + (es > 1 ? STATE_WORDS * 8u : 0u)) + 64u;
}
__device__ __forceinline__ void env_to_lds(uint64_t* w, uint32_t lane, const Env& e) {          // w = this sub-batch's [11][64] words
    w[0 * 64 + lane] = (uint64_t)__float_as_uint(e.vx) | ((uint64_t)__float_as_uint(e.vy) << 32);
    w[1 * 64 + lane] = (uint64_t)__float_as_uint(e.vz) | ((uint64_t)e.flags << 32);
    w[2 * 64 + lane] = (uint64_t)__double_as_longlong(e.px); w[3 * 64 + lane] = (uint64_t)__double_as_longlong(e.py);
    w[4 * 64 + lane] = (uint64_t)__double_as_longlong(e.z); w[5 * 64 + lane] = (uint64_t)__double_as_longlong(e.yaw);
    w[6 * 64 + lane] = (uint64_t)__double_as_longlong(e.trem);
#pragma unroll
    for (int k = 0; k < 4; ++k) w[(7 + k) * 64 + lane] = (uint64_t)__double_as_longlong(e.lk[k]);
}
__device__ __forceinline__ void env_from_lds(const uint64_t* w, uint32_t lane, Env& e) {
    const uint64_t a = w[0 * 64 + lane], b = w[1 * 64 + lane];
    e.vx = __uint_as_float((uint32_t)a); e.vy = __uint_as_float((uint32_t)(a >> 32));
    e.vz = __uint_as_float((uint32_t)b); e.flags = (uint32_t)(b >> 32);
    e.px = __longlong_as_double((long long)w[2 * 64 + lane]); e.py = __longlong_as_double((long long)w[3 * 64 + lane]);
    e.z = __longlong_as_double((long long)w[4 * 64 + lane]); e.yaw = __longlong_as_double((long long)w[5 * 64 + lane]);
    e.trem = __longlong_as_double((long long)w[6 * 64 + lane]);
#pragma unroll
    for (int k = 0; k < 4; ++k) e.lk[k] = __longlong_as_double((long long)w[(7 + k) * 64 + lane]);
}
__device__ __forceinline__ uint32_t tag_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void tag_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// spin (bounded) until *tag == want; wave-uniform
__device__ __forceinline__ bool wait_tag(const uint32_t* tag, uint32_t want, uint64_t timeout_ticks) {
    uint32_t polls = 0;
    uint64_t t_wait = 0;
    while (tag_load(tag) != want) {
        if ((++polls & 1023u) == 0u) {
            const uint64_t now = wall_clock64();
            if (t_wait == 0) t_wait = now;
            else if (now - t_wait > timeout_ticks) return false;
        }
    }
    asm volatile("" ::: "memory");                                // (LDS only: a wave's DS operations execute in order; no cache to invalidate)
    return true;
}
// publish: everything this wave wrote to LDS before is in LDS before the tag is.  NOT a workgroup-scope release fence: that one also
// waits for the wave's outstanding GLOBAL loads (vmcnt(0)) - the driver's prefetched actions, 1 - 2 us of HBM latency per tick.
__device__ __forceinline__ void publish_tag(uint32_t* tag, uint32_t v, bool lane0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane0) tag_store(tag, v);
}
}  // namespace q1pair

template <bool SPEC, int ES>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4)))
tick_pair_lds_kernel(Params p, StatePtrs s, int ticks, uint32_t tag0, uint64_t* results, float* obs_final, uint64_t seed, uint64_t counter0,
                     int auto_reset, const uint8_t* keys, const float* mouse, double* checksum, uint32_t* status, uint64_t timeout_ticks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x & 63u, n = (uint32_t)p.n;
    const bool is_server = threadIdx.x < 64u;
    // LDS map: tags (act_tag[ES], res_tag[ES]) | act[ES][64] u64 | res[ES][8][64] u32 | state[ES][11][64] u64 (ES > 1 only)
    uint32_t* act_tag = reinterpret_cast<uint32_t*>(lds);
    uint32_t* res_tag = act_tag + ES;
    uint64_t* act = reinterpret_cast<uint64_t*>(lds + 64);
    uint32_t* res = reinterpret_cast<uint32_t*>(lds + 64 + (size_t)ES * 64u * 8u);
    uint64_t* state = reinterpret_cast<uint64_t*>(lds + 64 + (size_t)ES * 64u * 40u);
    if (threadIdx.x < 2u * ES) act_tag[threadIdx.x] = 0u;
    __syncthreads();                                                      // (the only barrier)
    const uint32_t base = blockIdx.x * (64u * (uint32_t)ES) + lane;     // env of sub-batch e: base + 64 e

    if (is_server) {
        Env env{};
        if (ES == 1) {
            if (base < n) load_env(s, n, base, env);
        } else {
#pragma unroll 1
            for (int e = 0; e < ES; ++e) {
                Env tmp{};
                if (base + 64u * e < n) load_env(s, n, base + 64u * e, tmp);
                q1pair::env_to_lds(state + (size_t)e * q1pair::STATE_WORDS * 64u, lane, tmp);
            }
        }
        int completed = 0;
        bool timed_out = false;
        const TickConsts tc = tick_consts();
        for (int t = 0; t < ticks && !timed_out; ++t) {
            const uint64_t tag = tick_tag(tag0, (uint32_t)t);
#pragma unroll 1
            for (int e = 0; e < ES; ++e) {
                const uint32_t i = base + 64u * (uint32_t)e;
                const bool live = i < n;
                if (!q1pair::wait_tag(act_tag + e, (uint32_t)t + 1u, timeout_ticks)) { timed_out = true; break; }
                const uint64_t g = act[e * 64 + lane];
                if (ES > 1) q1pair::env_from_lds(state + (size_t)e * q1pair::STATE_WORDS * 64u, lane, env);
                TickOut<float> o;
                o.reward = 0.0f; o.done = false;
#pragma unroll
                for (int k = 0; k < 6; ++k) o.obs[k] = 0.0f;
                bool zs = false;
                if (live) {
                    const uint32_t kb = (uint32_t)(g >> 32) & ((1u << cfg_num_keys<SPEC>(p)) - 1u);
                    const double yaw_act = cfg_yaw_mode<SPEC>(p) ? (double)__uint_as_float((uint32_t)g) : 0.0;
                    tick<float, SPEC>(p, tc, env, kb, yaw_act, o);
                    zs = (env.flags & FLAG_ZERO_START) != 0;                       // of the episode the step belonged to
                    if (auto_reset && o.done) {
                        // (the counter passes through an empty asm INSIDE the branch: the reset's arithmetic - Philox rounds, sincos -
                        // is pure and, with one sub-batch, invariant in the e loop, and the compiler otherwise hoists ALL of it in
                        // front of the loop, i.e. executes it on every tick: 1.1 us per tick, measured)
                        uint64_t ctr = counter0 + (uint64_t)t + 1;
                        asm volatile("" : "+v"(ctr));
                        reset_philox(p, env, seed, (uint64_t)p.env_index_base + (uint64_t)i, ctr);
                        observe<float>(p, env, o.obs);
                    }
                }
                const uint32_t fl = (zs ? 2u : 0u) | (o.done ? 1u : 0u);
                uint32_t* r = res + (size_t)e * 8u * 64u;
#pragma unroll
                for (int k = 0; k < 6; ++k) r[k * 64 + lane] = __float_as_uint(o.obs[k]);
                r[6 * 64 + lane] = __float_as_uint(o.reward);
                r[7 * 64 + lane] = fl;
                if (ES > 1) q1pair::env_to_lds(state + (size_t)e * q1pair::STATE_WORDS * 64u, lane, env);
                q1pair::publish_tag(res_tag + e, (uint32_t)t + 1u, lane == 0);
                if (t == ticks - 1 && live) {
                    // what a launch leaves behind for its caller: the last tick as agent-scope granules, and its observation row
                    const uint64_t hi = tag << 40;
#pragma unroll
                    for (uint32_t q = 0; q < 3u; ++q)
                        granule_pair_store(pair_ptr(results, n, q, i), hi | (uint64_t)__float_as_uint(o.obs[2 * q]),
                                           hi | (uint64_t)__float_as_uint(o.obs[2 * q + 1]));
                    granule_pair_store(pair_ptr(results, n, 3u, i), hi | ((uint64_t)fl << 32) | (uint64_t)__float_as_uint(o.reward), hi);
                    if (obs_final) write_obs<float>(obs_final, (size_t)i, o.obs);
                }
            }
            if (!timed_out) completed = t + 1;
        }
        // (a wave that timed out in the middle of a tick has served that tick for its first sub-batches only - reported through status)
        if (ES == 1) {
            if (base < n) store_env(s, n, base, env);
        } else {
#pragma unroll 1
            for (int e = 0; e < ES; ++e) {
                q1pair::env_from_lds(state + (size_t)e * q1pair::STATE_WORDS * 64u, lane, env);
                if (base + 64u * e < n) store_env(s, n, base + 64u * e, env);
            }
        }
        if (lane == 0 && completed != ticks) {
            atomicAdd(&status[0], 1u);
            if (timed_out) atomicOr(&status[1], 1u);
            atomicMax(&status[2], (uint32_t)(ticks - completed));
        }
    } else {
        // the dependent reference producer: tick t + 1's action only after all eight result words of tick t were read
        double acc_r[ES], acc_o[ES];
        uint32_t k_cur[ES];                                      // the packed actions of the tick about to be handed over ...
        float m_cur[ES];
#pragma unroll
        for (int e = 0; e < ES; ++e) {
            const uint32_t i = base + 64u * (uint32_t)e;
            acc_r[e] = 0.0; acc_o[e] = 0.0;
            k_cur[e] = i < n ? keys[i] : 0u;
            m_cur[e] = i < n ? mouse[i] : 0.0f;
        }
        int handed = 0;
        bool timed_out = false;
        for (int t = 0; t < ticks && !timed_out; ++t) {
            // ... and the NEXT tick's are requested a whole tick ahead: an HBM load (1 - 2 us) is longer than the tick it would otherwise
            // have to hide under
            uint32_t k_nxt[ES];
            float m_nxt[ES];
#pragma unroll
            for (int e = 0; e < ES; ++e) {
                const uint32_t i = base + 64u * (uint32_t)e;
                const bool more = t + 1 < ticks && i < n;
                k_nxt[e] = more ? keys[(size_t)(t + 1) * n + i] : 0u;
                m_nxt[e] = more ? mouse[(size_t)(t + 1) * n + i] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < ES; ++e) {
                const uint32_t i = base + 64u * (uint32_t)e;
                const bool live = i < n;
                if (t > 0) {
                    if (!q1pair::wait_tag(res_tag + e, (uint32_t)t, timeout_ticks)) { timed_out = true; break; }
                    const uint32_t* r = res + (size_t)e * 8u * 64u;
                    uint32_t w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) w[q] = r[q * 64 + lane];
                    uint32_t sink = 0;                                   // every word is read (a policy would consume them all)
#pragma unroll
                    for (int q = 1; q < 6; ++q) sink |= w[q];
                    if (live) {
                        acc_r[e] += (double)__uint_as_float(w[6]);
                        acc_o[e] += (double)__uint_as_float(w[0]);
                    }
                    asm volatile("" ::"v"(sink), "v"(w[7]));
                }
                act[e * 64 + lane] = ((uint64_t)(k_cur[e] & 0xFu) << 32) | (uint64_t)__float_as_uint(m_cur[e]);
                q1pair::publish_tag(act_tag + e, (uint32_t)t + 1u, lane == 0);
            }
#pragma unroll
            for (int e = 0; e < ES; ++e) { k_cur[e] = k_nxt[e]; m_cur[e] = m_nxt[e]; }
            if (!timed_out) handed = t + 1;
        }
#pragma unroll
        for (int e = 0; e < ES; ++e) {
            const uint32_t i = base + 64u * (uint32_t)e;
            if (i < n && checksum) { checksum[i] += acc_r[e]; checksum[(size_t)n + i] += acc_o[e]; }
        }
        if (lane == 0 && handed != ticks) {
            if (timed_out) atomicOr(&status[3], 1u);
            atomicMax(&status[4], (uint32_t)(ticks - handed));
        }
    }
}
