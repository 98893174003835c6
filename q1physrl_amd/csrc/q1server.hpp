// q1server.hpp - the resident tick server of libq1env (q1env_step_persistent_start / _drive / _pair, include/q1env.h): device code.
// Included by q1env.hip after q1env_device.hpp; uses tick<>, reset_philox, observe<>, load_env / store_env from there.
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
// XCD-local fast path (the library's own resident driver only).  An sc1 store DROPS its line from the writer's L2 and an sc1 load
// of it is served over the fabric - ~0.9 us per hop wherever the two waves sit.  A PLAIN store stays in the XCD's L2, where an
// L1-bypassing (sc1) load of ANOTHER CU OF THE SAME XCD finds it in ~0.4 us (tools/ubench_handoff.hip: round trip 1.78 -> 0.79 us);
// a reader on another XCD never sees it.  So placement is not assumed but EXCHANGED: every wave reads HW_REG_XCC_ID and puts it
// into its granules (action bits 36..39, result granule 7 bits 0..3: 8 | xcc; 0 = "unknown", what an external producer writes).
// Tick 0 of a launch travels agent-scope (the driver stores both copies; it polls both for the results).  From then on a side
// whose partner is verified on its own XCD stores the XCD-local copy ONLY (buffers owned by the handle, never seen by an
// external producer) and polls it, looking at the agent-scope copy every eighth poll; every other pair keeps the sc1 protocol.
// Nothing is assumed about block -> XCD placement (the pair grid is merely padded so that block b and block B + b meet on
// one XCD when the dispatcher goes round-robin): a pair on two XCDs is slower, not wrong.  The last tick of a launch is also
// stored agent-scope, so `results` always holds it.
constexpr int RESULT_GRANULES = 7;
constexpr uint32_t PEER_VALID = 8u;            // granule bit: "the low three bits are my XCC id"
constexpr uint32_t NEAR_POLL_PERIOD = 8u;      // near-first polling: polls 0..6 of every 8 read the XCD-local copy, poll 7 the agent-scope copy

struct NearBufs { uint64_t* mailbox; uint64_t* results; };      // XCD-local copies (uint64[N], uint64[4][N][2]); null = sc1 protocol only

__device__ __forceinline__ uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u; }      // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ bool peer_is_near(uint32_t bits, uint32_t my_xcc) { return (bits & PEER_VALID) != 0u && (bits & 7u) == my_xcc; }

// Poll pacing, in units of s_sleep(1) (64 clocks): `first_*` before the first poll of a tick - the other side needs at least a hop
// plus its own work before anything new can be there, and thousands of waves polling early only load the fabric the hand-offs
// travel through - and `between` after every failed poll.
struct Backoff { int first_server, first_driver, between, diag; };     // diag != 0: driver waves count their XCD-local sub-batches into status[5..6] (status must then have 8 words)

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

__device__ __forceinline__ void granule_pair_store(uint64_t* p, uint64_t a, uint64_t b) {
    const u32x4 v = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    // (the s_nop covers the "VALU overwrites the data registers of a > 64-bit VMEM store" hazard: the compiler's hazard recognizer
    // does not look inside inline assembly, and without it lanes 12-15 of every 16 stored the NEXT pair's first word)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}

// the XCD-local flavours: plain stores (the line stays in this XCD's L2)
__device__ __forceinline__ void granule_store_near(uint64_t* p, uint64_t v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void granule_pair_store_near(uint64_t* p, uint64_t a, uint64_t b) {
    const u32x4 v = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
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
                                                 const uint64_t* mailbox, uint64_t* results, NearBufs near, float* obs_final,
                                                 uint64_t seed, uint64_t counter0, int auto_reset, uint32_t* status,
                                                 uint64_t timeout_ticks, Backoff bo) {
    const uint32_t lane = threadIdx.x, n = (uint32_t)p.n;
    const uint32_t my_xcc = xcc_id();
    uint32_t idx[E];
    bool live[E];
    bool near_peer[E];                   // the producer of this sub-batch said (last tick) that it sits on this XCD
    Env env[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        idx[e] = (block * (uint32_t)E + (uint32_t)e) * 64u + lane;
        live[e] = idx[e] < n;
        near_peer[e] = false;
        env[e] = Env{};
        if (live[e]) load_env(s, n, idx[e], env[e]);
    }
    int completed = 0;
    bool timed_out = false;
    for (int t = 0; t < ticks && !timed_out; ++t) {
        const uint64_t tag = tick_tag(tag0, (uint32_t)t);
        if (t > 0) nap(bo.first_server);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t i = idx[e];
            uint64_t g = 0;
            // every lane polls its own granule: one contiguous 512-B sc1 read per wave (of the XCD-local copy while the producer is near)
            const bool near_first = near_peer[e];
            if (!wait_for(live[e], timeout_ticks, bo, [&](uint32_t polls) {
                    const bool far = !near_first || (polls % NEAR_POLL_PERIOD) == NEAR_POLL_PERIOD - 1u;
                    g = granule_load((far ? mailbox : near.mailbox) + i);
                    return (g >> 40) == tag;
                })) {
                timed_out = true;
                break;
            }
            // where this tick's results go is decided by the action granule itself (wave-uniform: one producer wave per sub-batch)
            const bool peer_near = near.results != nullptr && __all(!live[e] || peer_is_near((uint32_t)(g >> 36) & 0xFu, my_xcc));
            near_peer[e] = peer_near;
            const bool store_far = !peer_near || t == ticks - 1, store_near = peer_near;
            if (live[e]) {
                const uint32_t keys = (uint32_t)(g >> 32) & ((1u << cfg_num_keys<SPEC>(p)) - 1u);
                const double yaw_act = cfg_yaw_mode<SPEC>(p) ? (double)__uint_as_float((uint32_t)g) : 0.0;
                TickOut<float> o;
                tick<float, SPEC>(p, env[e], keys, yaw_act, o);
                const bool zs = (env[e].flags & FLAG_ZERO_START) != 0;                  // of the episode the step belonged to
                if (auto_reset && o.done) {
                    reset_philox(p, env[e], seed, (uint64_t)p.env_index_base + (uint64_t)i, counter0 + (uint64_t)t + 1);
                    observe<float>(p, env[e], o.obs);
                }
                const uint64_t hi = tag << 40;
                const uint64_t last = hi | ((uint64_t)(zs ? 1u : 0u) << 33) | ((uint64_t)(o.done ? 1u : 0u) << 32) | (uint64_t)__float_as_uint(o.reward);
                const uint64_t pad = hi | (uint64_t)(PEER_VALID | my_xcc);                // granule 7: tag + where the server wave sits
                if (store_near) {
#pragma unroll
                    for (uint32_t q = 0; q < 3u; ++q)
                        granule_pair_store_near(pair_ptr(near.results, n, q, i), hi | (uint64_t)__float_as_uint(o.obs[2 * q]),
                                                hi | (uint64_t)__float_as_uint(o.obs[2 * q + 1]));
                    granule_pair_store_near(pair_ptr(near.results, n, 3u, i), last, pad);
                }
                if (store_far) {
#pragma unroll
                    for (uint32_t q = 0; q < 3u; ++q)
                        granule_pair_store(pair_ptr(results, n, q, i), hi | (uint64_t)__float_as_uint(o.obs[2 * q]),
                                           hi | (uint64_t)__float_as_uint(o.obs[2 * q + 1]));
                    granule_pair_store(pair_ptr(results, n, 3u, i), last, pad);
                }
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
// A driver wave feeds the server wave of the same block index (ED = ES sub-batches of 64 envs; block d and server block d meet on
// one XCD when the dispatcher places blocks round-robin and the server half is a multiple of 8 blocks - speed only, verified per
// pair).  The result polls of up to four sub-batches are in flight TOGETHER (a poll is an L2 round trip of ~0.4 us even when the
// granule is there): each round requests the pending sub-batches of the group, then consumes those that have arrived and hands
// their next action over at once.
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
                                                 uint64_t* mailbox, const uint64_t* results, NearBufs near, double* checksum,
                                                 uint32_t* status, uint64_t timeout_ticks, Backoff bo) {
    constexpr int GROUP = ED < 4 ? ED : 4;
    static_assert(ED % GROUP == 0, "sub-batches are polled in full groups");
    const uint32_t lane = threadIdx.x, n = (uint32_t)n_;
    const uint32_t my_xcc = xcc_id();
    const bool has_near = near.mailbox != nullptr;
    const uint32_t result_bytes = n * 64u;                                 // uint64[4][N][2]
    const __amdgpu_buffer_rsrc_t far_rsrc = granule_rsrc(results, result_bytes);
    const __amdgpu_buffer_rsrc_t near_rsrc = granule_rsrc(has_near ? near.results : results, result_bytes);
    uint32_t idx[ED];
    bool live[ED];
    bool near_peer[ED];                  // the server wave of this sub-batch is known to sit on this XCD (from its last results)
    double acc_r[ED], acc_o[ED];
#pragma unroll
    for (int e = 0; e < ED; ++e) {
        idx[e] = (dblock * (uint32_t)ED + (uint32_t)e) * 64u + lane;
        live[e] = idx[e] < n;
        near_peer[e] = false;
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
                // tick 0's results may be in either copy (the server knows by then where this wave sits, this wave does not know the
                // server's place yet): near-first polling covers both; afterwards only a near server is polled near-first
                const bool far_round = (polls % NEAR_POLL_PERIOD) == NEAR_POLL_PERIOD - 1u;
                u32x4v v[GROUP][4];
#pragma unroll
                for (int q = 0; q < GROUP; ++q) {
                    const int e = base + q;
                    const bool near_first = has_near && (t == 1 || near_peer[e]);
                    const __amdgpu_buffer_rsrc_t r = (!near_first || far_round) ? far_rsrc : near_rsrc;
#pragma unroll
                    for (uint32_t w = 0; w < 4u; ++w) v[q][w] = u32x4v{0u, 0u, 0u, 0u};
                    if (((pending >> q) & 1u) != 0u && live[e]) {
#pragma unroll
                        for (uint32_t w = 0; w < 4u; ++w) v[q][w] = granule_pair_load_sc1(r, (w * n + idx[e]) * 16u);
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
                    near_peer[e] = has_near && __all(!live[e] || peer_is_near(v[q][3][2] & 0xFu, my_xcc));               // granule 7's payload
                    if (live[e]) {
                        acc_r[e] += (double)__uint_as_float(v[q][3][0]);                                                  // granule 6: reward
                        acc_o[e] += (double)__uint_as_float(v[q][0][0]);                                                  // granule 0: obs[0]
                        const uint64_t a = (tag << 40) | ((uint64_t)(has_near ? (PEER_VALID | my_xcc) : 0u) << 36) | ((uint64_t)(k[e] & 0xFu) << 32) |
                                           (uint64_t)__float_as_uint(m[e]);
                        if (near_peer[e]) granule_store_near(near.mailbox + idx[e], a);
                        else granule_store(mailbox + idx[e], a);
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
            // the first tick of a launch travels agent-scope AND XCD-local: neither side knows the other's place yet
#pragma unroll
            for (int e = 0; e < ED; ++e)
                if (live[e]) {
                    const uint64_t a = (tag << 40) | ((uint64_t)(has_near ? (PEER_VALID | my_xcc) : 0u) << 36) | ((uint64_t)(k[e] & 0xFu) << 32) |
                                       (uint64_t)__float_as_uint(m[e]);
                    if (has_near) granule_store_near(near.mailbox + idx[e], a);
                    granule_store(mailbox + idx[e], a);
                }
        }
        handed = t + 1;
    }
#pragma unroll
    for (int e = 0; e < ED; ++e)
        if (live[e] && checksum) { checksum[idx[e]] += acc_r[e]; checksum[(size_t)n + idx[e]] += acc_o[e]; }
    if (bo.diag != 0 && lane == 0) {                             // measurement knob (Q1ENV_SERVER_DIAG): how many sub-batches ended XCD-local
        uint32_t near_count = 0, total = 0;
#pragma unroll
        for (int e = 0; e < ED; ++e) { near_count += near_peer[e] ? 1u : 0u; total += (idx[e] - lane) < n ? 1u : 0u; }
        atomicAdd(&status[5], near_count);
        atomicAdd(&status[6], total);
    }
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
tick_server_kernel(Params p, StatePtrs s, int ticks, uint32_t tag0, const uint64_t* mailbox, uint64_t* results, NearBufs near, float* obs_final,
                   uint64_t seed, uint64_t counter0, int auto_reset, uint32_t* status, uint64_t timeout_ticks, Backoff bo) {
    tick_server_body<SPEC, E>(p, s, blockIdx.x, ticks, tag0, mailbox, results, near, obs_final, seed, counter0, auto_reset, status, timeout_ticks, bo);
}

template <int E>
__global__ void __launch_bounds__(64)
tick_driver_kernel(int n, int ticks, uint32_t tag0, const uint8_t* keys, const float* mouse, uint64_t* mailbox,
                   const uint64_t* results, NearBufs near, double* checksum, uint32_t* status, uint64_t timeout_ticks, Backoff bo) {
    tick_driver_body<E>(n, blockIdx.x, ticks, tag0, keys, mouse, mailbox, results, near, checksum, status, timeout_ticks, bo);
}

// Server and reference driver in ONE dispatch (q1env_step_persistent_pair): blocks [0, B) are the server's waves (ES sub-batches of 64
// envs each), blocks [B, 2B) the driver's.  Two streams are only concurrent when the runtime maps them to different hardware
// queues, which HIP does not promise (a process that has created many streams re-uses queues: the producer then queues BEHIND the
// server it feeds and both sides can only time out).  One grid that fits the device is co-resident by construction - this is what
// the benchmark and most tests use; the two-stream entry points remain for an external producer.  The host pads B to a multiple of
// 8 (blocks whose envs are all beyond n idle through the loop), so that with the round-robin block -> XCD placement the dispatcher
// is observed to use, server block b and its driver block B + b share an XCD and take the XCD-local path (verified per wave pair at
// run time, never assumed).
#define Q1_PAIR_KERNEL_BODY                                                                                                                   \
    if (blockIdx.x < server_blocks)                                                                                                           \
        tick_server_body<SPEC, ES>(p, s, blockIdx.x, ticks, tag0, mailbox, results, near, obs_final, seed, counter0, auto_reset, status,     \
                                   timeout_ticks, bo);                                                                                        \
    else                                                                                                                                      \
        tick_driver_body<ES>(p.n, blockIdx.x - server_blocks, ticks, tag0, keys, mouse, mailbox, results, near, checksum, status,            \
                             timeout_ticks, bo)

template <bool SPEC, int ES>
__global__ void __launch_bounds__(64)
tick_pair_kernel(Params p, StatePtrs s, int ticks, uint32_t tag0, uint64_t* mailbox, uint64_t* results, NearBufs near, float* obs_final,
                 uint64_t seed, uint64_t counter0, int auto_reset, const uint8_t* keys, const float* mouse, double* checksum,
                 uint32_t* status, uint64_t timeout_ticks, Backoff bo, uint32_t server_blocks) {
    Q1_PAIR_KERNEL_BODY;
}

// The same dispatch compiled for FOUR waves per SIMD (at most 128 VGPRs; no spills at ES = 1): 131 072 envs are resident at one env
// per lane, with two server waves per SIMD overlapping their float64 chains (3.3 -> 2.6 us per tick against two envs per lane), and
// smaller batches are no slower than with the 142 registers the compiler takes when left alone.
template <bool SPEC, int ES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
tick_pair_kernel_dense(Params p, StatePtrs s, int ticks, uint32_t tag0, uint64_t* mailbox, uint64_t* results, NearBufs near, float* obs_final,
                       uint64_t seed, uint64_t counter0, int auto_reset, const uint8_t* keys, const float* mouse, double* checksum,
                       uint32_t* status, uint64_t timeout_ticks, Backoff bo, uint32_t server_blocks) {
    Q1_PAIR_KERNEL_BODY;
}
#undef Q1_PAIR_KERNEL_BODY
