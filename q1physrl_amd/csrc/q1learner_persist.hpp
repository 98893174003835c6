// q1learner_persist.hpp - the PERSISTENT PPO learner of libq1env (device code): a whole run of SGD steps at the reference's minibatch
// size - RLlib's sgd_minibatch_size 128 x num_sgd_iter 30 over a 50 k-sample train batch: 11 719 steps per training iteration
// (q1physrl/train.py:60-64, data/params.yml:4-13) - as ONE dispatch.  VERDICT r4 item 3: at that size q1env_learner_sgd_step's four
// launches cost ~43 us for ~0.1 GFLOP, 99 % of the reference-configuration iteration.  Included by q1env_plearner.hip.
//
// Decomposition.  The two networks (policy 6-256-256-10, value 6-256-256-1: separate, vf_share_layers = false) never meet inside a
// step, so each gets its own group of G = 8 workgroups and its own barrier.  Workgroup g of a group OWNS hidden units U = [32 g, 32 g + 32)
// of both hidden layers: rows U of W1 / b1, rows U of W2 (torch layout [out][in]: all 256 inputs of its 32 units) / b2, columns U of
// W3 - masters, Adam moments and gradients in their torch layouts in global memory (private to the owner, plain cached accesses), the
// float16 operand images of its slice in LDS for the whole launch.  A step of 128 samples (four 32-sample tiles, one per wave):
//   P1   H1[:, U] = tanh(X W1[U]^T + b1[U])                      local; published in both orientations            -> barrier 1
//   P2   H2[:, U] = tanh(H1 W2[U]^T + b2[U])                     B operands: H1 straight from the exchange buffer, W2[U] from LDS
//   L3   Yp_g = H2[:, U] W3[:, U]^T  (partial logits)            from the activation registers; published        -> barrier 2
//   loss every workgroup sums the G partials in the same order, adds b3 and differentiates the PPO loss of its group's network for all
//        128 samples (q1ppo_loss.hpp, two lanes per sample; redundant across the group, identical bits)
//   B3   dZ2[:, U] = (dY W3[:, U]) (1 - H2[:, U]^2)              local; published                                -> barrier 3 (arrive)
//   G2   dW2[U, :] = dZ2[:, U]^T H1 in two 32-input tiles per wave, Adam, new W2[U] image, W2^T slice published; dW3[:, U], db2[U], db3
//   B2   dH1[:, U] = dZ2 W2[:, U]  (all of dZ2 from the exchange buffer; W2's column block gathered into LDS behind barrier 2)
//        dZ1[:, U] = dH1 (1 - H1[:, U]^2);  dW1[U], db1[U]; Adam
// Three group barriers per step (8 arrivals on one counter each), nothing else crosses workgroups.
//
// Schedule.  One wave per SIMD issues in order, so every wait is dead time unless the program itself puts independent work there.  The
// critical chain of a step is P1 -> b1 -> P2 -> b2 -> loss -> B3 -> b3 -> B2 -> (next) P1; everything else hangs off it and is placed in its
// waits:  tile 0 of G2 sits between the ARRIVAL at barrier 3 and its wait (the counter is read from inside the tile's optimizer arithmetic -
// an ordinary compiler-visible atomic, requested early and looked at late - and B2's operands are requested the moment it shows eight
// tickets, under the tile's stores);  tile 1, db2 and the statistics are DEFERRED into the next step: tile 1 + db2 between the arrival at
// barrier 1 and its wait (operands requested at the top of that step), dW3 / db3 behind P2's operand requests (under their latency), the
// statistics in barrier 2's window.  What a deferred piece reads stays valid that long (LDS arrays rewritten only later in the next step; H1^T
// and W2^T double-buffered by step parity) and what it produces is needed no earlier (W2's image by the next P2, W3's by its L3, b3 behind
// barrier 2); the last step's deferred work runs behind the loop.
//
// Matrix products: v_mfma_f32_32x32x16_f16, D[i][n] += sum_k A[i][k] B[n][k] with BOTH operands stored "K-contiguous" (row i / n, eight
// consecutive k per lane: lane (c, h) reads 16 bytes at row c, k = 16 s + 8 h).  The accumulator of lane (c, h) holds column n = c, rows
// i = (r & 3) + 8 (r >> 2) + 4 h - four consecutive i per register quad, one 8-byte piece.  Where both layouts of a result are needed (H1, H2,
// dZ2) the second is a transposition on the matrix pipe of the float16 pieces of the first (transpose32: two instructions, exact) - not a
// second product and a second round of activations.
//
// Exchange layouts.  Everything that crosses workgroups is stored in the order its CONSUMER's 64-lane requests read it ("operand-fragment
// order": Net below), addressed through one buffer resource per group with scalar block offsets: a request covers one contiguous KB (8 cache
// lines) where the row-major forms touched 32 or 64 scattered lines - the CU's one request per clock, not latency or arithmetic, had been
// what bounded every phase that touches exchanged data (20.2 -> 14.4 us per step from this alone).  For the same reason the W2 slice's
// optimizer state stays in registers during the loop instead of streaming through L2 every step (12.65 -> 11.7).
//
// Exchange protocol.  Two modes, decided per launch (see "placement" in the kernel).  Agent scope: published data are written through
// (sc0 sc1), "arrive" = s_waitcnt vmcnt(0), workgroup barrier, one relaxed increment; "wait" = poll, workgroup barrier, buffer_inv sc1.
// L2-local (all eight workgroups of a group on one XCD - the dispatcher's round-robin placement, verified by a census at every launch):
// plain stores, device-scope loads (sc1: never the CU's vector cache), no invalidation at all.  H1 / H1^T and W2^T have two buffers, by
// step parity (both are read by deferred work while a faster workgroup may be publishing the next step's).  Every wait is bounded (status
// word) - the 16 workgroups must be co-resident.
//
// Numerics: float16 operands, float32 accumulation, float32 masters / moments, torch.optim.Adam's update - the same recipe, loss scales and
// saturating gradient conversions as q1learner.hpp; the summation orders differ, so results agree with q1env_learner_sgd_step to float16
// operand rounding, not bit for bit (tests/test_hip_learner.py: per-step gradients <= 3e-3 of autograd's, a 391-step epoch against 391
// calls of q1env_learner_sgd_step).
#pragma once
#include "q1policy.hpp"
#include "q1ppo_loss.hpp"

namespace q1pl {

using q1pol::f16x8;
using q1pol::f32x16;
using q1pol::TANH_PRESCALE;

constexpr int G = 8;               // workgroups per network
constexpr int MB = 128;            // samples per step
constexpr int HID = 256;
constexpr uint32_t LD_W = 528;     // bytes per row of a 256-wide float16 row (+ 16 pad)
constexpr uint32_t LD_B = 272;     // 128-wide (sample-contiguous) row
constexpr uint32_t LD_16 = 48;     // 16-wide row
constexpr uint32_t LD_32 = 80;     // 32-wide row

// LDS map (bytes)
constexpr uint32_t L_W2OWN = 0;                              // [32 u][256 k]  2 log2(e) W2[U]      (P2)
constexpr uint32_t L_W2COL = L_W2OWN + 32 * LD_W;            // [32 j][256 k]  W2[k][j in U]        (B2)
constexpr uint32_t L_H2T = L_W2COL + 32 * LD_W;              // [32 u][128 b]
constexpr uint32_t L_DZ2T = L_H2T + 32 * LD_B;
constexpr uint32_t L_DZ1T = L_DZ2T + 32 * LD_B;
constexpr uint32_t L_DYT = L_DZ1T + 32 * LD_B;               // [32 o][128 b]  rows >= OUT stay zero
constexpr uint32_t L_XT = L_DYT + 32 * LD_B;                 // [32 i][128 b]  rows 0..5 x hi, 6 ones, 8..13 x lo, rest zero
constexpr uint32_t L_DY = L_XT + 32 * LD_B;                  // [128 b][16 o]
constexpr uint32_t L_XH = L_DY + MB * LD_16;                 // [128 b][16]    x hi, 1, 0, x lo, 1, 0
constexpr uint32_t L_W1 = L_XH + MB * LD_16;                 // [32 u][16]     c W1 row, c b1 hi, 0, c W1 row, c b1 lo, 0
constexpr uint32_t L_W3T = L_W1 + 32 * LD_16;                // [32 u][16 o]   W3[o][u]
constexpr uint32_t L_W3 = L_W3T + 32 * LD_16;                // [32 o][32 slots] W3[o][U], units in K-slot order (w3_slot)
constexpr uint32_t L_FL = L_W3 + 32 * LD_32;                 // floats: b2p[32] | red2[4][32] | red3[2][16] | stat[4][4] | bc[2] | flags
constexpr uint32_t L_ST = L_FL + 4096;                       // floats: the optimizer state (master, m, v) of the SMALL owned parameters for the launch:
                                                             //   sW1[3][32 u][6] | sB1[3][32] | sB2[3][32] | sW3[3][10 o][32 u] | sB3[3][16]
constexpr uint32_t LDS_BYTES = L_ST + 8192;                  // 118 272

struct Net {
    float* w1; float* b1; float* w2; float* b2; float* w3; float* b3;          // masters (torch layouts), updated in place
    float* gw1; float* gb1; float* gw2; float* gb2; float* gw3; float* gb3;    // the LAST step's gradients (inspection / tests)
    float* m; float* v;                                                          // moments, flat [w2 | b2 | w1 | b1 | w3 | b3] (q1learner.hpp AdamNet)
    int out_dim;
    // exchange buffers of this network's group
    // (H1, H1^T, dZ2 and the partial logits are exchanged in OPERAND-FRAGMENT order: the 16 bytes lane l of consumer wave w reads for K-step s
    //  sit at ((block * steps + s) * 64 + l) * 16, so that every 64-lane request covers one contiguous KB - 8 cache lines instead of the 32 or 64
    //  scattered ones of the row-major forms, which made the CU's single request port the bound of every phase that touches exchanged data)
    uint16_t* h1x;        // [2 parity][4 sample tiles w][16 K-steps s][64 lanes (c = sample, h)][8]: H1[32 w + c][16 s + 8 h ..]
    uint16_t* h1tx;       // [2 parity][8 unit tiles g][8 K-steps s][64 lanes (c = unit, h)][8]: H1[16 s + 8 h ..][32 g + c]
    uint16_t* dz2x;       // [4][16][64][8]: dZ2 in H1's form
    uint16_t* w2tx;       // [2 update parities][8 consumers g'][8 producers g][4 q][32 c][2 h][4]: W2[32 g + 8 q + 4 h ..][32 g' + c] - what lane (c, h) of the owner publishes in one
                          // 8-byte store, so a publishing request is one contiguous 512 B and a consumer's column block one contiguous 16 KB
    float* yp;            // [G][4 sample tiles w][4 output quads v][32 samples c][4]: partial logits 4 v .. 4 v + 3 of sample 32 w + c
    float* b3x;           // [16]: the output layer's bias as the group reads it (published by workgroup 0 after every step), zero padded
    float* w2st;          // [G][3 (master, m, v)][4 waves][8 slot quads][64 lanes][4]: the W2 slice's optimizer state in its OWNER LANE's order (the
                          // hand-over between the torch layouts and the owner lanes' registers, where it lives during the step loop)
    uint32_t* bar;        // arrival counter (own 256-byte line)
    const char* xbase;    // lowest address of this group's exchange workspace (the buffer resource of its exchange loads)
    float inv_b;          // d loss / d output travels multiplied by this (loss scale / 1: per-sample, not averaged)
    float inv_scale;      // gradient = sum over the samples x this
};

struct Args {
    q1::Params p;
    Net net[2];
    // the train batch (q1env_learner_batch; mouse_u = mouse_u_kernel's output in the workspace) and the minibatch schedule: step n reads rows idx[(n / spe) * epoch_stride + (n % spe) * 128 + b]
    const int64_t* idx; int64_t spe, epoch_stride;
    int64_t rows, idx_rows;    // rows of the train batch arrays / entries of idx (the host validated the schedule against them; the assertion build re-checks every access)
    const float* obs; const float* old_logits; int old_stride;
    int wide_old;              // old_logits rows are 8-byte aligned: read as five 8-byte loads
    const uint8_t* keys; const float* mouse_u; const float* logp_old; const float* adv; const float* value_old; const float* vtarg;
    float clip, vf_clip, vf_coeff, ent_coeff;
    const float* klc_dev;
    float lr, beta1, beta2, eps;
    int64_t steps;
    long long* step_count;     // adam_state + 0 (written by the epilogue: the snapshot + steps)
    const long long* step0_snap;   // its value at the start of the launch, copied by mouse_u_kernel (which runs first on the stream) into the status line: the two
                               // groups never synchronise with each other, so a late group must not read the count the other's epilogue has already advanced
    float* stats_acc;          // adam_state + 16: [entropy, kl, policy_loss, total (left to the host), vf_loss] running sums of per-step means
    uint32_t* saturation;      // optional uint32[4] (q1env_learner_batch.saturation_dev)
    uint32_t* status;          // uint32[4]: [0] != 0: a barrier timed out (value = 1 + barrier index), [1] the step it happened in,
                               // [2], [3]: exchange mode of the policy / value group (1 + the XCD all its workgroups share, 0 = agent scope)
    int allow_local;           // 0: always the agent-scope exchange mode (q1env_learner_set_exchange_mode)
    int census_skew;           // != 0 (tests): every workgroup reports a different XCD, so that the census fails and the automatic fallback is what runs
    unsigned long long* prof;  // optional uint64[20] (the rest of the status line): 10-ns ticks wave 0 of workgroup prof_g of the policy group spent per phase, summed over the steps
    int prof_g;
    uint64_t timeout_ticks;
};

// ---- the assertion build (-DQ1_CHECK, libq1env_check.so; tests/test_hip_learner.py runs a whole update of 11 730 steps under it).  Every access
// to an exchange buffer is compared with the group's workspace size before it is issued (and the buffer resource is BOUNDED, so that a
// missed case is dropped by the hardware instead of landing in somebody else's memory), every row index is compared with the number of
// rows before it is used as an address, every position in the schedule with the length of idx, every reading of a barrier counter with the
// range the protocol allows.  A failed assertion writes 0x100 + its code into status[0] (the step into status[1]) - the words a barrier
// timeout is reported in - and the access is skipped / the index replaced by 0: a status word, not a memory fault.
#ifdef Q1_CHECK
constexpr uint32_t CHK_XOFF = 0x101, CHK_ROW = 0x102, CHK_SCHED = 0x103, CHK_BAR = 0x104;
__device__ uint32_t* g_chk_status;                 // the launch's status line
__device__ uint32_t g_chk_xbytes;                  // bytes of one group's exchange workspace
__device__ unsigned long long g_chk_counts[4];     // exchange accesses / row indices / barrier readings checked (one count per wave and call), failures
__device__ __forceinline__ void chk_fail(uint32_t code, uint32_t detail) {
    atomicAdd(&g_chk_counts[3], 1ull);
    if (__hip_atomic_load(g_chk_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __hip_atomic_store(g_chk_status + 1, detail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g_chk_status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ bool chk_x(uint32_t voff, uint32_t soff, uint32_t bytes) {
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_chk_counts[0], 1ull);
    const uint64_t end = (uint64_t)voff + soff + bytes;
    if (end > g_chk_xbytes || ((voff + soff) & (bytes < 16u ? bytes - 1u : 15u)) != 0u) { chk_fail(CHK_XOFF, voff + soff); return false; }
    return true;
}
__device__ __forceinline__ int64_t chk_row(int64_t r, int64_t rows) {
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_chk_counts[1], 1ull);
    if ((uint64_t)r >= (uint64_t)rows) { chk_fail(CHK_ROW, (uint32_t)r); return 0; }
    return r;
}
#define Q1PL_XCHK(voff, soff, bytes) chk_x((voff), (soff), (bytes))
#else
#define Q1PL_XCHK(voff, soff, bytes) true
#endif

__device__ __forceinline__ f16x8 lds16(const unsigned char* base, uint32_t off) { return *reinterpret_cast<const f16x8*>(base + off); }
// Loads of EXCHANGED data: raw buffer loads over ONE resource per network (base = the group's exchange workspace, Net::xbase), address =
// base + per-lane byte offset (one 32-bit register, loop invariant) + per-buffer byte offset (scalar) + immediate - no 64-bit per-lane
// pointers to keep alive across the step - at DEVICE scope (sc1: never served by the CU's vector cache, which is not coherent with
// another CU's stores; served by the XCD's L2, where - in the L2-local mode - the producers' plain stores sit, so that no cache has to be
// invalidated behind a local barrier at all: buffer_inv sc1 also empties this XCD's L2 of clean lines and the exchanged operands then came
// back from the memory side at ~2 us per phase, measured).  They are ordinary compiler-visible loads: the compiler counts them (vmcnt), so
// a phase's operands can be REQUESTED a phase ahead and waited for where they are used (the first version issued them from inline
// assembly, request and wait in one block).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int XAUX = 16;                                          // cache policy of the exchange loads: sc1 (gfx940+: bit 4 of the aux operand)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xrsrc(const void* base) {
#ifdef Q1_CHECK
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)g_chk_xbytes, 0x00020000);   // ... bounded: out-of-range loads return 0, stores are dropped
#else
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);      // raw buffer, 32-bit data format, no swizzle (unbounded: every offset is a
                                                                                                        // compile-time layout constant + a lane term below 64 KB; the assertion build checks each)
#endif
}
__device__ __forceinline__ f16x8 xld16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    union { u32x4 u; f16x8 v; } o;
    if (!Q1PL_XCHK(voff, soff, 16u)) { o.u = u32x4{0, 0, 0, 0}; return o.v; }
    o.u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, XAUX);
    return o.v;
}
// private data of the owning lane (the W2 slice's optimizer state): the same addressing, ordinary cache policy
// (four floats per instruction: a wave may have 63 vector-memory instructions in flight, the 64th stalls until the oldest has completed - with
//  one float per instruction the 96 state loads and 96 state stores of a step ran into that limit behind every batch of operand requests)
__device__ __forceinline__ void xld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float* o) {
    if (!Q1PL_XCHK(voff, soff, 16u)) { o[0] = o[1] = o[2] = o[3] = 0.0f; return; }
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    o[0] = __uint_as_float(u.x); o[1] = __uint_as_float(u.y); o[2] = __uint_as_float(u.z); o[3] = __uint_as_float(u.w);
}
__device__ __forceinline__ void xst4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, const float* v) {
    const u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    if (!Q1PL_XCHK(voff, soff, 16u)) return;
    __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
}
// published data (see the exchange modes below): plain stores in the L2-local mode, agent-scope write-through (sc0 sc1) otherwise
__device__ __forceinline__ void xpub8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, uint64_t v, bool loc) {
    const u32x2 d = {(uint32_t)v, (uint32_t)(v >> 32)};
    if (!Q1PL_XCHK(voff, soff, 8u)) return;
    if (loc) __builtin_amdgcn_raw_buffer_store_b64(d, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(d, r, voff, soff, 17);
}
__device__ __forceinline__ void xpub16f(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float a, float b, float c, float d, bool loc) {
    const u32x4 u = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
    if (!Q1PL_XCHK(voff, soff, 16u)) return;
    if (loc) __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 17);
}
__device__ __forceinline__ f32x4 xld4f(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    union { u32x4 u; f32x4 v; } o;
    if (!Q1PL_XCHK(voff, soff, 16u)) { o.u = u32x4{0, 0, 0, 0}; return o.v; }
    o.u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, XAUX);
    return o.v;
}

__device__ __forceinline__ f32x16 mm(f16x8 a, f16x8 b, f32x16 acc) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0); }
__device__ __forceinline__ constexpr uint32_t fq(int q) { return 512u * (uint32_t)(q & 1) + 1024u * (uint32_t)(q >> 1); }   // piece q of a lane's fragment-order publication
__device__ __forceinline__ uint32_t w3_slot(uint32_t u) { return (u & 0x13u) | ((u & 4u) << 1) | ((u & 8u) >> 1); }   // K slot of unit u in transpose32's order (bits 2 and 3 swapped)
__device__ __forceinline__ uint32_t rrow(int r, uint32_t h) { return (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * h; }

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// 32x32 transposition on the matrix pipe.  pk[q] = a lane's accumulator registers 4 q .. 4 q + 3 of a [u][b] tile (lane = column b, register r =
// row u = row(r, h)) converted to float16 - the 8-byte pieces it publishes anyway.  As an A operand, slot e of K-step s of lane (b, h) is then
// row u(s, h, e) = (e & 3) + 16 s + 8 (e >> 2) + 4 h of column b; against the 0 / 1 operand E_s[n][(h, e)] = [u(s, h, e) == n] the product is
// D[b][n] = X[n][b]: the tile with lane = row u, registers = columns b, exact (one term per sum).  Replaces computing every both-orientation
// result twice (the second matrix product AND the second round of activations / derivative factors): 2 instructions instead of ~150.
struct TrOps { f16x8 e0, e1; };
__device__ __forceinline__ TrOps transpose_ops(uint32_t c, uint32_t h) {
    TrOps t;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t u0 = (uint32_t)((e & 3) + 8 * (e >> 2)) + 4u * h;
        t.e0[e] = u0 == c ? (_Float16)1.0f : (_Float16)0.0f;
        t.e1[e] = u0 + 16u == c ? (_Float16)1.0f : (_Float16)0.0f;
    }
    return t;
}
__device__ __forceinline__ f32x16 transpose32(const uint64_t (&pk)[4], const TrOps& t) {
    union { uint64_t u[2]; f16x8 v; } a0, a1;
    a0.u[0] = pk[0]; a0.u[1] = pk[1]; a1.u[0] = pk[2]; a1.u[1] = pk[3];
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.v, t.e0, zero16, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.v, t.e1, d, 0, 0, 0);
}
__device__ __forceinline__ uint64_t pack4(float a, float b, float c, float d) {
    union { f16x4 v; uint64_t u; } o;
    o.v = f16x4{(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
    return o.u;
}
// float32 -> float16 with saturation, four at a time: the largest magnitude by two v_max3 (operand modifiers take the absolute values), the
// count of elements beyond float16's range only when that maximum says there is one (a branch the wave almost never takes), one v_med3 per
// element to clamp - 9 instructions per quad (a step converts ~170 gradient elements per lane)
__device__ __forceinline__ void sat16x4(float (&x)[4], float& amax, uint32_t& nsat) {
    const float m4 = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
    amax = fmaxf(amax, m4);
    if (__builtin_expect(m4 > 65504.0f, 0)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) nsat += fabsf(x[j]) > 65504.0f ? 1u : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = __builtin_amdgcn_fmed3f(x[j], -65504.0f, 65504.0f);
}
__device__ __forceinline__ float r16(float x) { return (float)(_Float16)x; }                       // the value the float16 operand carries
__device__ __forceinline__ float act(float z) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z) + 1.0f); }   // tanh, z prescaled by 2 log2 e
// Two exchange modes (wave-uniform `loc`, decided once per launch - see "placement" in the kernel):
//   loc = false  the group's workgroups may sit on different XCDs: agent-scope write-through stores (sc1), tickets and polls at agent scope
//                (executed at the memory side), acquire = buffer_inv sc1 - what the language's memory model offers.
//   loc = true   all workgroups of the group were found on ONE XCD, whose L2 is then the point of coherence: plain stores (the CU's
//                vector cache is write-through, so an acknowledged store IS in that L2, where it stays dirty - an invalidation does not
//                drop dirty lines), tickets AND polls as read-modify-write atomics without a scope bit (performed in that L2: a poll is
//                an atomic add of zero, so that it reads exactly where the tickets are counted - a load, whatever its scope bits, might be
//                served by the CU's vector cache or sent past the L2), acquire = buffer_inv sc1 (empties the CU's vector cache; the first
//                version used the sc0 forms of poll and invalidate, which only by-pass the vector cache in threadgroup-split mode: it
//                dead-locked after ~50 steps on a stale counter line).  Nothing travels to the memory side: a barrier is cheaper and the
//                exchanged operands come from L2, not HBM / MALL.
__device__ __forceinline__ void pub4f(float* p, float a, bool loc) {
    if (loc) *p = a;
    else __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t xcc_id() {                     // which XCD this wave runs on (gfx940+: hardware register XCC_ID, bits 3:0)
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xFu;
}

// torch.optim.Adam's update (q1learner.hpp adam_update) with the hardware's 1-ulp square root and reciprocal (v_sqrt_f32, v_rcp_f32) in
// place of the IEEE expansions: this kernel is bound by its instruction count (one wave per SIMD: ~5 cycles per instruction of any kind),
// a workgroup updates 8 448 parameters per step, and the gradients already differ from q1env_learner_sgd_step's by float16 rounding.
// rs_bc2 = 1 / sqrt(bias_correction2), lr_bc1 = lr / bias_correction1 (per step, not per element).
__device__ __forceinline__ float adam1(float w, float g, float& m, float& v, float b1, float b2, float eps, float lr_bc1, float rs_bc2) {
    m = m + (g - m) * (1.0f - b1);
    v = v * b2 + ((1.0f - b2) * g) * g;
    const float denom = __builtin_amdgcn_sqrtf(v) * rs_bc2 + eps;
    return w - lr_bc1 * (m * __builtin_amdgcn_rcpf(denom));
}

// two elements at a time: the same operations in the same order (packed float32 multiply / add; square root and reciprocal per element)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (An explicitly fused form - v_pk_fma_f32, 13 instead of 17 instructions per pair, -0.1 us per step - was built and is equivalent (391 steps
//  against the four-launch path: the same 1e-4); it is NOT used: at seed 0 the reference-configuration run ended at 5 345 with it instead of
//  5 666 - seeds 1 and 2 went the other way over their first 300 iterations, i.e. the run is that sensitive to rounding - and the unfused form
//  keeps this kernel on the trajectory of the round's earlier runs.  STATE.md.)
__device__ __forceinline__ f32x2 adam2(f32x2 w, f32x2 g, f32x2& m, f32x2& v, float b1, float b2, float eps, float lr_bc1, float rs_bc2) {
    m = m + (g - m) * (1.0f - b1);
    v = v * b2 + ((1.0f - b2) * g) * g;
    const f32x2 sq = {__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)};
    const f32x2 denom = sq * rs_bc2 + eps;
    const f32x2 rc = {__builtin_amdgcn_rcpf(denom.x), __builtin_amdgcn_rcpf(denom.y)};
    return w - lr_bc1 * (m * rc);
}

// arrive: this workgroup's published stores have been acknowledged (by the memory side / by the XCD's L2); one ticket
__device__ __forceinline__ void bar_arrive(uint32_t* ctr, bool loc) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (loc) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (no scope bits: performed in the XCD's L2)
        else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// One reading of the counter.  Local mode: a read-modify-write that adds a zero the compiler cannot see through (an atomic add of a literal
// zero is folded into a workgroup-scope LOAD, which the CU's vector cache may serve: the stale-counter dead-lock of the first version), without
// scope bits, i.e. performed in the L2 where the tickets are counted.  An ordinary compiler-visible operation: its result can be
// REQUESTED early and waited for late (vmcnt counts it), which is how barrier 3 is polled from inside the weight-gradient phase.
__device__ __forceinline__ uint32_t poll(uint32_t* ctr, bool loc) {
    if (loc) {
        uint32_t zero = 0u;
        asm volatile("" : "+v"(zero));
        return __hip_atomic_fetch_add(ctr, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until `target` tickets have been drawn; false (for every thread) on timeout / abort.  `first`: a reading thread 0 requested earlier
// (0 = none: tickets are only ever counted upwards from 0 and target > 0)
__device__ __forceinline__ bool bar_wait(uint32_t* ctr, uint32_t target, bool loc, uint32_t* status, uint32_t which, uint32_t step, uint64_t timeout_ticks,
                                         int* s_ok, uint32_t first = 0u) {
    if (threadIdx.x == 0) {
        int ok = 1;
        uint64_t t0 = 0;
        uint32_t spins = 0;
#ifdef Q1_CHECK
        // a reading lies in [what was seen before, target + G): a workgroup can be at most one barrier ahead of the slowest (it cannot pass
        // barrier n + 1 before everybody - this workgroup included - has arrived there)
        atomicAdd(&g_chk_counts[2], 1ull);
        if (first >= target + (uint32_t)G) chk_fail(CHK_BAR, first);
#endif
        while (first < target) {
#ifdef Q1_CHECK
            const uint32_t before = first;
#endif
            first = poll(ctr, loc);
#ifdef Q1_CHECK
            atomicAdd(&g_chk_counts[2], 1ull);
            if (first < before || first >= target + (uint32_t)G) chk_fail(CHK_BAR, first);
#endif
            if (first >= target) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0u) {
                const uint64_t now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > timeout_ticks || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        __hip_atomic_store(status + 1, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(status, which + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ok = 0;
                    break;
                }
            }
        }
        *s_ok = ok;
    }
    __syncthreads();
    const bool ok = *s_ok != 0;
    if (!loc) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1 (the local mode reads exchanged data with device-scope loads instead: xld16)
    return ok;
}

// The squashed-Gaussian pre-image of every mouse action of the train batch, u = S ndtri((x - low) / (high - low)) (q1ppo_loss.hpp;
// action_dist.py:186-192): a function of the ACTION alone, so it is computed once per launch for all rows - a float64 polynomial of ~220
// instructions that each of the 30 epochs x 8 workgroups would otherwise re-evaluate per sample inside the step loop.
__global__ void __launch_bounds__(256)
mouse_u_kernel(int64_t rows, const float* __restrict__ mouse, float low, float high, float* __restrict__ u, const long long* __restrict__ step_count,
               long long* __restrict__ step0_snap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *step0_snap = *step_count;                     // (Args::step0_snap)
    if (i < rows) u[i] = SQUASH_SCALE * normcdfinvf((mouse[i] - low) / (high - low));
}

// NI = 0: a workgroup of the policy group (10 outputs, the PPO policy loss), NI = 1: of the value group (1 output, the value loss): two
// straight-line specialisations instead of run-time tests on the output count and the network in every loop over the outputs.
template <int NI, bool PROF>
__device__ __forceinline__ void persistent_learner_body(const Args& a, unsigned char* lds) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, c = lane & 31u, h = lane >> 5;
    // Placement.  The dispatcher deals workgroups to the eight XCDs round-robin by workgroup id, so of the 64 launched only those with
    // id % 8 == 0 (policy group) and id % 8 == 1 (value group) take part, g = id / 8: each group's eight workgroups then share an XCD and
    // its L2.  That is a property of today's dispatcher, not a promise: every workgroup publishes the XCD it actually runs on, and only if
    // all eight of a group agree does the group use the L2-local exchange mode (loc); otherwise the agent-scope mode, correct anywhere.
    constexpr uint32_t ni = (uint32_t)NI;
    const uint32_t g = blockIdx.x >> 3;
    const Net net = a.net[NI];
    // COLD arguments - the masters / moments / gradient pointers of the torch layouts, the counters: used by the prologue, by the epilogue and by
    // the last step's inspection stores only - are re-read from the kernel-argument segment where they are used (through a pointer the
    // compiler cannot see through, so the loads stay there): kept in scalar registers across the step loop they were ~40 of the ~100
    // there are, and the loop's hot scalars were spilled to vector-register lanes instead (228 v_readlane per step).
    typedef const __attribute__((address_space(4))) Args* KArgs;
    auto cold = [&]() __attribute__((always_inline)) -> KArgs {
        KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        return ka;
    };
    constexpr int OUT = NI == 0 ? 10 : 1;
    float* const fl = reinterpret_cast<float*>(lds + L_FL);
    float* const b2p = fl;                 // [32]
    float* const red2 = fl + 32;           // [4][32]
    float* const red3 = fl + 160;          // [16]
    int* const s_ok = reinterpret_cast<int*>(fl + 212);
    float* const statbuf = fl + 288;       // [3][128]: the per-sample statistics of a step (summed off the critical path)
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float c2 = TANH_PRESCALE;
    const uint32_t U0 = 32u * g;
    const TrOps tro = transpose_ops(c, h);
    const uint32_t bsm = 32u * w + c;                           // the sample this lane (and its partner lane ^ 32) differentiates the loss of
    // exchange loads: one resource, scalar byte offsets of the buffers, per-lane byte offsets (see xld16)
    const __amdgpu_buffer_rsrc_t xr = xrsrc(net.xbase);
    auto xoff = [&](const void* q) { return (uint32_t)(reinterpret_cast<const char*>(q) - net.xbase); };
    const uint32_t o_h1x = xoff(net.h1x), o_h1tx = xoff(net.h1tx), o_dz2x = xoff(net.dz2x), o_w2tx = xoff(net.w2tx), o_yp = xoff(net.yp), o_b3x = xoff(net.b3x);
    const uint32_t wu = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);        // the wave index as the scalar it is
    const uint32_t o_w2st = xoff(net.w2st);
    // ... publishes: (row, 4 h) of a [128 b][256 k] array (+ 2 U0 + 16 q), of H1^T (+ 2 U0 MB + 16 q), of W2^T (+ 2 U0 + 16384 t + 16 q), of the partial logits
    const uint32_t p_frag = c * 16u + 8u * h;                                    // a 4-element piece of a fragment-order array: + 512 (q & 1) + 1024 (q >> 1), block offset scalar
    const uint32_t p_yp = c * 16u + 512u * h;                                    // this lane's output quad h (+ 1024: quad 2 + h) of sample c
    const uint32_t s_ypg = o_yp + (g * 4u + wu) * 2048u;
    // ... the W2 slice's optimizer state, [g][kind][wave][8 slot quads][64 lanes][4] floats: lane offset + scalar (kind, wave, tile) + 1024 (r / 4)
    const uint32_t v_st = lane * 16u;
    auto s_st = [&](uint32_t kind, uint32_t t) { return o_w2st + ((g * 3u + kind) * 4u + wu) * 8192u + t * 4096u; };
    const uint32_t v_yp = c * 16u + h * 32768u;                                  // partial logits of sample (w, c), groups 4 h .. 4 h + 3 (8192 bytes apart): + 512 v

    // ---------------------------------------------------------------- prologue: operand images of the owned slice from the masters
    for (uint32_t off = tid * 16u; off < L_FL; off += 256u * 16u) *reinterpret_cast<uint4*>(lds + off) = uint4{0, 0, 0, 0};
    __syncthreads();
    // The small owned parameters' optimizer state (W1 / b1 / b2 rows U, W3 columns U, b3 on workgroup 0: ~600 of the 8 448 parameters) lives in
    // LDS for the launch - their updates sit on the step's critical path (dW1 closes it) and an L2 round trip per array was most of them.
    float* const sW1 = reinterpret_cast<float*>(lds + L_ST);       // [kind][u * 6 + i]
    float* const sB1 = sW1 + 3 * 192;                               // [kind][u]
    float* const sB2 = sB1 + 3 * 32;
    float* const sW3 = sB2 + 3 * 32;                                // [kind][o * 32 + u]
    float* const sB3 = sW3 + 3 * 320;                               // [kind][o]
    // flat offsets of the moment arrays ([w2 | b2 | w1 | b1 | w3 | b3], q1learner.hpp AdamNet)
    const size_t E_B2 = 65536, E_W1 = 65536 + 256, E_B1 = E_W1 + 1536, E_W3 = E_B1 + 256, E_B3 = E_W3 + (size_t)OUT * 256;
    auto small_state = [&](bool to_lds) {
        struct { float* w1; float* b1; float* b2; float* w3; float* b3; float* m; float* v; } net = {cold()->net[NI].w1, cold()->net[NI].b1, cold()->net[NI].b2, cold()->net[NI].w3,
                                                                                              cold()->net[NI].b3, cold()->net[NI].m, cold()->net[NI].v};
                           // LDS <-> the torch layouts (prologue / epilogue)
        for (uint32_t e = tid; e < 192u; e += 256u) {
            const size_t i1 = (size_t)U0 * 6 + e;
            if (to_lds) { sW1[e] = net.w1[i1]; sW1[192 + e] = net.m[E_W1 + i1]; sW1[384 + e] = net.v[E_W1 + i1]; }
            else { net.w1[i1] = sW1[e]; net.m[E_W1 + i1] = sW1[192 + e]; net.v[E_W1 + i1] = sW1[384 + e]; }
        }
        if (tid < 32u) {
            const size_t u = U0 + tid;
            if (to_lds) { sB1[tid] = net.b1[u]; sB1[32 + tid] = net.m[E_B1 + u]; sB1[64 + tid] = net.v[E_B1 + u];
                          sB2[tid] = net.b2[u]; sB2[32 + tid] = net.m[E_B2 + u]; sB2[64 + tid] = net.v[E_B2 + u]; }
            else { net.b1[u] = sB1[tid]; net.m[E_B1 + u] = sB1[32 + tid]; net.v[E_B1 + u] = sB1[64 + tid];
                   net.b2[u] = sB2[tid]; net.m[E_B2 + u] = sB2[32 + tid]; net.v[E_B2 + u] = sB2[64 + tid]; }
        }
        for (uint32_t e = tid; e < (uint32_t)OUT * 32u; e += 256u) {
            const size_t i3 = (size_t)(e >> 5) * HID + U0 + (e & 31u);
            if (to_lds) { sW3[e] = net.w3[i3]; sW3[320 + e] = net.m[E_W3 + i3]; sW3[640 + e] = net.v[E_W3 + i3]; }
            else { net.w3[i3] = sW3[e]; net.m[E_W3 + i3] = sW3[320 + e]; net.v[E_W3 + i3] = sW3[640 + e]; }
        }
        if (g == 0 && tid < (uint32_t)OUT) {
            if (to_lds) { sB3[tid] = net.b3[tid]; sB3[16 + tid] = net.m[E_B3 + tid]; sB3[32 + tid] = net.v[E_B3 + tid]; }
            else { net.b3[tid] = sB3[tid]; net.m[E_B3 + tid] = sB3[16 + tid]; net.v[E_B3 + tid] = sB3[32 + tid]; }
        }
    };
    small_state(true);
    // The W2 slice's masters and moments (8 192 of the 8 448 parameters a workgroup owns) are kept, for the launch, in the order their owner
    // lanes use them - lane (c, h) of wave w owns inputs k = 64 w + 32 t + c of units U0 + row(r, h): slot 16 t + r - so that a step's 96
    // loads and 96 stores per lane are fully coalesced and addressed base + immediate; the torch layouts are read here and rewritten
    // once at the end.
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float w4[4], m4[4], v4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t e = (size_t)(U0 + rrow(4 * q + j, h)) * HID + 64u * w + 32u * (uint32_t)t + c;
                w4[j] = net.w2[e]; m4[j] = net.m[e]; v4[j] = net.v[e];
            }
            xst4(xr, v_st + 1024u * (uint32_t)q, s_st(0, t), w4); xst4(xr, v_st + 1024u * (uint32_t)q, s_st(1, t), m4); xst4(xr, v_st + 1024u * (uint32_t)q, s_st(2, t), v4);
        }
    {
        const uint32_t u = tid >> 3, k0 = (tid & 7u) * 32u;                            // 8 threads per owned unit, 32 inputs each
        const float* src = net.w2 + (size_t)(U0 + u) * HID + k0;
        for (uint32_t k = 0; k < 32u; ++k) {
            const float wv = src[k];
            *reinterpret_cast<_Float16*>(lds + L_W2OWN + u * LD_W + 2u * (k0 + k)) = (_Float16)(c2 * wv);
            union { _Float16 hh; uint16_t b; } o; o.hh = (_Float16)wv;
            const uint32_t kk = k0 + k;
            __hip_atomic_store(net.w2tx + 65536 + (((size_t)(kk >> 5) * 8 + g) * 4 + (u >> 3)) * 256 + (kk & 31u) * 8u + ((u >> 2) & 1u) * 4u + (u & 3u), o.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < 32u) {
        const uint32_t u = tid;
        const float bs = c2 * net.b1[U0 + u];
        const _Float16 bhi = (_Float16)bs, blo = (_Float16)(bs - (float)bhi);
        _Float16* row = reinterpret_cast<_Float16*>(lds + L_W1 + u * LD_16);
        for (int i = 0; i < 6; ++i) { const _Float16 wv = (_Float16)(c2 * net.w1[(size_t)(U0 + u) * 6 + i]); row[i] = wv; row[8 + i] = wv; }
        row[6] = bhi; row[14] = blo;
        b2p[u] = c2 * net.b2[U0 + u];
        for (int o = 0; o < OUT; ++o) {
            const _Float16 wv = (_Float16)net.w3[(size_t)o * HID + U0 + u];
            *reinterpret_cast<_Float16*>(lds + L_W3 + (uint32_t)o * LD_32 + 2u * w3_slot(u)) = wv;
            *reinterpret_cast<_Float16*>(lds + L_W3T + u * LD_16 + 2u * (uint32_t)o) = wv;
        }
    }
    if (tid < MB) *reinterpret_cast<_Float16*>(lds + L_XT + 6u * LD_B + 2u * tid) = (_Float16)1.0f;    // the ones row of [x | 1]^T
    if (g == 0 && tid < 16u) __hip_atomic_store(net.b3x + tid, (int)tid < OUT ? net.b3[tid] : 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // one agent-scope barrier (its own counter) behind the prologue's publications: also carries the XCD census
    bool loc = false;
    {
        if (tid == 0) __hip_atomic_store(net.bar + 16 + g, xcc_id() + 1u + (a.census_skew ? 16u * g : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bar_arrive(net.bar + 8, false);
        if (!bar_wait(net.bar + 8, (uint32_t)G, false, a.status, 3u, 0u, a.timeout_ticks, s_ok)) return;
        uint32_t same = 1u;
        const uint32_t mine = __hip_atomic_load(net.bar + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t k = 1; k < (uint32_t)G; ++k) same &= __hip_atomic_load(net.bar + 16 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine ? 1u : 0u;
        loc = a.allow_local != 0 && same != 0u && mine != 0u;
        // (the mode is REPORTED by the epilogue - status[2 + ni] - not here: no one-lane store in front of the step loop)
    }
    const float klc = *a.klc_dev;
    const long long step0 = *a.step0_snap;
    double pw1 = pow((double)a.beta1, (double)step0), pw2 = pow((double)a.beta2, (double)step0);
    float st_acc[3] = {0.0f, 0.0f, 0.0f};                       // running sums of the step statistics (thread 0 of workgroup 0 of each group)
    float amax = 0.0f;
    uint32_t nsat = 0;
    uint32_t bar_n = 0;                                         // barriers passed
    float lr_prev = 0.0f, rs_prev = 0.0f;                       // the previous step's bias corrections (its deferred work)
    // the row of the FIRST step (every later step's is fetched one step ahead): the sample whose observation this lane stages and whose loss this
    // lane pair differentiates, 32 w + c
    // (no division in the loop: the position of the next step's window is kept incrementally)
    int64_t win = 0;                                            // offset of the CURRENT step's 128-row window in idx
    int64_t in_epoch = 0;                                       // its index within the epoch
#ifdef Q1_CHECK
    auto row_at = [&](int64_t window, uint32_t b) -> int64_t {
        const int64_t at = window + (int64_t)b;
        if ((uint64_t)at >= (uint64_t)(a.idx ? a.idx_rows : a.rows)) { chk_fail(CHK_SCHED, (uint32_t)at); return 0; }
        return chk_row(a.idx ? a.idx[at] : at, a.rows);
    };
#else
    auto row_at = [&](int64_t window, uint32_t b) -> int64_t { return a.idx ? a.idx[window + (int64_t)b] : window + (int64_t)b; };
#endif
    // (Round 5 kept this index in two variables "because the kernel faulted with one".  The cause was found in round 6 and is not in this code:
    //  the register allocator had placed copies in front of a join block's exec restore - q1physrl_amd/isa_check.py, which now rejects any
    //  build that has such a block, whatever the source looks like.)
    int64_t src = row_at(0, bsm);
    // ... and the observation row itself (24 bytes from HBM at a random row: ~2 us of latency that would otherwise open every step)
    float oxn[6];
    auto request_obs = [&]() {
        const float2* o2 = reinterpret_cast<const float2*>(a.obs + (size_t)src * 6);
        const float2 p0 = o2[0], p1 = o2[1], p2 = o2[2];
        oxn[0] = p0.x; oxn[1] = p0.y; oxn[2] = p1.x; oxn[3] = p1.y; oxn[4] = p2.x; oxn[5] = p2.y;
    };
    request_obs();
    __syncthreads();

    // Lane-dependent address parts, made OPAQUE once per step (an empty assembly statement "modifies" them): otherwise every
    // `base + constant` is a loop invariant, is hoisted out of the step loop and occupies a register of its own for the whole launch (~150
    // of them, parked in accumulation registers and fetched back with v_accvgpr_read before each use: a quarter of the step's instructions
    // was such traffic); as values of the current iteration they stay ONE register each and the constants fold into the instructions' offset fields.
#define Q1PL_OPAQUE(x) asm volatile("" : "+v"(x))
    uint32_t lW = 0, lB = 0, l16 = 0, lS16 = 0, l32 = 0, lwB = 0, lCol = 0, lOwn = 0;            // LDS
    uint32_t vYp = 0, pFrag = 0, pYp = 0, vSt = 0;                                     // exchange / state buffer offsets
    auto step_bases = [&]() __attribute__((always_inline)) {
        lW = c * LD_W + 16u * h; lB = c * LD_B + 16u * h; l16 = c * LD_16 + 16u * h; lS16 = (32u * w + c) * LD_16 + 16u * h; l32 = c * LD_32 + 16u * h;
        lwB = c * LD_B + 2u * (32u * w + 4u * h); lCol = (tid & 31u) * LD_W + 16u * (tid >> 5);
        lOwn = 4u * h * LD_W + 2u * (64u * w + c);
        vYp = v_yp; pFrag = p_frag; pYp = p_yp; vSt = v_st;
        Q1PL_OPAQUE(lW); Q1PL_OPAQUE(lB); Q1PL_OPAQUE(l16); Q1PL_OPAQUE(lS16); Q1PL_OPAQUE(l32); Q1PL_OPAQUE(lwB); Q1PL_OPAQUE(lCol); Q1PL_OPAQUE(lOwn);
        Q1PL_OPAQUE(vYp); Q1PL_OPAQUE(pFrag); Q1PL_OPAQUE(pYp); Q1PL_OPAQUE(vSt);
    };
    // (two parts: the arithmetic + the new float16 image in LDS;  the global stores - state back, the W2^T rows published: the poll of
    //  barrier 3 sits between them, so that it never queues behind 64 stores)
    auto adam_tile = [&](const int t, const int q0, const int q1, const f32x16& acc, float (&w2v)[16], float (&m2v)[16], float (&v2v)[16], const float lr_bc1, const float rs_bc2,
                         const bool store_grads) __attribute__((always_inline)) {
        const uint32_t k = 64u * w + 32u * (uint32_t)t + c;
#pragma unroll
        for (int q = q0; q < q1; ++q) {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const int r = 4 * q + j;
                const f32x2 gr = f32x2{acc[r], acc[r + 1]} * net.inv_scale;
                f32x2 mm2 = {m2v[r], m2v[r + 1]}, vv2 = {v2v[r], v2v[r + 1]};
                const f32x2 wn = adam2(f32x2{w2v[r], w2v[r + 1]}, gr, mm2, vv2, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                w2v[r] = wn.x; w2v[r + 1] = wn.y; m2v[r] = mm2.x; m2v[r + 1] = mm2.y; v2v[r] = vv2.x; v2v[r + 1] = vv2.y;
                const f32x2 wi = c2 * wn;
                *reinterpret_cast<_Float16*>(lds + L_W2OWN + (uint32_t)((r & 3) + 8 * (r >> 2)) * LD_W + lOwn + 64u * (uint32_t)t) = (_Float16)wi.x;
                *reinterpret_cast<_Float16*>(lds + L_W2OWN + (uint32_t)(((r + 1) & 3) + 8 * ((r + 1) >> 2)) * LD_W + lOwn + 64u * (uint32_t)t) = (_Float16)wi.y;
                if (store_grads) {              // (inspection / tests: a cold block - its 32 addresses must not be hoisted out of the step loop)
                    uint32_t kk = k;
                    Q1PL_OPAQUE(kk);
                    float* const gw2 = cold()->net[NI].gw2;
                    gw2[(size_t)(U0 + rrow(r, h)) * HID + kk] = gr.x; gw2[(size_t)(U0 + rrow(r + 1, h)) * HID + kk] = gr.y;
                }
            }
        }
    };
    // (par2: which of the two W2^T exchange buffers - the parity of the step the update belongs to)
    auto store_tile = [&](const int t, const uint32_t par2, const float (&w2v)[16], const float (&m2v)[16], const float (&v2v)[16], const bool state = true) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            xpub8(xr, pFrag + 512u * (uint32_t)q, o_w2tx + par2 * 131072u + ((2u * wu + (uint32_t)t) * 8u + g) * 2048u, pack4(w2v[4 * q], w2v[4 * q + 1], w2v[4 * q + 2], w2v[4 * q + 3]), loc);
        if (state) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { xst4(xr, vSt + 1024u * (uint32_t)q, s_st(0, (uint32_t)t), w2v + 4 * q); xst4(xr, vSt + 1024u * (uint32_t)q, s_st(1, (uint32_t)t), m2v + 4 * q); xst4(xr, vSt + 1024u * (uint32_t)q, s_st(2, (uint32_t)t), v2v + 4 * q); }
        }
    };
    // The W2 slice's optimizer state (2 tiles x 48 values per lane) stays in REGISTERS for the whole launch: read here once, written back behind
    // the loop.  Streamed through L2 every step it was 48 KB of the ~125 KB a wave moved per step, on a CU whose address unit was the bound of
    // several phases: 12.65 -> 11.7 us per step (the same bits).
    float wA[16], mA[16], vA[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { xld4(xr, v_st + 1024u * (uint32_t)q, s_st(0, 0), wA + 4 * q); xld4(xr, v_st + 1024u * (uint32_t)q, s_st(1, 0), mA + 4 * q); xld4(xr, v_st + 1024u * (uint32_t)q, s_st(2, 0), vA + 4 * q); }
    // TILE 1 of a step's W2 gradient (inputs k = 64 w + 32 + c) is DEFERRED like the small gradients: its operands (H1^T of that step: the
    // exchange buffer of that parity stays intact for two steps; dZ2^T in LDS: rewritten by the next step's B3) are
    // requested at the top of the NEXT step and it runs between that step's arrival at barrier 1 and the wait for it - ~1.3 us of work in
    // a window in which the workgroup would otherwise only wait - instead of between barrier 3 and B2, on the step's critical path.  Its
    // new weights are needed by the next P2 (LDS image: behind the barrier wait's workgroup barrier) and by the other workgroups' column
    // gathers, which therefore read W2^T behind barrier 2 and from the buffer of the update's parity (two buffers: a fast workgroup
    // publishes tile 0 of the NEXT update right behind its arrival at barrier 3, possibly before a slow one has gathered).
    float wB[16], mB[16], vB[16];                                // (tile 1's optimizer state: see wA)
#pragma unroll
    for (int q = 0; q < 4; ++q) { xld4(xr, v_st + 1024u * (uint32_t)q, s_st(0, 1), wB + 4 * q); xld4(xr, v_st + 1024u * (uint32_t)q, s_st(1, 1), mB + 4 * q); xld4(xr, v_st + 1024u * (uint32_t)q, s_st(2, 1), vB + 4 * q); }
    f16x8 hU[8];
    auto tile1_request = [&](const uint32_t parity) __attribute__((always_inline)) {
        const uint32_t sb = o_h1tx + parity * (uint32_t)(MB * HID * 2) + wu * 16384u + 8192u;
#pragma unroll
        for (int s = 0; s < 8; ++s) hU[s] = xld16(xr, vSt + 1024u * (uint32_t)(s & 3), sb + 4096u * (uint32_t)(s >> 2));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tile1_run = [&](const uint32_t parity, const float lr_bc1, const float rs_bc2, const bool store_grads) __attribute__((always_inline)) {
        f32x16 acc = zero16;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = mm(lds16(lds, L_DZ2T + lB + 32u * (uint32_t)s), hU[s], acc);
        adam_tile(1, 0, 4, acc, wB, mB, vB, lr_bc1, rs_bc2, store_grads);
    };
    // The SMALL gradients of a step - db2[U] (wave 2), dW3[:, U] (wave 1), db3 (wave 3 of workgroup 0) and the step's statistics -, their
    // optimizer updates and new operand images.  They need nothing but this workgroup's LDS (dY^T, H2^T, red2, the ones row of [x | 1]^T,
    // statbuf) and nobody needs their results before the next step's P2 / L3 / loss, so they are DEFERRED into the next step, each to a
    // place where its wave would otherwise wait: db2 into the barrier-1 window; dW3 and db3 behind P2's operand requests, under their
    // latency - one workgroup barrier in P2 then orders the new W3 image before L3 and these reads of H2^T before its rewriting; the
    // statistics into the barrier-2 window.  b3 is republished by db3: every workgroup read the old one before it arrived at barrier 3,
    // the new one is read behind barrier 2.  lr_bc1 / rs_bc2: the bias corrections of the step the gradients belong to.
    auto grads_b2 = [&](const bool store_grads, const float lr_bc1, const float rs_bc2) __attribute__((always_inline)) {
        if (wu == 2u && h == 0u) {                               // db2[U]
            const size_t u = U0 + c;
            float b2v = sB2[c], mv = sB2[32 + c], vv = sB2[64 + c];
            const float gr = (((red2[c] + red2[32u + c]) + red2[64u + c]) + red2[96u + c]) * net.inv_scale;
            b2v = adam1(b2v, gr, mv, vv, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
            b2p[c] = c2 * b2v;
            sB2[c] = b2v; sB2[32 + c] = mv; sB2[64 + c] = vv;
            if (store_grads) cold()->net[NI].gb2[u] = gr;
        }
    };
    auto grads_w3b3 = [&](const bool store_grads, const float lr_bc1, const float rs_bc2) __attribute__((always_inline)) {
        if (wu == 1u) {                                          // dW3[:, U]: lane = owned unit, registers = outputs (o = row(r, h) < OUT <= 10: r < 8)
            float w3v[8], m3v[8], v3v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t o = rrow(r, h);
                const uint32_t i3 = ((int)o < OUT ? o : 0u) * 32u + c;
                w3v[r] = sW3[i3]; m3v[r] = sW3[320 + i3]; v3v[r] = sW3[640 + i3];
            }
            f32x16 acc = zero16;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                acc = mm(lds16(lds, L_DYT + lB + 32u * (uint32_t)s), lds16(lds, L_H2T + lB + 32u * (uint32_t)s), acc);
#pragma unroll
            for (int r = 0; r < 8; r += 2) {                    // (pairs: rows o, o + 1 - the packed optimizer arithmetic; slots beyond OUT are computed and dropped)
                const uint32_t o = rrow(r, h);
                const f32x2 gr = f32x2{acc[r], acc[r + 1]} * net.inv_scale;
                f32x2 m2 = {m3v[r], m3v[r + 1]}, v2 = {v3v[r], v3v[r + 1]};
                const f32x2 wn = adam2(f32x2{w3v[r], w3v[r + 1]}, gr, m2, v2, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                const float wq[2] = {wn.x, wn.y}, mq[2] = {m2.x, m2.y}, vq[2] = {v2.x, v2.y}, gq[2] = {gr.x, gr.y};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if ((int)(o + (uint32_t)e) < OUT) {
                        const uint32_t oo = o + (uint32_t)e, i3 = oo * 32u + c;
                        *reinterpret_cast<_Float16*>(lds + L_W3 + oo * LD_32 + 2u * w3_slot(c)) = (_Float16)wq[e];
                        *reinterpret_cast<_Float16*>(lds + L_W3T + c * LD_16 + 2u * oo) = (_Float16)wq[e];
                        sW3[i3] = wq[e]; sW3[320 + i3] = mq[e]; sW3[640 + i3] = vq[e];
                        if (store_grads) cold()->net[NI].gw3[(size_t)oo * HID + U0 + c] = gq[e];
                    }
                }
            }
        }
        if (g == 0 && wu == 3u) {                               // db3 / b3
            f32x16 acc_b3 = zero16;                             // [o][i']: lane (c = 6, h) holds db3[o = row(r, h)] (times the loss scale): the ones column
#pragma unroll
            for (int s = 0; s < 8; ++s)
                acc_b3 = mm(lds16(lds, L_DYT + lB + 32u * (uint32_t)s), lds16(lds, L_XT + lB + 32u * (uint32_t)s), acc_b3);
            // one output per lane for the optimizer: the ten sums cross through LDS (red3) instead of eight masked updates in a row on two lanes
            if (c == 6u) {
#pragma unroll
                for (int r = 0; r < 8; ++r) red3[rrow(r, h)] = acc_b3[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if ((int)lane < OUT) {
                const uint32_t o = lane;
                float bv = sB3[o], mv = sB3[16 + o], vv = sB3[32 + o];
                const float gr = red3[o] * net.inv_scale;
                bv = adam1(bv, gr, mv, vv, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                sB3[o] = bv; sB3[16 + o] = mv; sB3[32 + o] = vv;
                if (store_grads) cold()->net[NI].gb3[o] = gr;
                pub4f(net.b3x + o, bv, loc);
            }
        }
    };
    auto grads_stats = [&]() __attribute__((always_inline)) {
        if (g == 0 && wu == 3u) {
            float sv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float v = statbuf[k * MB + lane] + statbuf[k * MB + 64 + lane];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                sv[k] = v;
            }
            if (lane == 0) { st_acc[0] += sv[0] * (1.0f / (float)MB); st_acc[1] += sv[1] * (1.0f / (float)MB); st_acc[2] += sv[2] * (1.0f / (float)MB); }
        }
    };
    // (PROF: the instantiation launched under Q1_LEARNER_PROF - the product kernel carries neither the 40 registers nor the clock reads)
    const bool profiling = PROF && a.prof != nullptr && blockIdx.x == 8u * (uint32_t)(a.prof_g & 7) && tid == 64u * (uint32_t)(a.prof_g >> 3);      // (lane 0 of one wave of a workgroup of the policy group)
    unsigned long long pacc[PROF ? 20 : 1] = {};
    uint64_t tprev = profiling ? wall_clock64() : 0;
#define Q1PL_STAMP(k) do { if constexpr (PROF) { if (profiling) { const uint64_t now_ = wall_clock64(); pacc[k] += now_ - tprev; tprev = now_; } } } while (0)
    for (int64_t step = 0; step < a.steps; ++step) {
        const uint32_t par = (uint32_t)(step & 1);
        step_bases();
        const bool last = step + 1 == a.steps;
        const uint32_t s_h1x = o_h1x + par * (uint32_t)(MB * HID * 2), s_h1tx = o_h1tx + par * (uint32_t)(MB * HID * 2);
        // ------------------------------------------------------------ this step's rows: everything that depends only on the row index is
        // requested NOW (observation for the operand rows; the loss's per-sample inputs, which are not needed before barrier 2)
        pw1 *= (double)a.beta1;                                   // beta^t, t = step0 + step + 1 (every thread: two float64 multiplies)
        pw2 *= (double)a.beta2;
        float ox[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) ox[i] = oxn[i];              // (requested during the previous step's loss phase)
        if (h == 0u) {                                           // (lanes 0 .. 31 of every wave: sample 32 w + c - the rows this wave's own P1 reads)
            _Float16 hi[6], lo[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float xs = fminf(fmaxf(ox[i], -65504.0f), 65504.0f);
                hi[i] = (_Float16)xs;
                lo[i] = (_Float16)(xs - (float)hi[i]);
            }
            _Float16* row = reinterpret_cast<_Float16*>(lds + L_XH + bsm * LD_16);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                row[i] = hi[i]; row[8 + i] = lo[i];
                *reinterpret_cast<_Float16*>(lds + L_XT + (uint32_t)i * LD_B + 2u * bsm) = hi[i];
                *reinterpret_cast<_Float16*>(lds + L_XT + (uint32_t)(8 + i) * LD_B + 2u * bsm) = lo[i];
            }
            row[6] = (_Float16)1.0f; row[7] = (_Float16)0.0f; row[14] = (_Float16)1.0f; row[15] = (_Float16)0.0f;
        }
        // (wave-level ordering is enough: P1 reads this wave's own rows of L_XH; the transposed image L_XT is read workgroup barriers later)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float lr_bc1 = a.lr / (float)(1.0 - pw1), rs_bc2 = 1.0f / sqrtf((float)(1.0 - pw2));    // lr / bias_correction1, 1 / sqrt(bias_correction2)

        if (step > 0) tile1_request(par ^ 1u);                    // (the previous step's deferred tile: see tile1_run)
        // ------------------------------------------------------------ P1: layer 1 of the owned units, both orientations
        float h1B[16];                                          // tanh(H1)[b = 32 w + row(r)][u = c] as the float16 operand carries it
        {
            const f16x8 a1 = lds16(lds, L_W1 + l16);
            const f16x8 x1 = lds16(lds, L_XH + lS16);
            const f32x16 dA = mm(a1, x1, zero16);               // [u][b]: lane = sample, registers = units
            uint64_t pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pk[q] = pack4(act(dA[4 * q]), act(dA[4 * q + 1]), act(dA[4 * q + 2]), act(dA[4 * q + 3]));
                xpub8(xr, pFrag + fq(q), s_h1x + (wu * 16u + 2u * g) * 1024u, pk[q], loc);
            }
            const f32x16 dT = transpose32(pk, tro);             // [b][u]: lane = unit, registers = samples (the float16 values, exactly)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) h1B[4 * q + j] = dT[4 * q + j];
                xpub8(xr, pFrag + fq(q), s_h1tx + (g * 8u + 2u * wu) * 1024u, pack4(dT[4 * q], dT[4 * q + 1], dT[4 * q + 2], dT[4 * q + 3]), loc);
            }
        }
        Q1PL_STAMP(0);                                          // minibatch rows + P1 + publish
        bar_arrive(net.bar, loc);
        uint32_t seen1 = 0u;
        if (step > 0) {                                          // the previous step's deferred work: see tile1_run, grads_b2
            tile1_run(par ^ 1u, lr_prev, rs_prev, false);
            seen1 = tid == 0 ? poll(net.bar, loc) : 0u;         // (requested ~1 us behind the arrival and IN FRONT of the tile's stores, looked at behind the rest of the window's work)
            store_tile(1, par ^ 1u, wB, mB, vB, false);
            grads_b2(false, lr_prev, rs_prev);
        }
        Q1PL_STAMP(17);                                         // (the barrier-1 window's work)
        const float lr_prev2 = lr_prev, rs_prev2 = rs_prev;
        lr_prev = lr_bc1; rs_prev = rs_bc2;
        const int64_t src_now = src;
        // the NEXT step's row indices, requested while the barrier is in flight
        if (!last) {
            if (++in_epoch == a.spe) { in_epoch = 0; win += a.epoch_stride - (a.spe - 1) * MB; } else { win += MB; }
            src = row_at(win, bsm);
        }
        if (!bar_wait(net.bar, (uint32_t)G * ++bar_n, loc, a.status, 0u, (uint32_t)step, a.timeout_ticks, s_ok, seen1)) return;
        Q1PL_STAMP(1);                                          // barrier 1

        // ------------------------------------------------------------ W2's column block (rows j in U of W2^T), then P2 + the partial logits
        float h2A[16];
        // the loss's per-sample inputs (not needed before barrier 2) are requested BEHIND this phase's operands (the memory counter retires
        // in order: requested first, their HBM latency would sit in front of the first matrix product): policy group: keys, logp_old, adv,
        // the mouse pre-image, the old logits row;  value group: value_old, vtarg (in the same registers)
        uint32_t in_kb = 0u;
        float in_a = 0.0f, in_b = 0.0f, in_c = 0.0f;
        float oldrow[10];
        {
            // every global operand of this phase is requested first: H1 rows of this wave's tile (16 K-steps) and the column block
            f16x8 bH[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) bH[s] = xld16(xr, vSt + 1024u * (uint32_t)(s & 3), s_h1x + wu * 16384u + 4096u * (uint32_t)(s >> 2));
            {
                const size_t sl = (size_t)src_now;
                if (ni == 0) { in_kb = (uint32_t)a.keys[sl]; in_a = a.mouse_u[sl]; in_b = a.logp_old[sl]; in_c = a.adv[sl]; }
                else { in_a = a.value_old[sl]; in_b = a.vtarg[sl]; }
                if (ni == 0 && a.wide_old) {
                    const float2* r2 = reinterpret_cast<const float2*>(a.old_logits + sl * (size_t)a.old_stride);
#pragma unroll
                    for (int o = 0; o < 10; o += 2) { const float2 p2 = r2[o >> 1]; oldrow[o] = p2.x; oldrow[o + 1] = p2.y; }
                } else {
#pragma unroll
                    for (int o = 0; o < 10; ++o) oldrow[o] = ni == 0 ? a.old_logits[sl * (size_t)a.old_stride + (size_t)o] : 0.0f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // (the requests above are issued HERE, all of them: the scheduler would otherwise sink each to its first use)
            if (step > 0) grads_w3b3(false, lr_prev2, rs_prev2);      // (the previous step's, under the requests' latency)
            f32x16 accA = zero16;                               // [u][b]: lane = sample, registers = owned units
#pragma unroll
            for (int s = 0; s < 16; ++s) accA = mm(lds16(lds, L_W2OWN + lW + 32u * (uint32_t)s), bH[s], accA);
            uint64_t pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(b2p + 8 * q + 4 * h);
                const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
                float tA[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { tA[j] = act(accA[4 * q + j] + bb[j]); h2A[4 * q + j] = r16(tA[j]); }
                pk[q] = pack4(tA[0], tA[1], tA[2], tA[3]);
            }
            __syncthreads();                                    // (W3's new image is complete; nobody reads the previous step's H2^T any more)
            // the partial logits straight from the registers: the lane's pieces ARE a B operand whose K slots run over the units in the order
            // u(s, h, e) (see transpose32); W3's image in LDS is stored in that order (w3_slot)
            union { uint64_t u[2]; f16x8 v; } hb0, hb1;
            hb0.u[0] = pk[0]; hb0.u[1] = pk[1]; hb1.u[0] = pk[2]; hb1.u[1] = pk[3];
            f32x16 accY = mm(lds16(lds, L_W3 + l32), hb0.v, zero16);                        // [o][b]: lane = sample, registers = outputs
            accY = mm(lds16(lds, L_W3 + l32 + 32u), hb1.v, accY);
            const f32x16 hT2 = transpose32(pk, tro);            // H2^T[u][b] (lane = unit, registers = samples) for dW3
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint64_t*>(lds + L_H2T + lwB + 16u * (uint32_t)q) = pack4(hT2[4 * q], hT2[4 * q + 1], hT2[4 * q + 2], hT2[4 * q + 3]);
            xpub16f(xr, pYp, s_ypg, accY[0], accY[1], accY[2], accY[3], loc);              // outputs 4 h .. 4 h + 3: quad h
            xpub16f(xr, pYp + 1024u, s_ypg, accY[4], accY[5], accY[6], accY[7], loc);      // outputs 8 + 4 h ..: quad 2 + h
        }
        Q1PL_STAMP(2);                                          // W2 column gather + P2 + partial logits
        bar_arrive(net.bar, loc);
        if (step > 0) grads_stats();                             // (the previous step's statistics: statbuf is rewritten behind this barrier)
        if (!bar_wait(net.bar, (uint32_t)G * ++bar_n, loc, a.status, 1u, (uint32_t)step, a.timeout_ticks, s_ok)) return;
        Q1PL_STAMP(3);                                          // barrier 2

        // ------------------------------------------------------------ outputs, loss gradient: two lanes per sample (lanes l and l ^ 32 of
        // wave w: sample 32 w + c; q1ppo_loss.hpp PAIR form - the four keys' terms, half of the function's instructions, are split
        // between the pair), so all four waves share the work; identical in every workgroup of the group.  Each lane of the pair sums
        // HALF of the G partial rows (one round of loads), the halves are exchanged: y = b3 + (A + B) in both lanes, the same bits.
        float s3[3] = {0.0f, 0.0f, 0.0f};
        float gl[10];
        f16x8 hT[8];                                            // ... and its H1^T operand rows
        f16x8 zr[16];                                           // B2's operands: this wave's rows of ALL of dZ2 (requested inside G2, behind barrier 3)
        f16x8 wc[4];                                            // this thread's share of W2's column block (gathered behind barrier 2: see tile1_run)
        {
            float y[12];
            {
                f32x4 part[5][3];                               // four partial rows (groups 4 h .. 4 h + 3) + the bias row
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < 3; ++v) part[q][v] = xld4f(xr, vYp + 512u * (uint32_t)v, o_yp + ((uint32_t)q * 4u + wu) * 2048u);
#pragma unroll
                for (int v = 0; v < 3; ++v) part[4][v] = xld4f(xr, 16u * (uint32_t)v, o_b3x);
#pragma unroll
                for (int i = 0; i < 4; ++i) wc[i] = xld16(xr, vSt, o_w2tx + (par ^ 1u) * 131072u + g * 16384u + wu * 1024u + 4096u * (uint32_t)i);      // vector tid + 256 i of this workgroup's 16 KB
                __builtin_amdgcn_sched_barrier(0);                  // (the requests above are issued HERE, all of them: the scheduler would otherwise sink each to its first use)
                float half_[12];
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) half_[4 * v + e] = ((part[0][v][e] + part[1][v][e]) + part[2][v][e]) + part[3][v][e];
#pragma unroll
                for (int o = 0; o < 12; ++o) {
                    const float other = __shfl_xor(half_[o], 32, 64);
                    const float lo_ = h ? other : half_[o], hi_ = h ? half_[o] : other;      // (groups 0..3) + (groups 4..7): the same operand order in both lanes
                    y[o] = part[4][o >> 2][o & 3] + (lo_ + hi_);
                }
            }
            // tile 0's H1^T operand rows for G2: requested here, two phases early
#pragma unroll
            for (int s = 0; s < 8; ++s) hT[s] = xld16(xr, vSt + 1024u * (uint32_t)(s & 3), s_h1tx + wu * 16384u + 4096u * (uint32_t)(s >> 2));    // (H1^T has been complete since barrier 1)
            if (!last) request_obs();                           // the NEXT step's observation rows (src was advanced behind barrier 1); behind this
                                                                // phase's own requests, ~2 us ahead of the next wait on the memory counter
            __builtin_amdgcn_sched_barrier(0);                  // (the requests above are issued HERE, all of them: the scheduler would otherwise sink each to its first use)
            Q1PL_STAMP(10);                                     // (loss: the outputs summed)
#pragma unroll
            for (int o = 0; o < 10; ++o) gl[o] = 0.0f;
            if (ni == 0) {
                const PpoSample in{in_kb, in_a, in_b, in_c};
                const PpoSums ps = ppo_policy_grad<true, true, true>(a.p, y, oldrow, in, a.clip, a.ent_coeff, klc, net.inv_b, gl, 10, h);
                s3[0] = ps.ent; s3[1] = ps.kl; s3[2] = -ps.surr;
            } else {
                float vf;
                const float dvf = ppo_value_grad(y[0], in_a, in_b, a.vf_clip, vf);
                gl[0] = a.vf_coeff * dvf * net.inv_b;
                s3[0] = vf;
            }
            Q1PL_STAMP(11);                                     // (loss: differentiated)
            _Float16 row16[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) row16[o] = (_Float16)0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {                       // (quads of outputs: sat16x4; slots beyond OUT are zeros - they change neither statistic)
                if (4 * k < OUT) {
                    float g4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) g4[j] = (4 * k + j < OUT && 4 * k + j < 10) ? gl[4 * k + j] : 0.0f;
                    sat16x4(g4, amax, nsat);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int o = 4 * k + j;
                        if (o < OUT && o < 10) {
                            row16[o] = (_Float16)g4[j];
                            if (!h) *reinterpret_cast<_Float16*>(lds + L_DYT + (uint32_t)o * LD_B + 2u * bsm) = row16[o];      // (the pair computed the same row: lane h = 0 stores it)
                        }
                    }
                }
            }
            if (!h) {
                *reinterpret_cast<f16x8*>(lds + L_DY + bsm * LD_16) = *reinterpret_cast<const f16x8*>(row16);
                *reinterpret_cast<f16x8*>(lds + L_DY + bsm * LD_16 + 16u) = *reinterpret_cast<const f16x8*>(row16 + 8);
            } else {
                s3[0] = 0.0f; s3[1] = 0.0f; s3[2] = 0.0f;
            }
        }
        Q1PL_STAMP(12);                                         // (loss: rows converted + stored)
        // the sums over the samples (db3, the three statistics) are NOT formed here: db3 is a column of one more matrix product (dY^T times
        // the ones row of [x | 1]^T) and the statistics are summed from LDS, both by workgroup 0's otherwise idle wave 3 during G2
        if (!h) { statbuf[bsm] = s3[0]; statbuf[MB + bsm] = s3[1]; statbuf[2 * MB + bsm] = s3[2]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f16x8*>(lds + L_W2COL + lCol + 128u * (uint32_t)i) = wc[i];
        __syncthreads();
        Q1PL_STAMP(4);                                          // outputs + loss gradient + sums

        // ------------------------------------------------------------ B3: dZ2 of the owned units, both orientations
        {
            const f16x8 aT = lds16(lds, L_W3T + l16);
            const f16x8 bY = lds16(lds, L_DY + lS16);
            const f32x16 dA = mm(aT, bY, zero16);               // [u][b]: lane = sample (h2A's layout)
            float sb = 0.0f;
            uint64_t pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zA[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) zA[j] = dA[4 * q + j] * (1.0f - h2A[4 * q + j] * h2A[4 * q + j]);
                sat16x4(zA, amax, nsat);
                pk[q] = pack4(zA[0], zA[1], zA[2], zA[3]);
                xpub8(xr, pFrag + fq(q), o_dz2x + (wu * 16u + 2u * g) * 1024u, pk[q], loc);
            }
            const f32x16 zT = transpose32(pk, tro);             // [b][u]: lane = unit, registers = samples (the float16 values)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sb += zT[4 * q + j];
                *reinterpret_cast<uint64_t*>(lds + L_DZ2T + lwB + 16u * (uint32_t)q) = pack4(zT[4 * q], zT[4 * q + 1], zT[4 * q + 2], zT[4 * q + 3]);
            }
            sb += __shfl_xor(sb, 32, 64);
            if (h == 0) red2[w * 32u + c] = sb;
        }
        bar_arrive(net.bar, loc);                                    // (its workgroup barrier also orders dZ2T / H2T / dYT / red2 for the products below)
        Q1PL_STAMP(5);                                          // B3 + arrive 3

        // ------------------------------------------------------------ G2: dW2 rows U (two 32-input tiles per wave) + Adam + new images
        // The optimizer state (masters, both moments: torch layouts in global memory, private to the owning lane - lane (c, h) of wave w owns
        // inputs k = 64 w + 32 t + c of units U0 + row(r, h), the accumulator layout of the product) is REQUESTED before the matrix
        // products and stored after all of it has been used: every access of one kind is issued together (the compiler must assume that
        // a store may alias a later load of another array, and would otherwise serialise 32 round trips to L2 per step).
        {
            // Two 32-input tiles, software-pipelined by hand: tile 0's state + operands, its matrix products; THEN tile 1's state is requested
            // (in flight under tile 0's optimizer arithmetic), tile 0 is stored, tile 1's operands are requested - one wait covers them, tile 1's
            // state and tile 0's store acknowledgements.  (Everything of both tiles at once was measured: 96 + 64 live registers spill inside the
            // loop, 20.4 -> 23.0 us per step.)
            Q1PL_STAMP(14);                                     // (G2 entered: tile 0's state + operands were requested in the loss phase)
            f32x16 acc = zero16;                                // [u][k]: lane = input k, registers = owned units
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = mm(lds16(lds, L_DZ2T + lB + 32u * (uint32_t)s), hT[s], acc);
            // barrier 3 is polled from HERE: the reading is requested before tile 0's optimizer arithmetic (no store of this wave is in flight:
            // the counter retires behind nothing but loads) and looked at after it - every workgroup arrived right behind its dZ2, ~1 us ago -,
            // then B2's operands (all of dZ2: 16 vectors per lane) are requested and travel under tile 0's stores and all of tile 1
            // (two readings, one before and one in the middle of the arithmetic: the counter is ~1 us away, the later reading is the one that
            //  usually shows all eight tickets and it has come back by the time the arithmetic ends)
            const uint32_t seen3a = tid == 0 ? poll(net.bar, loc) : 0u;
            adam_tile(0, 0, 2, acc, wA, mA, vA, lr_bc1, rs_bc2, last);
            const uint32_t seen3b = tid == 0 ? poll(net.bar, loc) : 0u;
            adam_tile(0, 2, 4, acc, wA, mA, vA, lr_bc1, rs_bc2, last);
            Q1PL_STAMP(15);                                     // (G2: tile 0's products + optimizer)
            if (!bar_wait(net.bar, (uint32_t)G * ++bar_n, loc, a.status, 2u, (uint32_t)step, a.timeout_ticks, s_ok, seen3a > seen3b ? seen3a : seen3b)) return;
            Q1PL_STAMP(7);                                      // (barrier 3 wait)
#pragma unroll
            for (int s = 0; s < 16; ++s) zr[s] = xld16(xr, vSt + 1024u * (uint32_t)(s & 3), o_dz2x + wu * 16384u + 4096u * (uint32_t)(s >> 2));
            __builtin_amdgcn_sched_barrier(0);                  // (the requests above are issued HERE, all of them: the scheduler would otherwise sink each to its first use)
            store_tile(0, par, wA, mA, vA, false);
            Q1PL_STAMP(16);                                     // (B2's requests + tile 0's stores issued)
        }

        // ------------------------------------------------------------ B2: dH1 of the owned units from all of dZ2, dZ1, then dW1 / db1, db3
        {
            f32x16 acc = zero16;                                // [b][j]: lane = owned unit j, registers = samples (h1B's layout)
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = mm(zr[s], lds16(lds, L_W2COL + lW + 32u * (uint32_t)s), acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float z[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = acc[4 * q + j] * (1.0f - h1B[4 * q + j] * h1B[4 * q + j]);
                sat16x4(z, amax, nsat);
                *reinterpret_cast<uint64_t*>(lds + L_DZ1T + lwB + 16u * (uint32_t)q) = pack4(z[0], z[1], z[2], z[3]);
            }
        }
        __syncthreads();
        if (wu == 2u) {                                         // dW1[U] / db1[U]: [i'][u]: lane = owned unit, registers = rows of [x hi | 1 | x lo]^T
            // h = 0: registers 0..3 = inputs 0..3 (hi), 4..7 = the same inputs' lo rows;  h = 1: 0, 1 = inputs 4, 5 (hi), 2 = the ones row, 4, 5 = lo of 4, 5
            const size_t u = U0 + c;
            const int base = h ? 4 : 0;
            float w1v[4], m1v[4], v1v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool isw = h == 0u || j < 2;
                const uint32_t i1 = isw ? c * 6u + (uint32_t)(base + j) : c * 6u;
                w1v[j] = sW1[i1]; m1v[j] = sW1[192 + i1]; v1v[j] = sW1[384 + i1];
            }
            float b1v = sB1[c], mb1 = sB1[32 + c], vb1 = sB1[64 + c];
            f32x16 acc = zero16;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                acc = mm(lds16(lds, L_XT + lB + 32u * (uint32_t)s), lds16(lds, L_DZ1T + lB + 32u * (uint32_t)s), acc);
            _Float16* row = reinterpret_cast<_Float16*>(lds + L_W1 + c * LD_16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (h == 0u || j < 2) {
                    const float gr = (acc[j] + acc[4 + j]) * net.inv_scale;
                    w1v[j] = adam1(w1v[j], gr, m1v[j], v1v[j], a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                    const _Float16 wv = (_Float16)(c2 * w1v[j]);
                    row[base + j] = wv; row[8 + base + j] = wv;
                    if (last) cold()->net[NI].gw1[u * 6 + (size_t)(base + j)] = gr;
                }
            }
            if (h) {
                const float gr = acc[2] * net.inv_scale;
                b1v = adam1(b1v, gr, mb1, vb1, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                const float bs = c2 * b1v;
                const _Float16 bhi = (_Float16)bs;
                row[6] = bhi; row[14] = (_Float16)(bs - (float)bhi);
                if (last) cold()->net[NI].gb1[u] = gr;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (h == 0u || j < 2) { const uint32_t i1 = c * 6u + (uint32_t)(base + j); sW1[i1] = w1v[j]; sW1[192 + i1] = m1v[j]; sW1[384 + i1] = v1v[j]; }
            if (h) { sB1[c] = b1v; sB1[32 + c] = mb1; sB1[64 + c] = vb1; }
        }
        __syncthreads();
        Q1PL_STAMP(8);                                          // B2 + dW1 + end of step
    }
#undef Q1PL_STAMP
    step_bases();
    {                                                           // the last step's deferred work
        const uint32_t parl = (uint32_t)((a.steps - 1) & 1);
        tile1_request(parl);
        tile1_run(parl, lr_prev, rs_prev, true);
        store_tile(1, parl, wB, mB, vB, true);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { xst4(xr, vSt + 1024u * (uint32_t)q, s_st(0, 0), wA + 4 * q); xst4(xr, vSt + 1024u * (uint32_t)q, s_st(1, 0), mA + 4 * q); xst4(xr, vSt + 1024u * (uint32_t)q, s_st(2, 0), vA + 4 * q); }
    grads_b2(true, lr_prev, rs_prev);
    grads_w3b3(true, lr_prev, rs_prev);
    grads_stats();
    __syncthreads();
    if constexpr (PROF) {
        if (profiling)
            for (int k = 0; k < 20; ++k) cold()->prof[k] = pacc[k];
    }

    // ---------------------------------------------------------------- epilogue: the W2 slice's optimizer state back to its torch layouts; counters
    small_state(false);
    float* const e_w2 = cold()->net[NI].w2; float* const e_m = cold()->net[NI].m; float* const e_v = cold()->net[NI].v;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float w4[4], m4[4], v4[4];
            xld4(xr, vSt + 1024u * (uint32_t)q, s_st(0, t), w4); xld4(xr, vSt + 1024u * (uint32_t)q, s_st(1, t), m4); xld4(xr, vSt + 1024u * (uint32_t)q, s_st(2, t), v4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t e = (size_t)(U0 + rrow(4 * q + j, h)) * HID + 64u * w + 32u * (uint32_t)t + c;
                e_w2[e] = w4[j]; e_m[e] = m4[j]; e_v[e] = v4[j];
            }
        }
    if (g == 0 && tid == 192u) {                                // (wave 3's lane 0 kept the running statistics)
        // the exchange mode this group ran in: 1 + the XCD its eight workgroups share, or 0 = agent scope (the census line is still what it was)
        cold()->status[2 + ni] = loc ? __hip_atomic_load(cold()->net[NI].bar + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (ni == 0) {
            float* const sa = cold()->stats_acc;
            sa[0] += st_acc[0]; sa[1] += st_acc[1]; sa[2] += st_acc[2];
            *cold()->step_count = step0 + cold()->steps;
        } else {
            cold()->stats_acc[4] += st_acc[0];
        }
    }
    uint32_t* const satp = cold()->saturation;
    if (satp) {
        if (nsat) atomicAdd(satp + 2u * ni, nsat);
        if (amax > 0.0f) atomicMax(satp + 2u * ni + 1u, __float_as_uint(amax));
    }
}

template <bool PROF>
__global__ void __launch_bounds__(256, 1)
persistent_learner_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // (placement: see persistent_learner_body)
    const uint32_t role = blockIdx.x & 7u;
    if (role == 0u) persistent_learner_body<0, PROF>(a, lds);
    else if (role == 1u) persistent_learner_body<1, PROF>(a, lds);
}

}  // namespace q1pl
