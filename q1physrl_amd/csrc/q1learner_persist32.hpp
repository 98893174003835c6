// q1learner_persist32.hpp - the persistent PPO learner in FLOAT32 arithmetic (q1env_learner_sgd_epochs_f32): the control the float16-operand
// kernel of q1learner_persist.hpp is measured against.  RLlib / TF PPO - what the reference trains with
// (q1physrl/train.py:60-64, data/params.yml:4-13) - is float32 end to end and clips no gradient (grad_clip = None); the float16 kernel
// rounds every matrix operand to 11 bits and SATURATES per-sample gradients at +-65 504 / loss scale, which late in a run clips ~5e4
// elements per update.  This kernel computes the same 128-sample SGD steps - same decomposition (2 groups x 8 workgroups, workgroup g owns
// hidden units [32 g, 32 g + 32) of both hidden layers), same exchange protocol and barriers (q1pl::bar_arrive / bar_wait, both exchange
// modes), same loss code (q1ppo_loss.hpp), torch.optim.Adam's update - with float32 operands on v_mfma_f32_32x32x2_f32, float32 exchange
// buffers, float32 activations and optimizer, NO loss scale and NO saturation: gradients as close to float64 autograd's as float32 torch
// autograd's are (tests/test_hip_learner.py: value network 2e-7 relative per tensor, policy network 2 - 4e-4 - the float32 conditioning of the
// loss itself; the float16 kernel: 3e-4 .. 9e-4).
//
// It is written for clarity, not for the last microsecond: no deferred work, no software pipelining across barriers - every phase requests
// its operands, waits, computes.  A 32x32x2 float32 matrix instruction does 1/8 of the work of the 32x32x16 float16 one in twice the
// cycles, so a step is bound by ~440 matrix instructions x 64 cycles per wave (~13 us) + the three barriers + five exposed operand
// latencies: measured in profiles/r6_learner_f32.txt.
//
// Operand layout.  D[i][n] += sum_k A[i][k] B[n][k], lane (c, h) supplying A[c][h] and B[c][h] per instruction.  Both operands are kept
// "K-contiguous": lane (c, h) reads 16 bytes = four consecutive k at row c, k = 8 s + 4 h (a FRAGMENT, K-step s), and element e of the
// fragment feeds instruction e - which pairs k = 8 s + e (h = 0) with k = 8 s + 4 + e (h = 1) in BOTH operands: a permutation of the sum,
// nothing else.  The accumulator of lane (c, h) holds column n = c, rows i = (r & 3) + 8 (r >> 2) + 4 h: register quad q = rows 8 q + 4 h ..
// + 3, which IS fragment s = q of a [n][i] operand - activations go from accumulator to the next product's B operand, and to the exchange
// buffers in their consumer's fragment order, without a shuffle (as in the float16 kernel; csrc/q1learner_persist.hpp "Exchange layouts").
#pragma once
#include "q1learner_persist.hpp"

namespace q1pl32 {

using q1pl::Args;
using q1pl::Net;
using q1pl::f32x4;
using q1pl::G;
using q1pl::HID;
using q1pl::MB;
using q1pl::rrow;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t SW = 260;                                     // floats per row of a 256-wide LDS array (+ 4 pad)
constexpr uint32_t SB = 132;                                     // ... of a 128-wide (sample-contiguous) one
// LDS map (bytes)
constexpr uint32_t L_W2OWN = 0;                                  // [32 u][SW]   W2[U0 + u][k]                      (P2: A)
constexpr uint32_t L_W2COL = L_W2OWN + 32 * SW * 4;              // [32 j][SW]   W2[k][U0 + j]                      (B2: A)
constexpr uint32_t L_A1 = L_W2COL + 32 * SW * 4;                 // [32][SB]     H1^T staging (P1)  /  dZ1^T (B2 .. dW1)
constexpr uint32_t L_A2 = L_A1 + 32 * SB * 4;                    // [32][SB]     H2^T (P2 .. dW3)   /  dZ2^T (B3 .. G2)
constexpr uint32_t L_DYT = L_A2 + 32 * SB * 4;                   // [16 o][SB]   d loss / d output, transposed
constexpr uint32_t L_DY = L_DYT + 16 * SB * 4;                   // [128 b][20]  ... sample-major (16 outputs + pad)
constexpr uint32_t L_XH = L_DY + MB * 20 * 4;                    // [128 b][12]  x0 .. x5, 1, 0 (+ pad)
constexpr uint32_t L_XT = L_XH + MB * 12 * 4;                    // [8 i][SB]    the same, transposed
constexpr uint32_t L_W1 = L_XT + 8 * SB * 4;                     // [32 u][12]   W1[U0 + u][0 .. 5], b1[U0 + u], 0
constexpr uint32_t L_W3 = L_W1 + 32 * 12 * 4;                    // [16 o][36]   W3[o][U0 + u]
constexpr uint32_t L_W3T = L_W3 + 16 * 36 * 4;                   // [32 u][20]   W3[o][U0 + u], unit-major
constexpr uint32_t L_RED = L_W3T + 32 * 20 * 4;                  // floats: redW1[4][32][8] | redW3[4][16][32] | b2[32] | stat[3][128] | flags
constexpr uint32_t L_ST = L_RED + 16384;                         // the small parameters' optimizer state (as in the float16 kernel)
constexpr uint32_t LDS_BYTES = L_ST + 8192;                      // 160 384
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup's LDS");

// exchange buffers (bytes): H1 / H1^T per parity, dZ2 - twice the float16 kernel's (q1env_plearner.hip carves for these)
constexpr uint32_t XB_ACT = MB * HID * 4;                        // 131 072

__device__ __forceinline__ f32x4 ldsf4(const unsigned char* lds, uint32_t byte_off) { return *reinterpret_cast<const f32x4*>(lds + byte_off); }
__device__ __forceinline__ f32x16 mm32(f32x4 a, f32x4 b, f32x16 acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    return acc;
}
// ... into TWO accumulators alternately (their sum is the product): a matrix instruction that accumulates into the result of the one right before
// it waits for that result; with one wave per SIMD nothing else fills the gap
__device__ __forceinline__ void mm32x2(f32x4 a, f32x4 b, f32x16& acc0, f32x16& acc1) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc1, 0, 0, 0);
}
// A product over 32 K-steps whose B fragments come from an exchange buffer: streamed in chunks of 8 with TWO chunks in flight - chunk i + 2 is
// requested as soon as chunk i's matrix instructions have been issued (~1 us of matrix-pipe time: the L2 round trip), so that only the first
// chunk's latency is exposed.  lda(s) / ldb(s): the A fragment (LDS) / the B fragment (exchange load) of K-step s.
template <class LA, class LB>
__device__ __forceinline__ f32x16 product32(LA lda, LB ldb) {
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 b[2][8];
#pragma unroll
    for (int s = 0; s < 8; ++s) b[0][s] = ldb(s);
#pragma unroll
    for (int s = 0; s < 8; ++s) b[1][s] = ldb(8 + s);
    __builtin_amdgcn_sched_barrier(0);                          // (all sixteen requests are issued HERE: the scheduler would otherwise sink each to its first use)
    f32x16 acc0 = zero16, acc1 = zero16;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
        for (int s = 0; s < 8; ++s) mm32x2(lda(8 * ch + s), b[ch & 1][s], acc0, acc1);
        if (ch + 2 < 4) {
#pragma unroll
            for (int s = 0; s < 8; ++s) b[ch & 1][s] = ldb(8 * (ch + 2) + s);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc0 + acc1;
}
// torch.optim.Adam's update in float32 (q1learner.hpp adam_update): inv_sqrt_bc2 = 1 / sqrt(bias_correction2), step = lr / bias_correction1.  Square root
// and reciprocal are the hardware's (v_sqrt_f32 / v_rcp_f32, 1 ulp): the IEEE expansions cost ~25 instructions more per element, 8 448 elements per
// workgroup and step (measured: 33.3 -> 29 us per step), for a relative 1e-7 of a step that is lr x O(1).
__device__ __forceinline__ float adam32(float w, float g, float& m, float& v, float b1, float b2, float eps, float step, float inv_sqrt_bc2) {
    m = m + (g - m) * (1.0f - b1);
    v = v * b2 + ((1.0f - b2) * g) * g;
    const float denom = __builtin_amdgcn_sqrtf(v) * inv_sqrt_bc2 + eps;
    return w - step * (m * __builtin_amdgcn_rcpf(denom));
}
// tanh in float32: 1 - 2 / (exp(2 x) + 1) on the hardware's exponential and reciprocal - absolute error <= 2e-7 (float32 rounding of a value in
// [-1, 1]); the library's tanhf is ~10 x the instructions, 32 activations per lane and step
__device__ __forceinline__ float tanh32(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.885390081777927f * x) + 1.0f); }
// one dword of a published array (W2's master rows: read by the other workgroups' column gathers)
__device__ __forceinline__ void xpub4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float v, bool loc) {
    if (loc) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 17);
}

template <int NI, bool PROF>
__device__ __forceinline__ void body(const Args& a, unsigned char* lds) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, c = lane & 31u, h = lane >> 5;
    constexpr uint32_t ni = (uint32_t)NI;
    constexpr int OUT = NI == 0 ? 10 : 1;
    const uint32_t g = blockIdx.x >> 3;
    const Net net = a.net[NI];
    const uint32_t U0 = 32u * g;
    const uint32_t bsm = 32u * w + c;                           // the sample this lane pair stages and differentiates
    float* const red = reinterpret_cast<float*>(lds + L_RED);
    float* const redW1 = red;                                   // [4 w][32 u][8 i]
    float* const redW3 = red + 1024;                            // [4 w][16 o][32 u]
    float* const b2s = red + 3072;                              // [32]
    float* const statbuf = red + 3104;                          // [3][128]
    int* const s_ok = reinterpret_cast<int*>(red + 3488);
    float* const sW1 = reinterpret_cast<float*>(lds + L_ST);    // [kind][u * 6 + i]       (kind: master, m, v)
    float* const sB1 = sW1 + 3 * 192;
    float* const sB2 = sB1 + 3 * 32;
    float* const sW3 = sB2 + 3 * 32;                            // [kind][o * 32 + u]
    float* const sB3 = sW3 + 3 * 320;
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const size_t E_B2 = 65536, E_W1 = 65536 + 256, E_B1 = E_W1 + 1536, E_W3 = E_B1 + 256, E_B3 = E_W3 + (size_t)OUT * 256;      // (q1learner.hpp AdamNet)
    // exchange: one buffer resource over the group's workspace, one over W2's master (256 KB: rows are published by their owners)
    const __amdgpu_buffer_rsrc_t xr = q1pl::xrsrc(net.xbase);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(net.w2, 0, HID * HID * 4, 0x00020000);
    auto xoff = [&](const void* q) { return (uint32_t)(reinterpret_cast<const char*>(q) - net.xbase); };
    const uint32_t o_h1x = xoff(net.h1x), o_h1tx = xoff(net.h1tx), o_dz2x = xoff(net.dz2x), o_yp = xoff(net.yp), o_b3x = xoff(net.b3x);

    // ---------------------------------------------------------------- prologue
    for (uint32_t off = tid * 16u; off < L_ST; off += 256u * 16u) *reinterpret_cast<uint4*>(lds + off) = uint4{0, 0, 0, 0};
    __syncthreads();
    auto small_state = [&](bool to_lds) {                      // LDS <-> the torch layouts (masters, moments) of the small owned parameters
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
    // the W2 slice's optimizer state lives in registers for the launch: lane (c, h) of wave w owns inputs k = 64 w + 32 t + c of units U0 + rrow(r, h)
    float w2v[2][16], m2v[2][16], v2v[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t e = (size_t)(U0 + rrow(r, h)) * HID + 64u * w + 32u * (uint32_t)t + c;
            w2v[t][r] = net.w2[e]; m2v[t][r] = net.m[e]; v2v[t][r] = net.v[e];
        }
    {
        const uint32_t u = tid >> 3, k0 = (tid & 7u) * 32u;     // the owned rows' LDS image: 8 threads per unit, 32 inputs each
        const float* src = net.w2 + (size_t)(U0 + u) * HID + k0;
        for (uint32_t k = 0; k < 32u; ++k) *reinterpret_cast<float*>(lds + L_W2OWN + (u * SW + k0 + k) * 4u) = src[k];
    }
    __syncthreads();                                            // (sW1 .. sB3 are complete)
    auto images_small = [&]() {                                // the operand images of the small owned parameters from their LDS masters
        if (tid < 32u) {
            float* row = reinterpret_cast<float*>(lds + L_W1 + tid * 48u);
            for (int i = 0; i < 6; ++i) row[i] = sW1[tid * 6 + i];
            row[6] = sB1[tid]; row[7] = 0.0f;
            b2s[tid] = sB2[tid];
        }
        for (uint32_t e = tid; e < (uint32_t)OUT * 32u; e += 256u) {
            const uint32_t o = e >> 5, u = e & 31u;
            *reinterpret_cast<float*>(lds + L_W3 + (o * 36u + u) * 4u) = sW3[e];
            *reinterpret_cast<float*>(lds + L_W3T + (u * 20u + o) * 4u) = sW3[e];
        }
    };
    images_small();
    if (g == 0 && tid < 16u) __hip_atomic_store(net.b3x + tid, (int)tid < OUT ? net.b3[tid] : 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool loc = false;
    {                                                           // the census barrier (q1learner_persist.hpp "placement")
        if (tid == 0) __hip_atomic_store(net.bar + 16 + g, q1pl::xcc_id() + 1u + (a.census_skew ? 16u * g : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q1pl::bar_arrive(net.bar + 8, false);
        if (!q1pl::bar_wait(net.bar + 8, (uint32_t)G, false, a.status, 3u, 0u, a.timeout_ticks, s_ok)) return;
        uint32_t same = 1u;
        const uint32_t mine = __hip_atomic_load(net.bar + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t k = 1; k < (uint32_t)G; ++k) same &= __hip_atomic_load(net.bar + 16 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine ? 1u : 0u;
        loc = a.allow_local != 0 && same != 0u && mine != 0u;
    }
    const float klc = *a.klc_dev;
    const long long step0 = *a.step0_snap;
    double pw1 = pow((double)a.beta1, (double)step0), pw2 = pow((double)a.beta2, (double)step0);
    float st_acc[3] = {0.0f, 0.0f, 0.0f};
    uint32_t bar_n = 0;
    const float inv_mb = 1.0f / (float)MB;
    // The rows of a step - its index, its observation, the loss's per-sample inputs - come from HBM at a random row: two dependent latencies
    // (~3 us) that would open every step.  They are fetched AHEAD: the index of step n + 2 and the row data of step n + 1 are requested
    // behind barrier 1 of step n and have arrived long before they are used.
    int64_t win = 0, in_epoch = 0;
    auto next_src = [&]() -> int64_t {                          // the row of the NEXT window in the schedule (advances the window)
        int64_t at = win + (int64_t)bsm;
#ifdef Q1_CHECK
        if ((uint64_t)at >= (uint64_t)(a.idx ? a.idx_rows : a.rows)) { q1pl::chk_fail(q1pl::CHK_SCHED, (uint32_t)at); at = 0; }
#endif
        int64_t src = a.idx ? a.idx[at] : at;
#ifdef Q1_CHECK
        src = q1pl::chk_row(src, a.rows);
#endif
        if (++in_epoch == a.spe) { in_epoch = 0; win += a.epoch_stride - (a.spe - 1) * MB; } else { win += MB; }
        return src;
    };
    struct Rows { float ox[6]; uint32_t kb; float fa, fb, fc; float old[10]; };
    auto fetch_rows = [&](int64_t src) -> Rows {
        Rows r;
        const size_t sl = (size_t)src;
#pragma unroll
        for (int i = 0; i < 6; ++i) r.ox[i] = a.obs[sl * 6 + (size_t)i];
        r.kb = 0u; r.fa = 0.0f; r.fb = 0.0f; r.fc = 0.0f;
        if (ni == 0) { r.kb = (uint32_t)a.keys[sl]; r.fa = a.mouse_u[sl]; r.fb = a.logp_old[sl]; r.fc = a.adv[sl]; }
        else { r.fa = a.value_old[sl]; r.fb = a.vtarg[sl]; }
#pragma unroll
        for (int o = 0; o < 10; ++o) r.old[o] = ni == 0 ? a.old_logits[sl * (size_t)a.old_stride + (size_t)o] : 0.0f;
        return r;
    };
    // optional phase clock (q1env_learner_set_profiling): 10-ns ticks of lane 0 of wave (prof_g >> 3) of workgroup (prof_g & 7) of the policy group
    // (a WAVE-uniform condition - every lane of the stamped wave reads the clock: a scalar branch, no one-lane regions inside the step loop -, and its
    //  own instantiation: the product kernel carries neither the registers nor the clock reads)
    const bool profiling = PROF && a.prof != nullptr && blockIdx.x == 8u * (uint32_t)(a.prof_g & 7) &&
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)w) == (uint32_t)(a.prof_g >> 3);
    unsigned long long pacc[PROF ? 12 : 1] = {};
    uint64_t tprev = profiling ? wall_clock64() : 0;
#define Q1PL32_STAMP(k) do { if constexpr (PROF) { if (profiling) { const uint64_t now_ = wall_clock64(); pacc[k] += now_ - tprev; tprev = now_; } } } while (0)
    Rows rows_next = fetch_rows(next_src());                    // step 0's
    int64_t src_next = a.steps > 1 ? next_src() : 0;            // step 1's index
    __syncthreads();

    // Lane-dependent address parts are made OPAQUE once per step (an empty assembly statement "modifies" them), as in the float16 kernel: otherwise every
    // `base + constant` below is a loop invariant, gets hoisted out of the step loop and keeps a register for the whole launch - ~250 of them here, i.e.
    // 900 spilled registers and 42 us per step (measured) - ; as values of the current iteration each base is ONE register and the constants fold into the
    // instructions' offset fields.  (LDS offset fields hold 16 bits: arrays above 64 KB are addressed from bases that already contain L_A1.)
#define Q1PL32_OPAQUE(x) asm volatile("" : "+v"(x))
    const uint32_t wu = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);       // the wave index as the scalar it is (scalar parts of exchange offsets)
    for (int64_t step = 0; step < a.steps; ++step) {
        const uint32_t par = (uint32_t)(step & 1);
        const bool last = step + 1 == a.steps;
        uint32_t vX = lane * 16u;                               // this lane's 16 bytes of a 64-lane exchange fragment
        uint32_t lW = c * (SW * 4u) + 16u * h;                  // row c of a [32][SW] array, K offset 4 h           (+ L_W2OWN / L_W2COL + 32 s)
        uint32_t lB = L_A1 + c * (SB * 4u) + 16u * h;           // row c of A1 (A2: + L_A2 - L_A1), K offset 4 h     (+ 32 s)
        uint32_t lD = L_A1 + (c & 15u) * (SB * 4u) + 16u * h;   // row c & 15 of dY^T (+ L_DYT - L_A1)
        uint32_t lXt = L_A1 + (c & 7u) * (SB * 4u) + 16u * h;   // row c & 7 of X^T (+ L_XT - L_A1)
        uint32_t lT = L_A1 + (4u * h * SB + bsm) * 4u;          // element [rrow(r, h)][bsm] of A1 / A2: + ((r & 3) + 8 (r >> 2)) SB 4
        uint32_t lDY = L_DY + bsm * 80u + 16u * h;              // row bsm of dY, K offset 4 h
        uint32_t lXH = L_XH + bsm * 48u + 16u * h;
        uint32_t lW1 = L_W1 + c * 48u + 16u * h;
        uint32_t lW3 = L_W3 + (c & 15u) * 144u + 16u * h;
        uint32_t lW3T = L_W3T + c * 80u + 16u * h;
        uint32_t lCol = L_W2COL + tid * 4u;                     // column k = tid of W2's column block: + j SW 4
        uint32_t lOwn = L_W2OWN + (4u * h * SW + 64u * w + c) * 4u;      // element [rrow(r, h)][64 w + c (+ 32 t)] of the owned rows
        uint32_t vG = tid * (uint32_t)(HID * 4);                // row k = tid of W2's master
        uint32_t vK = (4u * h * (uint32_t)HID + 64u * w + c) * 4u;       // element [rrow(r, h)][64 w + c] of the owned rows of the master (+ U0 rows: scalar)
        uint32_t vYp = c * 16u + 512u * h, vYl = c * 16u + h * 32768u;
        Q1PL32_OPAQUE(vX); Q1PL32_OPAQUE(lW); Q1PL32_OPAQUE(lB); Q1PL32_OPAQUE(lD); Q1PL32_OPAQUE(lXt); Q1PL32_OPAQUE(lT); Q1PL32_OPAQUE(lDY); Q1PL32_OPAQUE(lXH);
        Q1PL32_OPAQUE(lW1); Q1PL32_OPAQUE(lW3); Q1PL32_OPAQUE(lW3T); Q1PL32_OPAQUE(lCol); Q1PL32_OPAQUE(lOwn); Q1PL32_OPAQUE(vG); Q1PL32_OPAQUE(vK);
        Q1PL32_OPAQUE(vYp); Q1PL32_OPAQUE(vYl);
        constexpr uint32_t D_A2 = L_A2 - L_A1, D_DYT = L_DYT - L_A1, D_XT = L_XT - L_A1;
        auto rq = [](int r) { return (uint32_t)((r & 3) + 8 * (r >> 2)); };         // rrow(r, h) - 4 h
        // exchange fragments: K-step s of a [tile][32 s][64 lanes][4] array = lane part + 1024 (s & 3) (offset field) + scalar (tile base + 4096 (s >> 2))
        const uint32_t s_h1x = o_h1x + par * XB_ACT + wu * 32768u, s_h1tx = o_h1tx + par * XB_ACT, s_dz2x = o_dz2x + wu * 32768u;
        pw1 *= (double)a.beta1;
        pw2 *= (double)a.beta2;
        const float lr_bc1 = a.lr / (float)(1.0 - pw1), rs_bc2 = 1.0f / sqrtf((float)(1.0 - pw2));
        // ------------------------------------------------------------ this step's row (fetched one step ahead): observation -> LDS, the loss's inputs stay in registers
        const Rows rows = rows_next;
        const uint32_t in_kb = rows.kb;
        const float in_a = rows.fa, in_b = rows.fb, in_c = rows.fc;
        float oldrow[10];
#pragma unroll
        for (int o = 0; o < 10; ++o) oldrow[o] = rows.old[o];
        if (h == 0u) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                *reinterpret_cast<float*>(lds + lXH + 4u * (uint32_t)i) = rows.ox[i];
                *reinterpret_cast<float*>(lds + lT + D_XT + (uint32_t)i * (SB * 4u)) = rows.ox[i];         // (h = 0: lT = L_A1 + 4 bsm)
            }
            *reinterpret_cast<float*>(lds + lXH + 24u) = 1.0f; *reinterpret_cast<float*>(lds + lXH + 28u) = 0.0f;
            *reinterpret_cast<float*>(lds + lT + D_XT + 6u * (SB * 4u)) = 1.0f;
        }
        __syncthreads();

        // ------------------------------------------------------------ P1: H1[:, U] = tanh(X W1[U]^T + b1[U]), published in both orientations
        float h1A[16];                                          // [u = rrow(r, h)][b = c]
        {
            const f32x16 acc = mm32(ldsf4(lds, lW1), ldsf4(lds, lXH), zero16);
#pragma unroll
            for (int r = 0; r < 16; ++r) h1A[r] = tanh32(acc[r]);
#pragma unroll
            for (int q = 0; q < 4; ++q)                        // quad q = units U0 + 8 q + 4 h ..: fragment s = 4 g + q of sample tile w
                q1pl::xpub16f(xr, vX + 1024u * (uint32_t)q, s_h1x + g * 4096u, h1A[4 * q], h1A[4 * q + 1], h1A[4 * q + 2], h1A[4 * q + 3], loc);
            // transposed through LDS (this wave's 32 columns of the staging tile): lane = unit, four consecutive samples per fragment
#pragma unroll
            for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(lds + lT + rq(r) * (SB * 4u)) = h1A[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 v = ldsf4(lds, lB + 128u * wu + 32u * (uint32_t)s);
                q1pl::xpub16f(xr, vX + 1024u * (uint32_t)s, s_h1tx + g * 16384u + wu * 4096u, v[0], v[1], v[2], v[3], loc);
            }
        }
        Q1PL32_STAMP(0);                                        // rows + P1 + publish
        q1pl::bar_arrive(net.bar, loc);
        if (!q1pl::bar_wait(net.bar, (uint32_t)G * ++bar_n, loc, a.status, 0u, (uint32_t)step, a.timeout_ticks, s_ok)) return;
        Q1PL32_STAMP(1);                                        // barrier 1
        if (!last) {                                            // the next step's rows, the index of the step after it
            rows_next = fetch_rows(src_next);
            if (step + 2 < a.steps) src_next = next_src();
        }

        // ------------------------------------------------------------ W2's column block (every owner's rows as of the previous step), P2, partial outputs
        float h2A[16];
        {
            f32x4 colv[8];                                      // thread k = tid: W2[k][U0 .. U0 + 31]
#pragma unroll
            for (int i = 0; i < 8; ++i) colv[i] = q1pl::xld4f(wr, vG + 16u * (uint32_t)i, U0 * 4u);
            // H1 rows of this wave's sample tile (32 K-steps) against the owned rows of W2
            const f32x16 acc = product32([&](int s) { return ldsf4(lds, L_W2OWN + lW + 32u * (uint32_t)s); },
                                         [&](int s) { return q1pl::xld4f(xr, vX + 1024u * (uint32_t)(s & 3), s_h1x + 4096u * (uint32_t)(s >> 2)); });
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) *reinterpret_cast<float*>(lds + lCol + (uint32_t)(4 * i + e) * (SW * 4u)) = colv[i][e];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(b2s + 8 * q + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) h2A[4 * q + j] = tanh32(acc[4 * q + j] + bq[j]);
            }
            // partial outputs of the owned units: the accumulator quads ARE the B fragments (K = 32 units: s = 0 .. 3)
            f32x16 accY = zero16;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                accY = mm32(ldsf4(lds, lW3 + 32u * (uint32_t)q), f32x4{h2A[4 * q], h2A[4 * q + 1], h2A[4 * q + 2], h2A[4 * q + 3]}, accY);
            q1pl::xpub16f(xr, vYp, o_yp + (g * 4u + wu) * 2048u, accY[0], accY[1], accY[2], accY[3], loc);               // outputs 4 h .. 4 h + 3
            q1pl::xpub16f(xr, vYp + 1024u, o_yp + (g * 4u + wu) * 2048u, accY[4], accY[5], accY[6], accY[7], loc);       // outputs 8 + 4 h ..
#pragma unroll
            for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(lds + lT + D_A2 + rq(r) * (SB * 4u)) = h2A[r];      // H2^T for dW3
        }
        Q1PL32_STAMP(2);                                        // column gather + P2 + partial outputs
        q1pl::bar_arrive(net.bar, loc);
        if (!q1pl::bar_wait(net.bar, (uint32_t)G * ++bar_n, loc, a.status, 1u, (uint32_t)step, a.timeout_ticks, s_ok)) return;
        Q1PL32_STAMP(3);                                        // barrier 2

        // ------------------------------------------------------------ outputs and the loss gradient (two lanes per sample; identical in all 8 workgroups)
        {
            float y[12];
            {
                f32x4 part[5][3];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < 3; ++v) part[q][v] = q1pl::xld4f(xr, vYl + 512u * (uint32_t)v, o_yp + ((uint32_t)q * 4u + wu) * 2048u);
#pragma unroll
                for (int v = 0; v < 3; ++v) part[4][v] = q1pl::xld4f(xr, 16u * (uint32_t)v, o_b3x);
#pragma unroll
                for (int o = 0; o < 12; ++o) {
                    const float half_ = ((part[0][o >> 2][o & 3] + part[1][o >> 2][o & 3]) + part[2][o >> 2][o & 3]) + part[3][o >> 2][o & 3];
                    const float other = __shfl_xor(half_, 32, 64);
                    const float lo_ = h ? other : half_, hi_ = h ? half_ : other;
                    y[o] = part[4][o >> 2][o & 3] + (lo_ + hi_);
                }
            }
            float gl[10], s3[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int o = 0; o < 10; ++o) gl[o] = 0.0f;
            if (ni == 0) {
                const PpoSample in{in_kb, in_a, in_b, in_c};
                const PpoSums ps = ppo_policy_grad<true, true, true>(a.p, y, oldrow, in, a.clip, a.ent_coeff, klc, 1.0f, gl, 10, h);
                s3[0] = ps.ent; s3[1] = ps.kl; s3[2] = -ps.surr;
            } else {
                float vf;
                gl[0] = a.vf_coeff * ppo_value_grad(y[0], in_a, in_b, a.vf_clip, vf);
                s3[0] = vf;
            }
            if (h == 0u) {
#pragma unroll
                for (int o = 0; o < 16; ++o) {
                    const float v = (o < OUT && o < 10) ? gl[o < 10 ? o : 0] : 0.0f;
                    *reinterpret_cast<float*>(lds + lDY + 4u * (uint32_t)o) = v;
                    *reinterpret_cast<float*>(lds + lT + D_DYT + (uint32_t)o * (SB * 4u)) = v;
                }
                statbuf[bsm] = s3[0]; statbuf[MB + bsm] = s3[1]; statbuf[2 * MB + bsm] = s3[2];
            }
        }
        __syncthreads();
        if (g == 0 && wu == 3u) {                               // the step's statistics
            float sv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float v = statbuf[k * MB + lane] + statbuf[k * MB + 64 + lane];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                sv[k] = v;
            }
            if (lane == 0) { st_acc[0] += sv[0] * inv_mb; st_acc[1] += sv[1] * inv_mb; st_acc[2] += sv[2] * inv_mb; }
        }
        Q1PL32_STAMP(4);                                        // outputs + loss gradient + statistics
        // dW3[:, U]: every wave sums over its own 32 samples, the four partial tiles meet in LDS
        {
            f32x16 acc = zero16;                                // [o][u]: lane = owned unit, registers = outputs
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = mm32(ldsf4(lds, lD + D_DYT + 128u * wu + 32u * (uint32_t)s), ldsf4(lds, lB + D_A2 + 128u * wu + 32u * (uint32_t)s), acc);
#pragma unroll
            for (int r = 0; r < 8; ++r) redW3[(w * 16u + rrow(r, h)) * 32u + c] = acc[r];      // (rows o = rrow(r, h) < 16: r < 8)
        }
        // ------------------------------------------------------------ B3: dZ2[:, U] = (dY W3[:, U]) (1 - H2[:, U]^2)
        float dz2A[16];
        {
            f32x16 acc = zero16;
#pragma unroll
            for (int s = 0; s < 2; ++s) acc = mm32(ldsf4(lds, lW3T + 32u * (uint32_t)s), ldsf4(lds, lDY + 32u * (uint32_t)s), acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) dz2A[r] = acc[r] * (1.0f - h2A[r] * h2A[r]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                q1pl::xpub16f(xr, vX + 1024u * (uint32_t)q, s_dz2x + g * 4096u, dz2A[4 * q], dz2A[4 * q + 1], dz2A[4 * q + 2], dz2A[4 * q + 3], loc);
        }
        __syncthreads();                                        // (H2^T has been read by every wave's dW3; the partial tiles are complete)
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(lds + lT + D_A2 + rq(r) * (SB * 4u)) = dz2A[r];        // dZ2^T for dW2 / db2
        Q1PL32_STAMP(5);                                        // dW3 partial + B3
        q1pl::bar_arrive(net.bar, loc);                          // (its workgroup barrier orders dZ2^T)

        // ------------------------------------------------------------ between the arrival at barrier 3 and its wait: the weight gradients that need
        // nothing of the others' dZ2 - dW3 / db3 / db2 (small), dW2 rows U (two 32-input tiles per wave) - and their optimizer updates
        for (uint32_t e = tid; e < (uint32_t)OUT * 32u; e += 256u) {                                    // dW3
            const uint32_t o = e >> 5, u = e & 31u;
            const float gr = (((redW3[(0u * 16u + o) * 32u + u] + redW3[(1u * 16u + o) * 32u + u]) + redW3[(2u * 16u + o) * 32u + u]) + redW3[(3u * 16u + o) * 32u + u]) * inv_mb;
            float mv = sW3[320 + e], vv = sW3[640 + e];
            const float wn = adam32(sW3[e], gr, mv, vv, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
            sW3[e] = wn; sW3[320 + e] = mv; sW3[640 + e] = vv;
            *reinterpret_cast<float*>(lds + L_W3 + (o * 36u + u) * 4u) = wn;
            *reinterpret_cast<float*>(lds + L_W3T + (u * 20u + o) * 4u) = wn;
            if (last) net.gw3[(size_t)o * HID + U0 + u] = gr;
        }
        float b3_new = 0.0f;                                    // (workgroup 0: the updated output bias, published BEHIND barrier 3 - see there)
        bool b3_mine = false;
        {                                                       // db2[U] (8 threads per unit) and, on workgroup 0, db3 (8 threads per output)
            const uint32_t u = tid >> 3, part = tid & 7u;
            float sb = 0.0f, s3o = 0.0f;
            const float* zrow = reinterpret_cast<const float*>(lds + L_A2 + (u * SB + 16u * part) * 4u);
            const float* yrow = reinterpret_cast<const float*>(lds + L_DYT + ((u & 15u) * SB + 16u * part) * 4u);
#pragma unroll
            for (int j = 0; j < 16; ++j) { sb += zrow[j]; s3o += yrow[j]; }
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) { sb += __shfl_xor(sb, off, 64); s3o += __shfl_xor(s3o, off, 64); }
            if (part == 0u) {
                const float gr = sb * inv_mb;
                float mv = sB2[32 + u], vv = sB2[64 + u];
                const float bn = adam32(sB2[u], gr, mv, vv, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                sB2[u] = bn; sB2[32 + u] = mv; sB2[64 + u] = vv;
                b2s[u] = bn;
                if (last) net.gb2[U0 + u] = gr;
                if (g == 0 && (int)u < OUT) {
                    const float g3 = s3o * inv_mb;
                    float m3 = sB3[16 + u], v3 = sB3[32 + u];
                    const float b3n = adam32(sB3[u], g3, m3, v3, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                    sB3[u] = b3n; sB3[16 + u] = m3; sB3[32 + u] = v3;
                    if (last) net.gb3[u] = g3;
                    b3_new = b3n; b3_mine = true;
                }
            }
        }
        Q1PL32_STAMP(6);                                        // arrive 3 + small gradients
        {                                                       // dW2[U, k]: lane = input k = 64 w + 32 t + c, registers = owned units; two 32-input tiles t
            // H1^T fragments of both tiles as ONE stream of 4 chunks of 8 (tile 0: chunks 0, 1; tile 1: chunks 2, 3), two chunks in flight: tile 1's
            // operands travel under tile 0's matrix instructions and optimizer arithmetic
            auto ldh = [&](int i) { return q1pl::xld4f(xr, vX + 1024u * (uint32_t)(i & 3), s_h1tx + (2u * wu + (uint32_t)(i >> 4)) * 16384u + 4096u * (uint32_t)((i & 15) >> 2)); };
            auto ldz = [&](int s) { return ldsf4(lds, lB + D_A2 + 32u * (uint32_t)s); };
            f32x4 hb[2][8];
#pragma unroll
            for (int s = 0; s < 8; ++s) hb[0][s] = ldh(s);
#pragma unroll
            for (int s = 0; s < 8; ++s) hb[1][s] = ldh(8 + s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 acc0 = zero16, acc1 = zero16;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) mm32x2(ldz(8 * ch + s), hb[ch][s], acc0, acc1);
                    if (t == 0) {
#pragma unroll
                        for (int s = 0; s < 8; ++s) hb[ch][s] = ldh(16 + 8 * ch + s);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                const f32x16 acc = acc0 + acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float gr = acc[r] * inv_mb;
                    const float wn = adam32(w2v[t][r], gr, m2v[t][r], v2v[t][r], a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                    w2v[t][r] = wn;
                    *reinterpret_cast<float*>(lds + lOwn + rq(r) * (SW * 4u) + 128u * (uint32_t)t) = wn;
                    xpub4(wr, vK + rq(r) * (uint32_t)(HID * 4) + 128u * (uint32_t)t, U0 * (uint32_t)(HID * 4), wn, loc);      // the master row: gathered by the others behind the next barrier 1
                    if (last) net.gw2[(size_t)(U0 + rrow(r, h)) * HID + 64u * w + 32u * (uint32_t)t + c] = gr;
                }
            }
        }
        Q1PL32_STAMP(7);                                        // dW2 + Adam
        if (!q1pl::bar_wait(net.bar, (uint32_t)G * ++bar_n, loc, a.status, 2u, (uint32_t)step, a.timeout_ticks, s_ok)) return;
        Q1PL32_STAMP(8);                                        // barrier 3 wait
        // the new output bias is published only NOW: every workgroup reads b3 for THIS step's loss right behind barrier 2, and a slow one may still be
        // there while workgroup 0 is already past its arrival at barrier 3 - behind the wait everybody's reads have completed (the arrival waits for them);
        // the next reading is behind the next step's barrier 2
        if (b3_mine) q1pl::pub4f(net.b3x + (tid >> 3), b3_new, loc);

        // ------------------------------------------------------------ B2: dH1[:, U] = dZ2 W2[:, U] (all of dZ2), dZ1 = dH1 (1 - H1^2), dW1 / db1
        {
            // [j][b]: lane = sample, registers = owned units (h1A's layout)
            const f32x16 acc = product32([&](int s) { return ldsf4(lds, L_W2COL + lW + 32u * (uint32_t)s); },
                                         [&](int s) { return q1pl::xld4f(xr, vX + 1024u * (uint32_t)(s & 3), s_dz2x + 4096u * (uint32_t)(s >> 2)); });
#pragma unroll
            for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(lds + lT + rq(r) * (SB * 4u)) = acc[r] * (1.0f - h1A[r] * h1A[r]);      // dZ1^T
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            f32x16 a1 = zero16;                                 // [u][i]: lane = column i of [x | 1 | 0] (c < 8), registers = owned units; this wave's 32 samples
#pragma unroll
            for (int s = 0; s < 4; ++s)
                a1 = mm32(ldsf4(lds, lB + 128u * wu + 32u * (uint32_t)s), ldsf4(lds, lXt + D_XT + 128u * wu + 32u * (uint32_t)s), a1);
            if (c < 8u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) redW1[(w * 32u + rrow(r, h)) * 8u + c] = a1[r];
            }
        }
        __syncthreads();
        {
            const uint32_t u = tid >> 3, i = tid & 7u;
            if (i < 7u) {
                const float gr = (((redW1[(0u * 32u + u) * 8u + i] + redW1[(1u * 32u + u) * 8u + i]) + redW1[(2u * 32u + u) * 8u + i]) + redW1[(3u * 32u + u) * 8u + i]) * inv_mb;
                float* row = reinterpret_cast<float*>(lds + L_W1 + u * 48u);
                if (i < 6u) {
                    const uint32_t e = u * 6u + i;
                    float mv = sW1[192 + e], vv = sW1[384 + e];
                    const float wn = adam32(sW1[e], gr, mv, vv, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                    sW1[e] = wn; sW1[192 + e] = mv; sW1[384 + e] = vv;
                    row[i] = wn;
                    if (last) net.gw1[(size_t)(U0 + u) * 6 + i] = gr;
                } else {
                    float mv = sB1[32 + u], vv = sB1[64 + u];
                    const float bn = adam32(sB1[u], gr, mv, vv, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                    sB1[u] = bn; sB1[32 + u] = mv; sB1[64 + u] = vv;
                    row[6] = bn;
                    if (last) net.gb1[U0 + u] = gr;
                }
            }
        }
        __syncthreads();
        Q1PL32_STAMP(9);                                        // B2 + dW1 + end of step
    }
#undef Q1PL32_OPAQUE
#undef Q1PL32_STAMP
    if constexpr (PROF) {
        if (profiling && lane == 0u)
            for (int k = 0; k < 12; ++k) a.prof[k] = pacc[k];
    }

    // ---------------------------------------------------------------- epilogue: state back to the torch layouts; counters
    small_state(false);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t e = (size_t)(U0 + rrow(r, h)) * HID + 64u * w + 32u * (uint32_t)t + c;
            net.w2[e] = w2v[t][r]; net.m[e] = m2v[t][r]; net.v[e] = v2v[t][r];
        }
    if (g == 0 && tid == 192u) {
        a.status[2 + ni] = loc ? __hip_atomic_load(net.bar + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (ni == 0) {
            a.stats_acc[0] += st_acc[0]; a.stats_acc[1] += st_acc[1]; a.stats_acc[2] += st_acc[2];
            *a.step_count = step0 + a.steps;
        } else {
            a.stats_acc[4] += st_acc[0];
        }
    }
}

template <bool PROF>
__global__ void __launch_bounds__(256, 1)
persistent_learner_f32_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t role = blockIdx.x & 7u;                      // (placement: q1learner_persist.hpp persistent_learner_body)
    if (role == 0u) body<0, PROF>(a, lds);
    else if (role == 1u) body<1, PROF>(a, lds);
}

}  // namespace q1pl32
