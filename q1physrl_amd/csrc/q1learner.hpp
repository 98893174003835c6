// q1learner.hpp - the native PPO learner step of libq1env (device code): forward, backward and weight gradients of the policy and the
// value network of the reference's shape (RLlib fcnet of train.py:60-64 / params.yml: obs 6 -> 256 tanh -> 256 tanh -> OUT, two
// separate networks) on the gfx950 matrix cores - float16 operands, float32 accumulation, master weights and the optimizer float32.
// Counterpart of the torch modules + autograd of q1physrl_amd/ppo.py; included by q1env_learner.hip.
//
// One SGD step on a minibatch of B samples is five launches (+ the optimizer):
//   learner_forward_kernel    gathers the minibatch's observations through idx, runs both networks exactly like the sampler's
//                             q1pol::mlp_forward_kernel (same mlp_tile: bit-identical outputs) and additionally stores tanh(H1), tanh(H2)
//                             as float16 "T-format" tiles (lane = sample, registers = hidden units: the next layer's B operands as is)
//   ppo_loss_grad_kernel      (q1env_policy.hip) d loss / d(logits, value) per sample, closed form, gathered through the same idx
//   learner_backward_kernel   dH2 = W3^T dY, dZ2 = dH2 (1 - h2^2), dH1 = W2^T dZ2, dZ1 = dH1 (1 - h1^2) per 32-sample tile with the
//                             TRANSPOSED weight images in LDS, then transposes dZ2, dZ1 (and [x | 1], dY) to "N-format" (lane = hidden unit,
//                             registers = samples) with two identity-operand MFMAs per 32x32 tile and stores them
//   learner_wgrad_kernel      dW2 = dZ2^T h1, dW3 = dY^T h2, dW1 = dZ1^T x and the three bias gradients as matrix products whose
//                             contraction runs over the SAMPLES (N-format operands, split over workgroups); partial sums per workgroup.
//                             Round 4: h1 / h2 arrive in the forward kernel's T-format and are transposed here (the matrix pipe is idle
//                             in this memory-bound kernel), so the step no longer materialises N-format copies of them: 519 -> 452 MB
//                             per 32 768-sample step, same time (profiles/r4_learner_bytes.txt)
//   learner_reduce_kernel     sums the partials, undoes the tile permutation, scales and writes the gradients in torch layout
//   learner_images_kernel     (after the optimizer) rebuilds the float16 weight images of both directions from the float32 masters
// Round 6 (large minibatches, q1env_learner_sgd_step from 2 048 samples on): learner_fwdbwd_kernel (q1learner_fused.hpp) does the first three of
// these in ONE launch, and learner_wgrad_shared_kernel (below) is the weight-gradient kernel with its h1 operands shared through LDS; the kernels
// listed here remain the path of small minibatches, of q1env_learner_step / _forward / _backward, and the reference the fused ones are tested against.
//
// Why two activation formats: a 32x32x16 MFMA contracts the index its operands hold eight-at-a-time per lane.  The forward and the
// data-gradient chain contract over hidden units (the C/D layout of one layer - lane = sample - is already the next B operand), the
// weight gradients contract over samples, so every activation is needed once with lane = sample and once with lane = unit.  The
// transposition costs two MFMAs per 32x32 tile: D = A E with E a 0/1 selection matrix delivers A^T in the C/D layout, exactly (float16
// values times 1.0, float32 accumulation).
//
// T-format (per network, per layer): f16x8[tile][t = 0..7][u = 0..1][lane 64]; element e of lane (c, h) = hidden unit
//          32 t + 16 u + (e & 3) + 8 (e >> 2) + 4 h of sample 32 tile + c (the B operand of K-step 2 t + u, see q1policy.hpp).
// N-format (per network, per array): f16x8[tile][t = 0..7][ks = 0..1][lane 64]; lane (c, h) = hidden unit 32 t + sigma(c), sigma = swap
//          bits 2 and 3; element e = sample 32 tile + (e & 3) + 16 ks + 8 (e >> 2) + 4 h.  Both MFMA operands of a weight-gradient
//          product use the same sample order, so it never has to be undone; sigma is undone by learner_reduce_kernel.
// Gradient scaling (loss scaling): d loss / d(logits, value) arrive multiplied by a per-network scale so that they AND everything derived
// from them (dZ2 = W3^T dY, dZ1 = W2^T dZ2, each times 1 - h^2) sit in float16's normal range: 256 B for the policy network (B makes the
// gradient per-sample instead of averaged; 256 because the policy head's weights start at 1e-2 x and dZ would otherwise sit at 1e-3 ..
// 1e-6, partly subnormal - which cost one training seed of five its final 400 reward units), B for the value network (per-sample value
// errors are O(1..1000) already).  learner_reduce_kernel / learner_adam_kernel divide the sums by the scale again.  Outliers (a
// probability ratio that explodes for one sample, a value error of tens of thousands) SATURATE at float16's largest finite value in
// every float32 -> float16 conversion of a gradient: an inf would become NaN in the very next transposition (inf x 0 of the selection
// operand) and from there every weight - which is what the first long training run of a second seed did before the conversions saturated.
#pragma once
#include "q1policy.hpp"
#include "q1ppo_loss.hpp"

namespace q1learn {

using q1pol::f16x8;
using q1pol::f32x16;
using q1pol::HID;
using q1pol::OBS;
using q1pol::ROW_BYTES;

constexpr uint32_t TILE_VECS = 8u * 2u * 64u;                 // f16x8 vectors of one activation array of one 32-sample tile (16 KiB)
constexpr int W3T_ROW_BYTES = 32 * 2 + 16;                    // row of the W3^T image: 32 outputs (K) + 16 B pad
constexpr size_t LDS_W2T = (size_t)HID * ROW_BYTES;           // 135168
constexpr size_t LDS_W3T = (size_t)HID * W3T_ROW_BYTES;       // 20480
constexpr size_t LDS_BWD = LDS_W2T + LDS_W3T;                 // 155648 <= 163840

// weight-gradient products of one network, 32x32 float32 tiles: [p][reg 16][lane 64]
//   p = 8 jt + kt   dW2 tile (rows: units of dZ2 tile jt, cols: units of h1 tile kt)          0 .. 63
//   p = 64 + jt     dZ2 tile jt x [x | 1]: column sigma(6) = db2                                      64 .. 71
//   p = 72 + kt     dZ1 tile kt x [x | 1]: columns 0..5 = dW1, column 6 = db1                  72 .. 79
//   p = 80 + jt     dY x h2 tile jt: rows = outputs, cols = units -> dW3                       80 .. 87
//   p = 88          dY x [x | 1]: column 6 = db3
constexpr int WG_PRODUCTS = 89;
constexpr size_t PARTIAL_FLOATS = (size_t)WG_PRODUCTS * 1024u;
constexpr size_t PARTIAL_STRIDE = PARTIAL_FLOATS;    // (round 4: 256 B of padding between the splits - their 356 KB stride is a multiple of 4 KiB - changed nothing)

__device__ __forceinline__ uint32_t sigma(uint32_t c) { return (c & ~0xCu) | ((c & 4u) << 1) | ((c & 8u) >> 1); }   // swap bits 2 and 3

__device__ __forceinline__ f16x8 cvt8(const f32x16& a, int u) {
    union { f16x8 v; q1pol::f16x2 p[4]; } o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const q1pol::f32x2 t = {a[8 * u + 2 * j], a[8 * u + 2 * j + 1]};
        o.p[j] = __builtin_convertvector(t, q1pol::f16x2);
    }
    return o.v;
}

// the same with saturation at float16's largest finite value: for the data gradients, whose magnitude is not bounded by construction -
// an inf would turn into NaN in the very next transposition (inf x 0 of the selection operand) and from there into every weight
// amax: running maximum of |value| BEFORE the clamp over everything this lane converts (one v_max3_f32 with |.| modifiers per pair) - what
// the kernel reports so that a saturation does not go unnoticed
__device__ __forceinline__ f16x8 cvt8_sat(const f32x16& a, int u, float& amax) {
    union { f16x8 v; q1pol::f16x2 p[4]; } o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        amax = fmaxf(fmaxf(fabsf(a[8 * u + 2 * j]), fabsf(a[8 * u + 2 * j + 1])), amax);
        const q1pol::f32x2 t = {__builtin_amdgcn_fmed3f(a[8 * u + 2 * j], -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(a[8 * u + 2 * j + 1], -65504.0f, 65504.0f)};
        o.p[j] = __builtin_convertvector(t, q1pol::f16x2);
    }
    return o.v;
}

// ------------------------------------------------------------------------------------------------------------------ forward
struct FwdNet {
    const float* w1; const float* b1; const uint16_t* w23; const float* b2; const float* b3;
    float* out; int out_dim;
    f16x8* h1T; f16x8* h2T;
};

// The sampler's forward kernel (q1pol::mlp_forward_kernel<512>) with a gather in front and the activation stores inside; outputs are
// bit-identical to q1env_policy_value_forward on the gathered rows.
__global__ void __launch_bounds__(512, 1)
learner_forward_kernel(int n, const float* __restrict__ obs, const int64_t* __restrict__ idx, const int64_t* __restrict__ idx_cursor, FwdNet net_a,
                       FwdNet net_b, int nets) {
    if (idx && idx_cursor) idx += *idx_cursor;         // the minibatch = rows idx[cursor .. cursor + n) (q1env_learner_batch.idx_cursor_dev)
    using namespace q1pol;
    const uint32_t bgrid = nets == 2 ? gridDim.x / 2u : gridDim.x;
    const bool second = nets == 2 && blockIdx.x >= bgrid;
    const uint32_t bid = second ? blockIdx.x - bgrid : blockIdx.x;
    const FwdNet net = second ? net_b : net_a;
    const float* __restrict__ b3 = net.b3;
    float* __restrict__ out = net.out;
    const int OUT = net.out_dim;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* l_w2 = lds;
    unsigned char* l_w3 = lds + LDS_W2;
    float* l_b2 = reinterpret_cast<float*>(lds + LDS_W2 + LDS_W3);
    unsigned char* l_w1 = lds + LDS_W2 + LDS_W3 + LDS_B2;
    const uint32_t tid = threadIdx.x;
    stage_image<512>(lds, net.w23, tid);
    if (tid < (uint32_t)HID) {
        l_b2[tid] = TANH_PRESCALE * net.b2[tid];
        stage_w1_row(l_w1, tid, net.w1, net.b1);
    }
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, half = lane >> 5;
    __syncthreads();
    const uint32_t ntiles = ((uint32_t)n + 31u) / 32u;
    const uint32_t tstride = bgrid * 8u;
    const unsigned char* w1row = l_w1 + (size_t)col * 32u + half * 16u;
    const unsigned char* wrow = l_w2 + (size_t)col * ROW_BYTES + half * 16u;
    const unsigned char* w3row = l_w3 + (size_t)col * ROW_BYTES + half * 16u;
    for (uint32_t tile = bid * 8u + wave; tile < ntiles; tile += tstride) {
        const uint32_t s = tile * 32u + col;
        const bool live = s < (uint32_t)n;
        float x[3];
        {
            const size_t src = live ? (idx ? (size_t)idx[s] : (size_t)s) : 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = live ? obs[src * OBS + 2u * k + half] : 0.0f;
        }
        const f16x8 xb = split_inputs(x, half);
        f16x8* h1 = net.h1T + (size_t)tile * TILE_VECS + lane;
        f16x8* h2 = net.h2T + (size_t)tile * TILE_VECS + lane;
        const f32x16 y = mlp_tile_t<true>(xb, w1row, wrow, w3row, l_b2, half, nullptr, h1, h2);
        if (live) {
            float* dst = out + (size_t)s * (uint32_t)OUT;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (8 * g < OUT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = r + 8 * g + 4 * (int)half;
                        if (row < OUT) dst[row] = y[4 * g + r] + b3[row];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ backward
struct BwdNet {
    const uint16_t* w2t;      // float16[256][264]: row k = W2[perm(p)][k], the forward image's column permutation applied to the K index j
    const uint16_t* w3t;      // float16[256][40]:  row j = W3[o][j], o = 0..31 (zero beyond out_dim), natural order
    const float* dy;          // float[n][dy_stride]: d loss / d output x grad_scale (dlogits rows, or dvalue with stride 1)
    int dy_stride; int out_dim;
    const f16x8* h1T; const f16x8* h2T;
    f16x8* dz2N; f16x8* dz1N;     // (round 4: h1 / h2 are no longer written in N-format - the weight-gradient kernel transposes the T-format itself)
    f16x8* xN;                // f16x8[tile][ks][lane]: the gathered observations + the constant 1 (input slot 6), lane = sigma(input index)
    f16x8* dyN;               // f16x8[tile][ks][lane]: dY, lane = output index (natural)
    uint32_t* sat;            // optional uint32[2]: += (lane, launch) pairs that converted a gradient element beyond float16's 65504 (it was
                              // clamped); max= the float32 bits of the largest |element| seen before the clamp (dY, dZ2, dZ1, scaled as they travel)
};

// saturation report of one workgroup (BwdNet::sat): the count is added when there is one, the maximum is published only when it raises the word
// (the plain read may be stale - the atomic max decides; in the steady state of a training run almost no workgroup issues an atomic at all)
__device__ __forceinline__ void sat_report(uint32_t* sat, uint32_t count, uint32_t max_bits) {
    if (count) atomicAdd(sat, count);
    if (max_bits > *reinterpret_cast<volatile uint32_t*>(sat + 1)) atomicMax(sat + 1, max_bits);
}

template <uint32_t THREADS, uint32_t NVEC>
struct StageRegs { uint4 v[(NVEC + THREADS - 1u) / THREADS]; };

// straight 16-byte copies global -> LDS with ALL of a thread's loads in flight at once, in two halves so that a caller may put
// independent work between the requests and the LDS stores that wait for them
template <uint32_t THREADS, uint32_t NVEC>
__device__ __forceinline__ void stage_issue(StageRegs<THREADS, NVEC>& r, const uint16_t* __restrict__ src_, uint32_t tid) {
    const uint4* src = reinterpret_cast<const uint4*>(src_);
    constexpr uint32_t PER = (NVEC + THREADS - 1u) / THREADS;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t c = k * THREADS + tid;
        r.v[k] = c < NVEC ? src[c] : make_uint4(0, 0, 0, 0);
    }
}
template <uint32_t THREADS, uint32_t NVEC>
__device__ __forceinline__ void stage_commit(unsigned char* dst, const StageRegs<THREADS, NVEC>& r, uint32_t tid) {
    uint4* d = reinterpret_cast<uint4*>(dst);
    constexpr uint32_t PER = (NVEC + THREADS - 1u) / THREADS;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t c = k * THREADS + tid;
        if (c < NVEC) d[c] = r.v[k];
    }
}
template <uint32_t THREADS, uint32_t NVEC>
__device__ __forceinline__ void stage_copy(unsigned char* dst, const uint16_t* __restrict__ src_, uint32_t tid) {
    StageRegs<THREADS, NVEC> r;
    stage_issue<THREADS, NVEC>(r, src_, tid);
    stage_commit<THREADS, NVEC>(dst, r, tid);
}

// The fused SGD step (q1env_learner_sgd_step, round 4): the backward kernel computes its own dY - the PPO loss gradient of the tile's
// samples (q1ppo_loss.hpp, the reference's action structure: 4 keys + continuous mouse) from the forward kernel's logits / value and
// the trajectory batch - instead of reading rows a separate loss kernel wrote (11.5 us of a 129 us step at 32 768 samples: an
// elementwise kernel that is three dependent gathers deep), while the tile's h2 vectors are in flight.  Lanes (c, 0) and (c, 1) both
// evaluate sample c (the same instructions) and each keeps the outputs its operand slots hold.  Statistics: per-lane running sums
// (half 0 only), one row per workgroup at the end: stats_rows float[gridDim.x][5] = (entropy, kl, -surrogate, total, vf) - the policy
// network's workgroups fill 0..2 and their part of the total, the value network's the rest.
struct LossArgs {
    Params p;
    const float* logits; const float* value;           // the forward kernel's outputs: minibatch-local rows (10 / 1 floats)
    const float* old_logits; int old_stride;           // trajectory batch, rows idx[i] (q1env_learner_batch)
    const uint8_t* keys; const float* mouse; const float* logp_old; const float* adv; const float* value_old; const float* vtarg;
    const float* kl_coeff_dev;
    float clip, vf_clip, vf_coeff, ent_coeff;
    float inv_b, inv_bv;                                // (loss scale) / B of the policy / value network's dY
    float* stats_rows;
    int wide;                                           // logits / old_logits / obs rows are 8-byte aligned: 8-byte gathers
};
// ... and thread 0 of workgroup 0 derives the optimizer's bias corrections of step count + 1 (torch: 1 - beta ** step, in double) while
// the weight image's loads are in flight - the Adam kernel behind reads them (before: a one-wave kernel of its own, 5 us)
struct BcArgs { const long long* step; float* bc; float beta1, beta2; };

// 32x32 transposition on the matrix pipe: x0 / x1 = the T-format vectors (u = 0 / 1) of one 32-unit tile; returns the tile with
// lane = unit sigma(c), registers = samples, as float32 (exact), to be packed with cvt8.
__device__ __forceinline__ f32x16 transpose_tile(const f16x8 x0, const f16x8 x1, const f16x8 e0, const f16x8 e1) {
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, e0, zero16, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1, e1, d, 0, 0, 0);
    return d;
}

__device__ __forceinline__ void store_n(f16x8* dstN, uint32_t t, const f32x16& d) {      // dstN already points at (tile, lane)
    // (plain stores: with the nt policy the backward kernel ends 1.4 us sooner and the weight-gradient kernel, which reads these back,
    // takes 8.7 us longer - profiles/r4_learner_bytes.txt)
    dstN[(2u * t) * 64u] = cvt8(d, 0);
    dstN[(2u * t + 1u) * 64u] = cvt8(d, 1);
}

// acc (float32, C/D layout) *= 1 - h^2 with h the matching T-format vectors; returns nothing, acc updated in place
// (two elements per instruction - v_pk_mul_f32 / v_pk_add_f32 on register pairs: the same IEEE operations in the same order as the scalar
// form, RN(acc RN(1 - RN(h h))), so the bits are the same; 1.5 instead of 3 vector instructions per element, and this function was 41 % of
// the backward kernel's vector instructions)
__device__ __forceinline__ void times_dtanh(f32x16& acc, const f16x8 h0, const f16x8 h1) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t one = {1.0f, 1.0f};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const f32x2_t a = {(float)h0[e], (float)h0[e + 1]}, b = {(float)h1[e], (float)h1[e + 1]};
        f32x2_t x = {acc[e], acc[e + 1]}, y = {acc[8 + e], acc[8 + e + 1]};
        x = x * (one - a * a);
        y = y * (one - b * b);
        acc[e] = x[0]; acc[e + 1] = x[1];
        acc[8 + e] = y[0]; acc[8 + e + 1] = y[1];
    }
}

// FUSED: the per-sample inputs of a tile (everything that hangs on the gathered row index: two dependent round trips) are requested
// one tile AHEAD - the first tile's before the weight staging, the next tile's at the current tile's start (the row index a tile earlier
// still) - so that a lone wave per SIMD never sits on them (round 4: they were the front of every tile's critical path).
// Lanes (c, 0) and (c, 1) evaluate the same sample, so they SHARE its gathers: every gather instruction costs the address unit one
// step per active lane whatever the lanes ask for (tools/exp_bwd_stamps.py: 2.6 us per tile for the policy network's sixteen 64-lane
// gathers), so half 0 fetches the sample's logits row, key bits and mouse action, half 1 its old row, old log-probability and advantage,
// and the halves swap when the values are used, a tile later (tile_unpack) - 9 gathers' worth of lanes instead of 16, and 16 registers
// in flight per tile instead of 30.
struct TileIn {
    float r[10];              // policy network: half 0 the forward kernel's logits row, half 1 the behaviour policy's
    float sc[2];              // policy network: half 0 (key bits, mouse), half 1 (logp_old, adv); value network: (value_old, vtarg) in both halves
    float v;                  // value network: the forward kernel's value
    float x[4];               // observation inputs 4 half + e of the lane's sample (both networks)
};
struct TileVals { uint32_t kb; float mouse, logp_old, adv; float lg[10], ol[10]; };
__device__ __forceinline__ size_t tile_src(const int64_t* __restrict__ idx, uint32_t s, bool live) {
    return live ? (idx ? (size_t)idx[s] : (size_t)s) : 0;
}
__device__ __forceinline__ void tile_inputs(TileIn& ti, const LossArgs& la, const float* __restrict__ obs, bool second, bool live, uint32_t s, size_t src,
                                            uint32_t half) {
#pragma unroll
    for (int c = 0; c < 10; ++c) ti.r[c] = 0.0f;
    ti.sc[0] = ti.sc[1] = ti.v = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) ti.x[e] = 0.0f;
    if (!live) return;
    // (la.wide: rows are 8-byte aligned - the 10-float rows as five 8-byte loads, the observation as 8-byte loads)
    if (!second) {
        const float* row = half ? la.old_logits + src * (size_t)la.old_stride : la.logits + (size_t)s * 10u;
        if (la.wide) {
#pragma unroll
            for (int c = 0; c < 10; c += 2) {
                const float2 a = *reinterpret_cast<const float2*>(row + c);
                ti.r[c] = a.x; ti.r[c + 1] = a.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 10; ++c) ti.r[c] = row[c];
        }
        if (half == 0u) ti.sc[0] = __uint_as_float((uint32_t)la.keys[src]);
        else ti.sc[0] = la.logp_old[src];
        ti.sc[1] = (half ? la.adv : la.mouse)[src];
    } else {
        ti.v = la.value[s]; ti.sc[0] = la.value_old[src]; ti.sc[1] = la.vtarg[src];
    }
    static_assert(OBS == 6, "the observation row is read as 4 + 2 floats");
    if (la.wide) {
        const float2* o2 = reinterpret_cast<const float2*>(obs + src * OBS);
        if (half == 0u) { const float2 a = o2[0], b = o2[1]; ti.x[0] = a.x; ti.x[1] = a.y; ti.x[2] = b.x; ti.x[3] = b.y; }
        else { const float2 a = o2[2]; ti.x[0] = a.x; ti.x[1] = a.y; ti.x[2] = 1.0f; ti.x[3] = 0.0f; }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * (int)half + e;
            ti.x[e] = i < OBS ? obs[src * OBS + (uint32_t)i] : (i == OBS ? 1.0f : 0.0f);
        }
    }
}
// the policy network's values of a tile as every lane needs them: its own half's gathers + its partner lane's (lane ^ 32)
__device__ __forceinline__ TileVals tile_unpack(const TileIn& ti, uint32_t half) {
    TileVals v;
    float pr[10], ps[2];
#pragma unroll
    for (int c = 0; c < 10; ++c) pr[c] = __shfl_xor(ti.r[c], 32, 64);
    ps[0] = __shfl_xor(ti.sc[0], 32, 64); ps[1] = __shfl_xor(ti.sc[1], 32, 64);
#pragma unroll
    for (int c = 0; c < 10; ++c) { v.lg[c] = half ? pr[c] : ti.r[c]; v.ol[c] = half ? ti.r[c] : pr[c]; }
    v.kb = __float_as_uint(half ? ps[0] : ti.sc[0]);
    v.mouse = half ? ps[1] : ti.sc[1];
    v.logp_old = half ? ti.sc[0] : ps[0];
    v.adv = half ? ti.sc[1] : ps[1];
    return v;
}

template <bool FUSED>
__global__ void __launch_bounds__(256, 1)
learner_backward_kernel(int n, const float* __restrict__ obs, const int64_t* __restrict__ idx, const int64_t* __restrict__ idx_cursor, BwdNet net_a,
                        BwdNet net_b, int nets, LossArgs la, BcArgs bca) {
    if (idx && idx_cursor) idx += *idx_cursor;
    const uint32_t bgrid = nets == 2 ? gridDim.x / 2u : gridDim.x;
    const bool second = nets == 2 && blockIdx.x >= bgrid;
    const uint32_t bid = second ? blockIdx.x - bgrid : blockIdx.x;
    const BwdNet net = second ? net_b : net_a;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* l_w2t = lds;
    unsigned char* l_w3t = lds + LDS_W2T;
    const uint32_t tid = threadIdx.x;
#ifdef Q1_BWD_STAMPS
    const uint64_t stamp0 = wall_clock64();
#endif
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, half = lane >> 5;
    const uint32_t ntiles = ((uint32_t)n + 31u) / 32u;
    const uint32_t tstride = bgrid * 4u;
    TileIn cur, nxt;
    size_t src_next = 0;
    // stage both images: straight 16-byte copies (135168 + 20480 bytes) with ALL of a thread's loads in flight at once (a load -> store
    // loop would expose one L2 / HBM round trip per 4 KiB: 38 of them)
    {
        const uint32_t tile0 = bid * 4u + wave, s0 = tile0 * 32u + col;
        const bool live0 = tile0 < ntiles && s0 < (uint32_t)n;
        size_t src0 = 0;
        if constexpr (FUSED) {
            src0 = tile_src(idx, s0, live0);
            const uint32_t s1 = (tile0 + tstride) * 32u + col;                     // the row index of this wave's SECOND tile too: its inputs
            src_next = tile_src(idx, s1, tile0 + tstride < ntiles && s1 < (uint32_t)n);      // are requested at the first tile's start
        }
        StageRegs<256, (uint32_t)(LDS_W2T / 16)> r;
        stage_issue<256, (uint32_t)(LDS_W2T / 16)>(r, net.w2t, tid);
        if constexpr (FUSED) {
            tile_inputs(cur, la, obs, second, live0, s0, src0, half);              // the first tile's inputs land under the staging
            if (bca.step && blockIdx.x == 0 && tid == 0) {
                const long long t = *bca.step + 1;
                bca.bc[0] = (float)(1.0 - pow((double)bca.beta1, (double)t));
                bca.bc[1] = (float)(1.0 - pow((double)bca.beta2, (double)t));
            }
        }
        stage_commit<256, (uint32_t)(LDS_W2T / 16)>(l_w2t, r, tid);
    }
    stage_copy<256, (uint32_t)(LDS_W3T / 16)>(l_w3t, net.w3t, tid);
#ifdef Q1_BWD_STAMPS      // diagnostic build (tools/exp_bwd_stamps.py): where does a wave's time go?  100 MHz stamps of wave 0, in statistics rows 1024..
    float stamps[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int stamp_tile = 0;
#define Q1_STAMP(k) do { if (FUSED && stamp_tile == 0) stamps[k] = (float)(wall_clock64() - stamp0) * 0.01f; } while (0)
#else
#define Q1_STAMP(k) do { } while (0)
#endif
    __syncthreads();
    Q1_STAMP(0);
    // selection operands of the transposition: E0[K][c] = [K == c], E1[K][c] = [K == c - 16]; a lane (c, h) holds K = 8 h + e
    f16x8 e0, e1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        e0[e] = (8u * half + (uint32_t)e == col) ? (_Float16)1.0f : (_Float16)0.0f;
        e1[e] = (8u * half + (uint32_t)e + 16u == col) ? (_Float16)1.0f : (_Float16)0.0f;
    }
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int OUT = net.out_dim;
    const unsigned char* w2trow = l_w2t + (size_t)col * ROW_BYTES + half * 16u;           // + 32 t1 rows, + K-step q * 32 B
    const unsigned char* w3trow = l_w3t + (size_t)col * W3T_ROW_BYTES + half * 16u;       // + 32 t2 rows, + ks * 32 B
    float amax = 0.0f;                                                                    // largest |gradient element| this lane converted
    float st[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};                                         // FUSED: this lane's statistics sums
    float klc = 0.0f;
    if constexpr (FUSED) klc = *la.kl_coeff_dev;
    for (uint32_t tile = bid * 4u + wave; tile < ntiles; tile += tstride) {
        const uint32_t s = tile * 32u + col;
        const bool live = s < (uint32_t)n;
        const size_t tbase = (size_t)tile * TILE_VECS + lane;
        // every global operand of the tile is requested up front (one wave per SIMD: nothing else hides an HBM round trip): the 16
        // h2 vectors now, the 16 h1 vectors before the 128-MFMA data-gradient loop they are needed after.  (Round 4 tried requesting
        // them a phase earlier still - h2 of the next tile after this tile's dZ2 phase, h1 at the tile's start: 506 registers, the
        // second tile of a wave twice as slow; tools/exp_bwd_stamps.py shows the phases are issue-bound, not waiting for these loads.)
        f16x8 hv[8][2];
#pragma unroll
        for (int t = 0; t < 8; ++t) { hv[t][0] = net.h2T[tbase + (2u * t) * 64u]; hv[t][1] = net.h2T[tbase + (2u * t + 1u) * 64u]; }
        // ---- dY as B operand(s): K = output index o = 16 ks + 8 h + e
        f16x8 dyb0, dyb1;
        const uint32_t tile_n = tile + tstride, s_n = tile_n * 32u + col;
        const bool live_n = tile_n < ntiles && s_n < (uint32_t)n;
        if constexpr (FUSED) {
            // the NEXT tile's inputs are requested here, a whole tile ahead (their row index came in a tile earlier still): the allocator
            // parks them in the other register file before the MFMA loop, and a move waits for its load - requested just in front of
            // the loop (this round's first version) they held the wave for 4 us there (tools/exp_bwd_stamps.py)
            tile_inputs(nxt, la, obs, second, live_n, s_n, src_next, half);
            {
                const uint32_t s_nn = (tile_n + tstride) * 32u + col;
                src_next = tile_src(idx, s_nn, tile_n + tstride < ntiles && s_nn < (uint32_t)n);
            }
            float y0[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y0[e] = 0.0f;
            if (!second) {
                float g[10];
#pragma unroll
                for (int c = 0; c < 10; ++c) g[c] = 0.0f;
                const TileVals tv = tile_unpack(cur, half);                        // (all 64 lanes: the halves swap their gathers)
                {
                    // (all 64 lanes as well: the two lanes of a sample split its four keys - q1ppo_loss.hpp PAIR; a dead lane's inputs are zeros)
                    const PpoSample in{tv.kb, tv.mouse, tv.logp_old, tv.adv};
                    const PpoSums ps = ppo_policy_grad<true, true>(la.p, tv.lg, tv.ol, in, la.clip, la.ent_coeff, klc, la.inv_b, g, 10, half);
                    if (live && half == 0u) { st[0] += ps.ent; st[1] += ps.kl; st[2] += -ps.surr; st[3] += -ps.surr + klc * ps.kl - la.ent_coeff * ps.ent; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) y0[e] = live ? (half ? (e < 2 ? g[8 + e] : 0.0f) : g[e]) : 0.0f;
            } else if (live) {
                float vf;
                const float dvf = ppo_value_grad(cur.v, cur.sc[0], cur.sc[1], la.vf_clip, vf);
                if (half == 0u) { y0[0] = la.vf_coeff * dvf * la.inv_bv; st[3] += la.vf_coeff * vf; st[4] += vf; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                amax = fmaxf(fabsf(y0[e]), amax);
                dyb0[e] = (_Float16)fminf(fmaxf(y0[e], -65504.0f), 65504.0f);
                dyb1[e] = (_Float16)0.0f;
            }
        } else {
            const float* row = net.dy + (size_t)(live ? s : 0) * (uint32_t)net.dy_stride;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int o0 = 8 * (int)half + e, o1 = 16 + o0;
                // (saturating: a per-sample gradient beyond float16's 65504 - a value error of tens of thousands - must not become inf)
                const float y0 = (live && o0 < OUT) ? row[o0] : 0.0f, y1 = (live && o1 < OUT) ? row[o1] : 0.0f;
                amax = fmaxf(fmaxf(fabsf(y0), fabsf(y1)), amax);
                dyb0[e] = (_Float16)fminf(fmaxf(y0, -65504.0f), 65504.0f);
                dyb1[e] = (_Float16)fminf(fmaxf(y1, -65504.0f), 65504.0f);
            }
        }
        // ---- the weight-gradient kernel's small operands, transposed here: [x | 1] (inputs as "units" 0..7 of a T-format tile: element e
        //      < 4 of lane (c, h) = input 4 h + e) and dY (K slot = output index, so the transposition delivers lane = output)
        {
            f16x8 x0;
            const size_t src = FUSED ? 0 : tile_src(idx, s, live);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = 4 * (int)half + e;
                float v = 0.0f;
                if constexpr (FUSED) { if (e < 4) v = cur.x[e]; }
                else if (live && e < 4) v = i < OBS ? obs[src * OBS + (uint32_t)i] : (i == OBS ? 1.0f : 0.0f);
                x0[e] = (_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f);
            }
            const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
            const f32x16 dx = transpose_tile(x0, zero8, e0, e1);
            const size_t sb = (size_t)tile * 128u + lane;
            net.xN[sb] = cvt8(dx, 0); net.xN[sb + 64u] = cvt8(dx, 1);
            const f32x16 dd = transpose_tile(dyb0, dyb1, e0, e1);
            net.dyN[sb] = cvt8(dd, 0); net.dyN[sb + 64u] = cvt8(dd, 1);
        }
        Q1_STAMP(1);
        // ---- dH2^T = W3^T dY^T, dZ2 = dH2 (1 - h2^2); h2 and dZ2 leave in N-format
        f16x8 dzb[8][2];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(w3trow + (size_t)t * 32u * W3T_ROW_BYTES);
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, dyb0, zero16, 0, 0, 0);
            if (OUT > 16) {
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(w3trow + (size_t)t * 32u * W3T_ROW_BYTES + 32u);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, dyb1, acc, 0, 0, 0);
            }
            times_dtanh(acc, hv[t][0], hv[t][1]);
            dzb[t][0] = cvt8_sat(acc, 0, amax);
            dzb[t][1] = cvt8_sat(acc, 1, amax);
            store_n(net.dz2N + tbase, (uint32_t)t, transpose_tile(dzb[t][0], dzb[t][1], e0, e1));
            Q1_STAMP(5 + t);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) { hv[t][0] = net.h1T[tbase + (2u * t) * 64u]; hv[t][1] = net.h1T[tbase + (2u * t + 1u) * 64u]; }
        Q1_STAMP(2);
        // ---- dH1^T = W2^T dZ2^T: 16 K-steps (j) x 8 row tiles (k).  (Round 4 also built this loop over PAIRS of row tiles with the
        //      previous pair's epilogue - (1 - h^2), conversion, transposition, stores - cut into 16 pieces between the K-steps, so that
        //      vector and matrix instructions overlap and four accumulators are live instead of eight: same bits, same 105 us step.)
        // Two passes of FOUR row tiles (round 4): with eight accumulators live the loop held 128 + 32 + 64 registers of its own next to the
        // 64 of the h1 vectors and the 30 of the next tile's inputs that are in flight across it - and the allocator parked those in the
        // other register file BEFORE the loop, i.e. waited for their loads there (tools/exp_bwd_stamps.py: 1.3 - 5.7 us per tile between
        // the dZ2 phase and the first MFMA).  Four accumulators leave room for everything; each pass's dZ1 epilogue follows it.
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            f32x16 acc1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc1[t] = zero16;
            {
                f16x8 a[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const f16x8*>(w2trow + (size_t)(4 * pass + t) * 32u * ROW_BYTES);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    // every W2^T operand register is re-requested for the next K-step right after the MFMA that consumed it was issued
                    // (operands are read at issue), i.e. four MFMAs ahead of its next use
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], dzb[q >> 1][q & 1], acc1[t], 0, 0, 0);
                        if (q < 15) a[t] = *reinterpret_cast<const f16x8*>(w2trow + (size_t)(4 * pass + t) * 32u * ROW_BYTES + (uint32_t)(q + 1) * 32u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (pass == 1) Q1_STAMP(3);
            // ---- dZ1 = dH1 (1 - h1^2); dZ1 leaves in N-format
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                times_dtanh(acc1[t], hv[4 * pass + t][0], hv[4 * pass + t][1]);
                const f16x8 z0 = cvt8_sat(acc1[t], 0, amax), z1 = cvt8_sat(acc1[t], 1, amax);
                store_n(net.dz1N + tbase, (uint32_t)(4 * pass + t), transpose_tile(z0, z1, e0, e1));
            }
        }
        if constexpr (FUSED) cur = nxt;
        Q1_STAMP(4);
#ifdef Q1_BWD_STAMPS
        ++stamp_tile;
#endif
    }
#ifdef Q1_BWD_STAMPS
    if constexpr (FUSED) {
        if (tid == 0) {
            stamps[15] = (float)(wall_clock64() - stamp0) * 0.01f;
            for (int k = 0; k < 16; ++k) la.stats_rows[5120u + (size_t)blockIdx.x * 16u + k] = stamps[k];
        }
    }
#endif
    if constexpr (FUSED) {
        __shared__ float red[4][5];
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) st[k] += __shfl_down(st[k], off, 64);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) red[wave][k] = st[k];
        }
        __syncthreads();
        if (tid < 5u) la.stats_rows[(size_t)blockIdx.x * 5u + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
    // saturation report: one per WORKGROUP (round 6: per wave it was 2 x 1 024 atomics on two words per launch, executed one after the other by the
    // memory side while the dispatch waited - see q1learner_fused.hpp)
    if (net.sat) {
        __shared__ uint32_t red_sat[4][2];
        const uint64_t over = __ballot(amax > 65504.0f);
        float wmax = amax;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
        if (lane == 0) { red_sat[wave][0] = (uint32_t)__popcll(over); red_sat[wave][1] = __float_as_uint(wmax); }
        __syncthreads();
        if (tid == 0)
            sat_report(net.sat, (red_sat[0][0] + red_sat[1][0]) + (red_sat[2][0] + red_sat[3][0]),
                       max(max(red_sat[0][1], red_sat[1][1]), max(red_sat[2][1], red_sat[3][1])));
    }
}

// ------------------------------------------------------------------------------------------------------------------ weight gradients
struct WgNet {
    // round 4: h1 / h2 are read in the forward kernel's T-format and transposed HERE, on the matrix pipe this kernel leaves idle (it
    // streams its operands at the memory system's rate: 6.2 TB/s, profiles/r4_learner_bytes.txt) - the backward kernel no longer writes
    // N-format copies of them (67 MB of the step's 514)
    const f16x8* dz2N; const f16x8* dz1N; const f16x8* h1T; const f16x8* h2T; const f16x8* xN; const f16x8* dyN;
    float* partial;           // float[splits][PARTIAL_STRIDE]
    const float* dw1p;        // DW1 (round 6, q1learner_fused.hpp): float[tile][unit tile 8][lane 32][reg 8], the per-tile products [x | 1]^T dZ1 the fused
                              // forward + backward kernel leaves instead of dZ1 itself (dz1N is then never read)
    const float4* dw3a;       // DW1: float4[tile][unit tile 8][lane 64], rows 0..7 of the per-tile products dY^T h2 (policy network; NULL for a one-output network)
    const float* dw3b;        // DW1: float[tile][unit tile 8][lane 64]: row 8 / 12 (outputs 8, 9) of the same products - or row 0 of a one-output network's
                              // (h2T is then never read)
};

// One workgroup = four waves = four of the eight 32-unit row tiles of the gradient side (blockIdx.z picks the half); it owns a contiguous
// range of sample tiles (split-K over workgroups, blockIdx.x) and one half of the products (blockIdx.y): y = 0 the dW2 column tiles 0..3 + the [x | 1] products
// (dW1, db1, db2), y = 1 the column tiles 4..7 + the dY products (dW3, db3) - six float32 accumulator tiles per wave.  Every operand
// is a plain 16-byte-per-lane streaming load in MFMA layout (the backward kernel has done all gathering and transposing); the next
// tile's operands are requested before the current tile's MFMAs (register double buffer).  It leaves its float32 partial sums in
// its split's slot.
struct WgOps { f16x8 a2[2], s0[2], s1[2], x[2], b[4][2]; float4 p[2]; float q; };     // dZ2 rows | y=0: dZ1 rows (DW1: the tile's dW1 products in p), [x|1]  y=1: dY, h2 cols, [x|1] | h1 column tiles

// COL0 / NKT: the wave's dW2 column tiles are COL0 .. COL0 + NKT - 1 (NKT = 4: round 4's two halves per row tile; NKT = 2: four quarters, two
// workgroups per CU - round 6); the quarter that starts at column 0 also carries the [x | 1] products (EX = 0), the one at column 4 the dY products (EX = 1)
template <int COL0, int NKT, bool DW1>
__device__ __forceinline__ void wg_load(WgOps& o, const WgNet& net, uint32_t tile, uint32_t lane, uint32_t w) {
    constexpr int EX = COL0 == 0 ? 0 : (COL0 == 4 ? 1 : -1);
    const size_t tb = (size_t)tile * TILE_VECS + lane, sb = (size_t)tile * 128u + lane;
    if constexpr (EX == 1 && DW1) {
        const size_t pb = ((size_t)tile * 8u + w) * 64u + lane;
        o.p[0] = net.dw3a ? net.dw3a[pb] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);       // (wave-uniform)
        o.q = net.dw3b[pb];
    }
    if constexpr (EX == 0 && DW1) {
        // 32 x 7 float32 products of the tile (registers 0..6 of lanes 0..31 in the C / D layout of [x | 1]^T dZ1); the upper lanes re-read the lower lanes' (unused)
        const float4* q = reinterpret_cast<const float4*>(net.dw1p + ((size_t)tile * 8u + w) * 256u + (size_t)(lane & 31u) * 8u);
        o.p[0] = q[0]; o.p[1] = q[1];
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        o.a2[ks] = net.dz2N[tb + (2u * w + (uint32_t)ks) * 64u];
        if constexpr (EX == 0) {
            if constexpr (!DW1) o.s0[ks] = net.dz1N[tb + (2u * w + (uint32_t)ks) * 64u];
            o.s1[ks] = net.xN[sb + 64u * (uint32_t)ks];
        }
        // (h2 / h1: T-format vectors u = ks of the unit tile - the same addressing as an N-format array; transposed by wg_transpose)
        else if constexpr (EX == 1) {
            o.s0[ks] = net.dyN[sb + 64u * (uint32_t)ks]; o.x[ks] = net.xN[sb + 64u * (uint32_t)ks];
            if constexpr (!DW1) o.s1[ks] = net.h2T[tb + (2u * w + (uint32_t)ks) * 64u];
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) o.b[kt][ks] = net.h1T[tb + (2u * (uint32_t)(COL0 + kt) + (uint32_t)ks) * 64u];
    }
}

// T-format pair (u = 0, 1) of one 32-unit tile -> the N-format pair (ks = 0, 1): what learner_backward_kernel's store_n wrote in round 3
__device__ __forceinline__ void wg_transpose(f16x8 (&v)[2], const f16x8 e0, const f16x8 e1) {
    const f32x16 d = transpose_tile(v[0], v[1], e0, e1);
    v[0] = cvt8(d, 0);
    v[1] = cvt8(d, 1);
}

#ifndef Q1_WGRAD_DEPTH                  // operand sets in flight per wave (measurement knob): tiles requested ahead = depth - 1
#define Q1_WGRAD_DEPTH 2
#endif
// The sample-tile loop of one wave.  One workgroup per CU (splits x 2 x 2 = 256 of them at the default 32 splits) = one wave per
// SIMD: nothing hides a load's latency but the wave's own requests, so the ring keeps DEPTH - 1 tiles in flight (the workgroup owns
// its CU's register file: 4 x 56..64 operand registers + 96 accumulators), and the body is straight-line - KHALF is a template
// argument, the db3 product (dY x [x | 1]) is computed by every wave of the y = 1 half and stored by one - so that the compiler's
// s_waitcnt are counts, not drains (the first version's in-loop direct load for that product drained the queue every tile).
template <int COL0, int NKT, bool DW1>
__device__ __forceinline__ void wg_loop(const WgNet& net, uint32_t t_begin, uint32_t t_end, uint32_t lane, uint32_t w, f32x16 (&aW2)[NKT], f32x16& aX, f32x16& aY) {
    constexpr int EX = COL0 == 0 ? 0 : (COL0 == 4 ? 1 : -1);
    constexpr int D = Q1_WGRAD_DEPTH;
    WgOps ring[D];
    if (t_begin >= t_end) return;
    // selection operands of the transposition (as in learner_backward_kernel): E0[K][c] = [K == c], E1[K][c] = [K == c - 16]
    f16x8 e0, e1;
    {
        const uint32_t col = lane & 31u, half = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            e0[e] = (8u * half + (uint32_t)e == col) ? (_Float16)1.0f : (_Float16)0.0f;
            e1[e] = (8u * half + (uint32_t)e + 16u == col) ? (_Float16)1.0f : (_Float16)0.0f;
        }
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d) wg_load<COL0, NKT, DW1>(ring[d], net, min(t_begin + (uint32_t)d, t_end - 1u), lane, w);
    for (uint32_t tile0 = t_begin; tile0 < t_end; tile0 += (uint32_t)D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const uint32_t tile = tile0 + (uint32_t)j;
            if (tile >= t_end) return;                                             // wave-uniform
            wg_load<COL0, NKT, DW1>(ring[(j + D - 1) % D], net, min(tile + (uint32_t)(D - 1), t_end - 1u), lane, w);   // (past the end: re-requests the last tile, harmless)
            __builtin_amdgcn_sched_barrier(0);                                     // the requests go out HERE, not next to their uses
            WgOps& cur = ring[j];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) wg_transpose(cur.b[kt], e0, e1);        // h1 column tiles: T -> N
            if constexpr (EX == 1 && !DW1) wg_transpose(cur.s1, e0, e1);          // h2 tile w: T -> N
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) aW2[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a2[ks], cur.b[kt][ks], aW2[kt], 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (EX == 0) {
                    aX = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a2[ks], cur.s1[ks], aX, 0, 0, 0);
                    if constexpr (!DW1) aY = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.s0[ks], cur.s1[ks], aY, 0, 0, 0);
                } else if constexpr (EX == 1) {
                    if constexpr (!DW1) aX = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.s0[ks], cur.s1[ks], aX, 0, 0, 0);
                    aY = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.s0[ks], cur.x[ks], aY, 0, 0, 0);
                }
            }
            if constexpr (EX == 1 && DW1) {                                          // the tile's dW3 products (rows = outputs), added in tile order
                aX[0] += cur.p[0].x; aX[1] += cur.p[0].y; aX[2] += cur.p[0].z; aX[3] += cur.p[0].w;
                if (net.dw3a) aX[4] += cur.q; else aX[0] += cur.q;                   // (wave-uniform: ten outputs / one)
            }
            if constexpr (EX == 0 && DW1) {                                          // the tile's dW1 / db1 products, added in tile order
                aY[0] += cur.p[0].x; aY[1] += cur.p[0].y; aY[2] += cur.p[0].z; aY[3] += cur.p[0].w;
                aY[4] += cur.p[1].x; aY[5] += cur.p[1].y; aY[6] += cur.p[1].z; aY[7] += cur.p[1].w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// DW1: products 72..79 of the slab hold [x | 1]^T dZ1 (rows = inputs: registers 0..6 of lanes 0..31; slot_of(.., dw1 = true)) instead of dZ1^T [x | 1],
// and output 9 of products 80..87 sits in row 12 (register 4 of the half-1 lanes) instead of row 9
// NKT = 2 (round 6): blockIdx.y = 0..3 picks a QUARTER of the column tiles - 512 workgroups of at most 256 registers per lane, two per CU: eight waves per
// CU keep twice the loads in flight in a loop that waits for one tile's operands per iteration (the same accumulation chains: the same bits)
template <bool DW1, int NKT = 4>
__global__ void __launch_bounds__(256, NKT == 2 ? 2 : 1)
learner_wgrad_kernel(int n, WgNet net_a, WgNet net_b, int splits) {
    const bool second = blockIdx.x >= (uint32_t)splits;
    const uint32_t split = second ? blockIdx.x - (uint32_t)splits : blockIdx.x;
    const uint32_t kq = blockIdx.y;                    // column group: tiles NKT kq .. NKT kq + NKT - 1
    const WgNet net = second ? net_b : net_a;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = (tid >> 6) + 4u * blockIdx.z;      // row tile of the gradient side
    const uint32_t ntiles = ((uint32_t)n + 31u) / 32u;
    const uint32_t per = (ntiles + (uint32_t)splits - 1u) / (uint32_t)splits;
    const uint32_t t_begin = split * per, t_end = min(t_begin + per, ntiles);
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 aW2[NKT], aX = zero16, aY = zero16;        // columns 0..: aX = dZ2 x [x|1], aY = dZ1 x [x|1];  columns 4..: aX = dY x h2, aY = dY x [x|1] (kept by wave 0)
#pragma unroll
    for (int k = 0; k < NKT; ++k) aW2[k] = zero16;
    if constexpr (NKT == 4) {
        if (kq == 0) wg_loop<0, 4, DW1>(net, t_begin, t_end, lane, w, aW2, aX, aY);
        else wg_loop<4, 4, DW1>(net, t_begin, t_end, lane, w, aW2, aX, aY);
    } else {
        if (kq == 0) wg_loop<0, 2, DW1>(net, t_begin, t_end, lane, w, aW2, aX, aY);
        else if (kq == 1) wg_loop<2, 2, DW1>(net, t_begin, t_end, lane, w, aW2, aX, aY);
        else if (kq == 2) wg_loop<4, 2, DW1>(net, t_begin, t_end, lane, w, aW2, aX, aY);
        else wg_loop<6, 2, DW1>(net, t_begin, t_end, lane, w, aW2, aX, aY);
    }
    float* out = net.partial + (size_t)split * PARTIAL_STRIDE;
    auto put = [&](uint32_t p, const f32x16& a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((size_t)p * 16u + (uint32_t)r) * 64u + lane] = a[r];
    };
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) put(8u * w + (uint32_t)NKT * kq + (uint32_t)kt, aW2[kt]);
    if (kq == 0) {
        put(64u + w, aX);
        put(72u + w, aY);
    } else if ((uint32_t)NKT * kq == 4u) {
        put(80u + w, aX);
        if (w == 0) put(88u, aY);
    }
}

// Round 6: the weight-gradient kernel of the fused step (product arrays: no dZ1, no tanh(H2)) with the h1 column tiles SHARED through LDS.  In
// learner_wgrad_kernel all four waves of a workgroup load and transpose the same four h1 tiles (8 of the 13 KB a wave requests per sample tile; the
// CU's vector cache filled at half its rate, and more waves made it worse: the column-quarter form above, 35 us against 27.6) - here wave v loads and
// transposes ONE of them (column tile COL0 + v) per sample tile, leaves it in LDS (two stages of 4 x 2 KB, one barrier per tile: a stage is
// rewritten two tiles later, behind the barrier every reader of it has passed) and all four read their B operands from there: 5 - 7 KB of requests
// per wave and tile, one transposition instead of four, and a ring of three tiles in flight in the registers that frees.  The accumulation chains
// are learner_wgrad_kernel<true>'s: the same bits.
constexpr int WGS_DEPTH = 3;
struct WgsOps { f16x8 a2[2], bo[2], s[2], x[2]; float4 p[2]; float q; };

template <int COL0>
__device__ __forceinline__ void wgs_load(WgsOps& o, const WgNet& net, uint32_t tile, uint32_t lane, uint32_t v, uint32_t w) {
    constexpr int EX = COL0 == 0 ? 0 : 1;
    const size_t tb = (size_t)tile * TILE_VECS + lane, sb = (size_t)tile * 128u + lane;
    if constexpr (EX == 1) {
        const size_t pb = ((size_t)tile * 8u + w) * 64u + lane;
        o.p[0] = net.dw3a ? net.dw3a[pb] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);       // (wave-uniform)
        o.q = net.dw3b[pb];
    } else {
        const float4* q = reinterpret_cast<const float4*>(net.dw1p + ((size_t)tile * 8u + w) * 256u + (size_t)(lane & 31u) * 8u);
        o.p[0] = q[0]; o.p[1] = q[1];
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        o.a2[ks] = net.dz2N[tb + (2u * w + (uint32_t)ks) * 64u];
        o.bo[ks] = net.h1T[tb + (2u * ((uint32_t)COL0 + v) + (uint32_t)ks) * 64u];
        if constexpr (EX == 0) o.s[ks] = net.xN[sb + 64u * (uint32_t)ks];
        else { o.s[ks] = net.dyN[sb + 64u * (uint32_t)ks]; o.x[ks] = net.xN[sb + 64u * (uint32_t)ks]; }
    }
}

template <int COL0>
__device__ __forceinline__ void wgs_loop(const WgNet& net, uint32_t t_begin, uint32_t t_end, uint32_t lane, uint32_t v, uint32_t w, f32x16 (&aW2)[4], f32x16& aX,
                                         f32x16& aY, f16x8* lds) {
    constexpr int EX = COL0 == 0 ? 0 : 1;
    constexpr int D = WGS_DEPTH;
    WgsOps ring[D];
    if (t_begin >= t_end) return;                                                  // (the same range for the workgroup's four waves: the barriers below are uniform)
    f16x8 e0, e1;
    {
        const uint32_t col = lane & 31u, half = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            e0[e] = (8u * half + (uint32_t)e == col) ? (_Float16)1.0f : (_Float16)0.0f;
            e1[e] = (8u * half + (uint32_t)e + 16u == col) ? (_Float16)1.0f : (_Float16)0.0f;
        }
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d) wgs_load<COL0>(ring[d], net, min(t_begin + (uint32_t)d, t_end - 1u), lane, v, w);
    uint32_t stage = 0;
    for (uint32_t tile0 = t_begin; tile0 < t_end; tile0 += (uint32_t)D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const uint32_t tile = tile0 + (uint32_t)j;
            if (tile >= t_end) return;                                             // workgroup-uniform
            wgs_load<COL0>(ring[(j + D - 1) % D], net, min(tile + (uint32_t)(D - 1), t_end - 1u), lane, v, w);
            __builtin_amdgcn_sched_barrier(0);
            WgsOps& cur = ring[j];
            wg_transpose(cur.bo, e0, e1);                                          // this wave's h1 column tile: T -> N
            f16x8* mine = lds + ((stage * 4u + v) * 2u) * 64u + lane;
            mine[0] = cur.bo[0]; mine[64] = cur.bo[1];
            __syncthreads();
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const f16x8 b = lds[((stage * 4u + (uint32_t)kt) * 2u + (uint32_t)ks) * 64u + lane];
                    aW2[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a2[ks], b, aW2[kt], 0, 0, 0);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (EX == 0) aX = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a2[ks], cur.s[ks], aX, 0, 0, 0);
                else aY = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.s[ks], cur.x[ks], aY, 0, 0, 0);
            }
            if constexpr (EX == 1) {                                               // the tile's dW3 products, added in tile order
                aX[0] += cur.p[0].x; aX[1] += cur.p[0].y; aX[2] += cur.p[0].z; aX[3] += cur.p[0].w;
                if (net.dw3a) aX[4] += cur.q; else aX[0] += cur.q;
            } else {                                                               // the tile's dW1 / db1 products
                aY[0] += cur.p[0].x; aY[1] += cur.p[0].y; aY[2] += cur.p[0].z; aY[3] += cur.p[0].w;
                aY[4] += cur.p[1].x; aY[5] += cur.p[1].y; aY[6] += cur.p[1].z; aY[7] += cur.p[1].w;
            }
            stage ^= 1u;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ void __launch_bounds__(256, 1)
learner_wgrad_shared_kernel(int n, WgNet net_a, WgNet net_b, int splits) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[2 * 4 * 2 * 64];            // 16 KB: two stages of four N-format h1 tiles
    const bool second = blockIdx.x >= (uint32_t)splits;
    const uint32_t split = second ? blockIdx.x - (uint32_t)splits : blockIdx.x;
    const uint32_t khalf = blockIdx.y;
    const WgNet net = second ? net_b : net_a;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, v = tid >> 6, w = v + 4u * blockIdx.z;
    const uint32_t ntiles = ((uint32_t)n + 31u) / 32u;
    const uint32_t per = (ntiles + (uint32_t)splits - 1u) / (uint32_t)splits;
    const uint32_t t_begin = split * per, t_end = min(t_begin + per, ntiles);
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 aW2[4], aX = zero16, aY = zero16;
#pragma unroll
    for (int k = 0; k < 4; ++k) aW2[k] = zero16;
    if (khalf == 0) wgs_loop<0>(net, t_begin, t_end, lane, v, w, aW2, aX, aY, lds);
    else wgs_loop<4>(net, t_begin, t_end, lane, v, w, aW2, aX, aY, lds);
    float* out = net.partial + (size_t)split * PARTIAL_STRIDE;
    // Only the slots slot_of(.., dw1 = true) maps to a parameter are written (nobody reads the others): of the 25 small products' 100 KB per split and
    // network, 19 KB - 6.5 of the 23 MB this kernel writes in its last microsecond, which no load is left to hide.
    auto put = [&](uint32_t p, const f32x16& a, int regs, bool lanes) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r < regs && lanes) out[((size_t)p * 16u + (uint32_t)r) * 64u + lane] = a[r];
    };
    const uint32_t cb = lane & 31u;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) put(8u * w + 4u * khalf + (uint32_t)kt, aW2[kt], 16, true);
    if (khalf == 0) {
        put(64u + w, aX, 16, cb == sigma((uint32_t)OBS));      // db2: the column of the constant 1
        put(72u + w, aY, 7, lane < 32u);                       // dW1 / db1: rows = inputs, the half-0 lanes
    } else {
        if (net.dw3a) put(80u + w, aX, 5, true);               // dW3, ten outputs: rows 0..8 and 12
        else put(80u + w, aX, 1, lane < 32u);                  // one output: row 0
        if (w == 0) put(88u, aY, 16, cb == sigma((uint32_t)OBS));   // db3
    }
}

// ------------------------------------------------------------------------------------------------------------------ reduction
struct Grads {
    float* w1; float* b1; float* w2; float* b2; float* w3; float* b3;      // torch layouts: [256][6], [256], [256][256], [256], [out][256], [out]
    int out_dim;
};

// The reduction and the optimizer walk the split-K partial sums in the slab's OWN order: a thread owns slot off = product x 1024 +
// register row x 64 + lane, so a wave reads 256 contiguous bytes per split and the grid streams every slab front to back, eight
// splits in flight per thread (round 4; in parameter order - two 128-byte pieces 4 KB apart per wave and split, one load in flight -
// the fused Adam kernel took 15 us for 23 MB, with eight in flight 25: the more splits open at once, the fewer DRAM pages reused; in slab
// order 13 us, with eight in flight 7.3).  slot_of() is the inverse of the products' index map (sigma is its own inverse): which
// parameter a slot holds; slots that belong to none (columns of the [x | 1] products that only the bias gradients use, output rows beyond
// out_dim) retire.
struct Slot { int arr; uint32_t i, e, x, y; };     // arr: 0 w2[x][y], 1 b2, 2 w1, 3 b1, 4 w3[x][y], 5 b3, -1 none; i = index in arr, e = flat index [w2|b2|w1|b1|w3|b3]
__device__ __forceinline__ Slot slot_of(uint32_t off, uint32_t OUT, bool dw1 = false) {
    const uint32_t nW2 = 65536u, nB2 = 256u, nW1 = 256u * (uint32_t)OBS, nB1 = 256u, nW3 = OUT * 256u;
    const uint32_t p = off >> 10, r = (off >> 6) & 15u, ln = off & 63u, h = ln >> 5, cb = ln & 31u;
    const uint32_t ia = (r & 3u) + 4u * h + 8u * (r >> 2);
    const uint32_t ua = sigma(ia), ub = sigma(cb);                       // the A-side / B-side index within the 32 x 32 tile
    Slot s{-1, 0, 0, 0, 0};
    if (p < 64u) {
        s.x = 32u * (p >> 3) + ua; s.y = 32u * (p & 7u) + ub; s.arr = 0; s.i = s.x * 256u + s.y; s.e = s.i;
    } else if (p < 72u) {
        if (ub == (uint32_t)OBS) { s.arr = 1; s.i = 32u * (p - 64u) + ua; s.e = nW2 + s.i; }
    } else if (p < 80u) {
        // (dw1: the transposed product of the fused forward + backward kernel - register r of a half-0 lane = input r, lane = unit sigma(cb))
        const uint32_t k = 32u * (p - 72u) + (dw1 ? ub : ua);
        const uint32_t in = dw1 ? (h == 0u ? r : 32u) : ub;
        if (in < (uint32_t)OBS) { s.arr = 2; s.i = k * (uint32_t)OBS + in; s.e = nW2 + nB2 + s.i; }
        else if (in == (uint32_t)OBS) { s.arr = 3; s.i = k; s.e = nW2 + nB2 + nW1 + k; }
    } else if (p < 88u) {
        // (dw1: the fused kernel's per-tile products keep outputs 0..8 in rows 0..8 and output 9 in row 12 - five registers per lane)
        const uint32_t o = dw1 ? (ia == 12u ? 9u : (ia <= 8u ? ia : 99u)) : ia;
        if (o < OUT) { s.x = o; s.y = 32u * (p - 80u) + ub; s.arr = 4; s.i = s.x * 256u + s.y; s.e = nW2 + nB2 + nW1 + nB1 + s.i; }
    } else if (p == 88u) {
        if (ub == (uint32_t)OBS && ia < OUT) { s.arr = 5; s.i = ia; s.e = nW2 + nB2 + nW1 + nB1 + nW3 + ia; }
    }
    return s;
}
__device__ __forceinline__ float slab_sum(const float* __restrict__ partial, int splits, uint32_t off) {
    constexpr int D = 8;                                                 // loads in flight; added in split order (the bits do not depend on D)
    float g = 0.0f;
    int k = 0;
    for (; k + D <= splits; k += D) {
        float t[D];
#pragma unroll
        for (int u = 0; u < D; ++u) t[u] = partial[(size_t)(k + u) * PARTIAL_STRIDE + off];
#pragma unroll
        for (int u = 0; u < D; ++u) g += t[u];
    }
    for (; k < splits; ++k) g += partial[(size_t)k * PARTIAL_STRIDE + off];
    return g;
}
__device__ __forceinline__ float* slot_ptr(const Slot& s, float* w2, float* b2, float* w1, float* b1, float* w3, float* b3) {
    float* const base = s.arr == 0 ? w2 : s.arr == 1 ? b2 : s.arr == 2 ? w1 : s.arr == 3 ? b1 : s.arr == 4 ? w3 : b3;
    return base + s.i;
}

__global__ void __launch_bounds__(256)
learner_reduce_kernel(const float* __restrict__ pa, const float* __restrict__ pb, Grads ga, Grads gb, int splits, float inv_scale_a, float inv_scale_b,
                      int dw1 = 0) {
    const bool second = blockIdx.y == 1;
    const float inv_scale = second ? inv_scale_b : inv_scale_a;
    const float* __restrict__ partial = second ? pb : pa;
    const Grads g = second ? gb : ga;
    const uint32_t off = blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= (uint32_t)PARTIAL_FLOATS) return;
    const Slot sl = slot_of(off, (uint32_t)g.out_dim, dw1 != 0);
    if (sl.arr < 0) return;
    *slot_ptr(sl, g.w2, g.b2, g.w1, g.b1, g.w3, g.b3) = inv_scale * slab_sum(partial, splits, off);
}

__device__ __forceinline__ uint32_t kperm(uint32_t p) { return (p & ~0xCu) | ((p & 4u) << 1) | ((p & 8u) >> 1); }   // image column <-> hidden index (an involution)

// ------------------------------------------------------------------------------------------------------------------ optimizer
// Reduction + Adam + weight images in ONE pass over the parameters (single-process training: no gradient all-reduce sits between
// them): a thread owns one parameter element - sums its split-K partials, writes the gradient (inspection / parity tests), updates the
// float32 moments and the master weight exactly as torch.optim.Adam does (no weight decay, no amsgrad: exp_avg.lerp_, exp_avg_sq
// mul_/addcmul_, step_size = lr / bias_correction1, denom = sqrt(v) / sqrt(bias_correction2) + eps), and re-emits the element's float16
// copies in the forward and backward weight images.  bias corrections come from adam_tick_kernel (device-resident step count: the
// step is replayable from a captured graph).
struct AdamNet {
    float* w1; float* b1; float* w2; float* b2; float* w3; float* b3;          // float32 masters, updated in place
    Grads g;                                                                       // gradients out (torch layouts)
    float* m; float* v;                                                            // moments, flat: [w2 | b2 | w1 | b1 | w3 | b3]
    uint16_t* w23; uint16_t* w2t; uint16_t* w3t;                                   // weight images (ImgNet)
};

struct AdamHyper { float lr, beta1, beta2, eps; };

// One small block per optimizer step: advances the device-resident step count and the minibatch cursor, publishes the two bias corrections, and folds the loss
// kernel's per-block statistics (float[nblocks][5] sums over 256 samples each) into the running sums stats_acc[5] += sum / batch -
// what three torch launches (sum, scale, accumulate) did per SGD step.
__global__ void __launch_bounds__(64)
adam_tick_kernel(long long* step, float* bc, float beta1, float beta2, const float* __restrict__ stats_partials, int nblocks, float inv_batch,
                 float* stats_acc, long long* idx_cursor, long long minibatch) {
    const uint32_t lane = threadIdx.x;
    if (stats_partials) {
        float s[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int b = (int)lane; b < nblocks; b += 64)
#pragma unroll
            for (int k = 0; k < 5; ++k) s[k] += stats_partials[(size_t)b * 5 + k];
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off, 64);
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 5; ++k) stats_acc[k] += s[k] * inv_batch;
    }
    if (lane == 0) {
        const long long t = *step + 1;
        *step = t;
        *idx_cursor += minibatch;                        // the next step's minibatch starts where this one ended (callers that pass no cursor ignore it)
        bc[0] = (float)(1.0 - pow((double)beta1, (double)t));
        bc[1] = (float)(1.0 - pow((double)beta2, (double)t));
    }
}

__device__ __forceinline__ float adam_update(float w, float g, float& m, float& v, const AdamHyper& hp, float bc1, float bc2) {
    m = m + (g - m) * (1.0f - hp.beta1);
    v = v * hp.beta2 + (1.0f - hp.beta2) * g * g;
    const float denom = sqrtf(v) / sqrtf(bc2) + hp.eps;
    return w - (hp.lr / bc1) * (m / denom);
}

// The fused SGD step's bookkeeping (q1env_learner_sgd_step): the bias corrections were left by the backward kernel (BcArgs), so nothing
// here depends on the step count - workgroup (0, 0) of the Adam kernel advances it and the minibatch cursor (their readers, the
// forward / backward kernels of this step, are done) and folds the backward kernel's statistics rows into the running sums:
// adam_tick_kernel's duties without its launch.  step == NULL: the stand-alone q1env_learner_adam, which runs adam_tick_kernel first.
struct AdamTick {
    long long* step; long long* idx_cursor; long long minibatch;
    const float* stats_rows; int nrows; float inv_batch; float* stats_acc;
    int dw1;                  // the slab's products 72..79 are in the fused kernel's orientation (slot_of)
};

__global__ void __launch_bounds__(256)
learner_adam_kernel(const float* __restrict__ pa, const float* __restrict__ pb, AdamNet na, AdamNet nb, int splits, float inv_scale_a,
                    float inv_scale_b, AdamHyper hp, const float* __restrict__ bc, AdamTick tk) {
    if (tk.step && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x >= 192u) {           // the workgroup's last wave: after its own element
        const uint32_t lane = threadIdx.x - 192u;
        if (tk.stats_rows) {
            float s[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            for (int b = (int)lane; b < tk.nrows; b += 64)
#pragma unroll
                for (int k = 0; k < 5; ++k) s[k] += tk.stats_rows[(size_t)b * 5 + k];
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off, 64);
            if (lane == 0)
#pragma unroll
                for (int k = 0; k < 5; ++k) tk.stats_acc[k] += s[k] * tk.inv_batch;
        }
        if (lane == 0) { *tk.step += 1; *tk.idx_cursor += tk.minibatch; }
    }
    // a thread owns one slot of the partial-sum slab (slot_of / slab_sum above)
    const bool second = blockIdx.y == 1;
    const float inv_scale = second ? inv_scale_b : inv_scale_a;
    const float* __restrict__ partial = second ? pb : pa;
    const AdamNet net = second ? nb : na;
    const uint32_t off = blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= (uint32_t)PARTIAL_FLOATS) return;
    const Slot sl = slot_of(off, (uint32_t)net.g.out_dim, tk.dw1 != 0);
    if (sl.arr < 0) return;
    float* const wp = slot_ptr(sl, net.w2, net.b2, net.w1, net.b1, net.w3, net.b3);
    float* const gp = slot_ptr(sl, net.g.w2, net.g.b2, net.g.w1, net.g.b1, net.g.w3, net.g.b3);
    const float bc1 = bc[0], bc2 = bc[1];
    float m = net.m[sl.e], v = net.v[sl.e];
    const float g = inv_scale * slab_sum(partial, splits, off);
    const float w = adam_update(*wp, g, m, v, hp, bc1, bc2);
    *gp = g; *wp = w;
    net.m[sl.e] = m; net.v[sl.e] = v;
    if (sl.arr == 0) {
        net.w23[sl.x * 264u + kperm(sl.y)] = q1pol::f16_bits(q1pol::TANH_PRESCALE * w);
        net.w2t[sl.y * 264u + kperm(sl.x)] = q1pol::f16_bits(w);
    } else if (sl.arr == 4) {
        net.w23[(256u + sl.x) * 264u + kperm(sl.y)] = q1pol::f16_bits(w);
        net.w3t[sl.y * 40u + sl.x] = q1pol::f16_bits(w);
    }
}

// ------------------------------------------------------------------------------------------------------------------ weight images
struct ImgNet {
    const float* w2; const float* w3; int out_dim;       // float32 masters, torch layouts
    uint16_t* w23;            // forward image float16[288][264] (q1policy.hpp): W2 x 2 log2 e, then W3, K index permuted
    uint16_t* w2t;            // backward image float16[256][264]: W2^T, K index (j) permuted the same way
    uint16_t* w3t;            // backward image float16[256][40]:  W3^T, natural order
};

__global__ void __launch_bounds__(256)
learner_images_kernel(ImgNet na, ImgNet nb) {
    const ImgNet net = blockIdx.y == 1 ? nb : na;
    const uint32_t OUT = (uint32_t)net.out_dim;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nF = 288u * 264u, nT2 = 256u * 264u, nT3 = 256u * 40u;
    if (i < nF) {
        const uint32_t row = i / 264u, p = i % 264u;
        float v = 0.0f;
        if (p < 256u) {
            if (row < 256u) v = q1pol::TANH_PRESCALE * net.w2[row * 256u + kperm(p)];
            else if (row - 256u < OUT) v = net.w3[(row - 256u) * 256u + kperm(p)];
        }
        net.w23[i] = q1pol::f16_bits(v);
        return;
    }
    i -= nF;
    if (i < nT2) {
        const uint32_t k = i / 264u, p = i % 264u;
        net.w2t[i] = q1pol::f16_bits(p < 256u ? net.w2[kperm(p) * 256u + k] : 0.0f);
        return;
    }
    i -= nT2;
    if (i < nT3) {
        const uint32_t j = i / 40u, o = i % 40u;
        net.w3t[i] = q1pol::f16_bits(o < OUT ? net.w3[o * 256u + j] : 0.0f);
    }
}

}  // namespace q1learn
