// q1policy.hpp - fused forward pass of one policy / value network of the reference's shape (RLlib fcnet of the published
// checkpoint, SURVEY.md section 2: obs 6 -> 256 tanh -> 256 tanh -> OUT, OUT = 10 logits or 1 value) for gfx950; one launch
// evaluates one network or the policy AND the value network (half of the CUs each).
//
// Operand precision: float16, not bfloat16 - activations live in [-1, 1] and the weights of such a policy are O(1), so f16's
// 11-bit significand costs nothing in range, runs at the same MFMA rate and packs with the same single convert instruction,
// and it makes the behaviour policy 8x closer to the float32 learner policy (matters for PPO: DESIGN.md section 8).
//
// This is the GEMM-shaped neighbour of the env hot path (the sampler tick is: this, then q1env_sample_step), so it is the
// one place matrix cores are used - for all three layers, computed TRANSPOSED
// (hidden units are MFMA rows, envs are MFMA columns), so that the output of one layer is already distributed the way the
// next layer's B operand needs it and activations never leave registers:
//   layer 1  H1^T = W1b . (Xhi + Xlo)^T  ONE v_mfma_f32_32x32x16_f16 per 32-row tile: f16 weights, inputs and bias split into
//                                       two f16 each (hi + lo = 22 mantissa bits), K = 16 = 2 x (6 inputs + bias hi/lo)
//   layer 2  H2^T = W2 . tanh(H1)^T     v_mfma_f32_32x32x16_f16, float32 accumulate, accumulators start at the bias b2
//   layer 3  Y^T  = W3 . tanh(H2)^T     v_mfma_f32_32x32x16_f16 (rows = outputs, padded to 32), + b3 in float32
// The C/D register layout of a 32x32 tile gives lane (c, h) the rows (r&3) + 8(r>>2) + 4h, r = 0..15, of column c, while a
// B operand wants 8 consecutive K indices per lane.  Instead of shuffling activations, the K index of the NEXT layer's
// weights is permuted once, when they are staged into LDS: within every 16 hidden units the four groups of four are stored
// in the order (0, 2, 1, 3).  Then lane (c, h)'s accumulator registers 8u..8u+7 ARE its B operand of K-step 2t + u.
// One workgroup per CU keeps W2 / W3 (f16, rows padded by 16 B so the 16-byte operand reads are bank-conflict free), the
// layer-1 operand image and b2 in LDS - 157.5 KB of the CU's 160 KB; every wave walks its own 32-env tiles grid-stride, so the
// weights are fetched once per CU, not once per tile.  tanh = 1 - 2/(2^(2x log2 e) + 1) on v_exp_f32 / v_rcp_f32, the factor
// 2 log2 e folded into the producing layer's weights.
// Measured on MI355X (tools/trace_mlp.py, gpurun_scratch micro-benchmarks): under this load the shader
// clock settles at ~1.6 GHz; a tile costs ~7 600 cycles per SIMD against a VALU floor of ~6 500 (512 tanh per lane: 4 250 cycles
// of v_exp/v_rcp at 8.3 cycles each, the rest packed float32 and f16 converts) and an MFMA floor of 4 864 (152 x 32 cycles).
// What mattered, in order: accumulators in VGPRs (a 256-register budget; with 512 the compiler parks them in AGPRs and pays a
// v_accvgpr_read per tanh input), layer 1 as one f16 MFMA instead of four dependent float32 ones, MFMA / VALU alternation in the
// instruction stream (one wave issues in order), both networks in one launch (weight staging overlaps).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace q1pol {

constexpr int HID = 256;
constexpr int OBS = 6;
constexpr int ROW_BYTES = HID * 2 + 16;               // 528: padded row stride of the f16 weight rows in LDS
constexpr size_t LDS_W2 = (size_t)HID * ROW_BYTES;    // 135168
constexpr size_t LDS_W3 = (size_t)32 * ROW_BYTES;     // 16896 (output rows padded to one 32-row tile)
constexpr size_t LDS_B2 = (size_t)HID * 4;            // 1024
constexpr size_t LDS_W1 = (size_t)HID * 32;           // 8192: layer-1 operand image, [k][half][8 K slots] f16
constexpr size_t LDS_TOTAL = LDS_W2 + LDS_W3 + LDS_B2 + LDS_W1;    // 161280 B <= 163840

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// tanh(z) = 1 - 2 / (2^(c z) + 1) with c = 2 log2(e).  The factor c is folded into the weights and biases of the layer that
// PRODUCES z (W1, b1 here at staging; the W2 block of the host-built image; b2 at staging), so the accumulators already hold c z
// and an activation is v_exp_f32, +1, v_rcp_f32, 1 - 2r: one packed multiply less per pair of values on the VALU that bounds
// this kernel.
constexpr float TANH_PRESCALE = 2.8853900817779268f;                 // 2 log2(e)

// tanh of accumulator registers 8u .. 8u+7 (holding c z) -> the lane's f16 B operand of K-step 2t + u.  Two values at a time:
// the +1 and the final 1 - 2r are packed float32 instructions (v_pk_add/fma_f32), only v_exp_f32 and v_rcp_f32 are per
// element; the pair is converted with one v_cvt_pk_f16_f32 (RNE).
__device__ __forceinline__ f16x8 activate(const f32x16& acc, int u) {
    union { f16x8 v; f16x2 p[4]; } o;
#ifdef Q1POL_ACT_PK16
    // EXPERIMENT (round 6; profiles/r6_policy_tanh_experiment.txt): the exponentials stay float32, everything behind them runs
    // in packed float16 - convert the pair, + 1 (v_pk_add_f16), two v_rcp_f16, 1 - 2 r (v_pk_fma_f16) - so that the result IS the operand pair.
    // Not the product form: compiled only with -DQ1POL_ACT_PK16 (tools/r6_policy_tanh.sh builds libq1env_pk16.so and times / checks it).
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 t = {__builtin_amdgcn_exp2f(acc[8 * u + 2 * j]), __builtin_amdgcn_exp2f(acc[8 * u + 2 * j + 1])};
        const f32x2 tc = {fminf(t[0], 65504.0f), fminf(t[1], 65504.0f)};      // (2^(c z) overflows float16 from z = 5.5: tanh = 1 - 2 / 65505 there)
        f16x2 e = __builtin_convertvector(tc, f16x2);
        e = e + (f16x2){(_Float16)1.0f, (_Float16)1.0f};
        const f16x2 r = {(_Float16)__builtin_amdgcn_rcph(e[0]), (_Float16)__builtin_amdgcn_rcph(e[1])};
        o.p[j] = __builtin_elementwise_fma((f16x2){(_Float16)-2.0f, (_Float16)-2.0f}, r, (f16x2){(_Float16)1.0f, (_Float16)1.0f});
    }
    return o.v;
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 t = {__builtin_amdgcn_exp2f(acc[8 * u + 2 * j]), __builtin_amdgcn_exp2f(acc[8 * u + 2 * j + 1])};
        t = t + 1.0f;
        const f32x2 r = {__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
        const f32x2 y = 1.0f - 2.0f * r;
        o.p[j] = __builtin_convertvector(y, f16x2);
    }
    return o.v;
}

// Layer 1 on the f16 matrix path with SPLIT inputs: x = hi + lo (two f16, 22 mantissa bits together), so ONE
// v_mfma_f32_32x32x16_f16 computes W1b . (x_hi + x_lo) + (b1_hi + b1_lo) for a 32-row tile (the float32 32x32x2 path it replaces
// needed four dependent 16-pass MFMAs per tile).  K layout of half h: slots 0..3 = hi of inputs h, 2+h, 4+h, 6+h, slots 4..7 = lo
// of the same inputs; inputs 6 and 7 are the constant 1 carrying f16(b1) and f16(b1 - f16(b1)).
__device__ __forceinline__ uint16_t f16_bits(float x) {               // RNE, like v_cvt_pk_f16_f32
    union { _Float16 h; uint16_t u; } o;
    o.h = (_Float16)x;
    return o.u;
}
__device__ __forceinline__ float f16_value(uint16_t b) {
    union { _Float16 h; uint16_t u; } o;
    o.u = b;
    return (float)o.h;
}

__device__ __forceinline__ f16x8 split_inputs(const float (&x)[3], uint32_t half) {
    (void)half;
    union { f16x8 v; uint16_t u[8]; } o;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        // (observations are O(1..1e3); the clamp keeps an out-of-range input finite - f16 saturates at 65504 - instead of inf/NaN)
        const float xs = fminf(fmaxf(x[s], -65504.0f), 65504.0f);
        const uint16_t hi = f16_bits(xs);
        o.u[s] = hi;
        o.u[4 + s] = f16_bits(xs - f16_value(hi));
    }
    o.u[3] = 0x3C00;                                                  // 1.0: bias slot (hi part for half 0, lo part for half 1)
    o.u[7] = 0;
    return o.v;
}

// row k of the layer-1 operand image: 16 f16 = [half][8 K slots] (32 B), built from float w1[k][0..5], b1[k], pre-scaled by c
__device__ __forceinline__ void stage_w1_row(unsigned char* l_w1, uint32_t k, const float* __restrict__ w1, const float* __restrict__ b1) {
    union { uint4 q[2]; uint16_t u[16]; } r;
    const float bs = TANH_PRESCALE * b1[k];
    const uint16_t bhi = f16_bits(bs);
    const uint16_t blo = f16_bits(bs - f16_value(bhi));
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 2 * j + h;                                  // input index of slot j in half h
            const uint16_t w = i < OBS ? f16_bits(TANH_PRESCALE * w1[k * OBS + i]) : (i == 6 ? bhi : blo);
            r.u[8 * h + j] = w;
            r.u[8 * h + 4 + j] = i < OBS ? w : (uint16_t)0;           // lo parts of the inputs meet the same weight; lo(1) = 0
        }
    uint4* dst = reinterpret_cast<uint4*>(l_w1 + (size_t)k * 32u);
    dst[0] = r.q[0]; dst[1] = r.q[1];
}

// The host hands W2 (times 2 log2 e, see TANH_PRESCALE) and W3 over as ONE f16 image that is already in LDS layout: (256 + 32) rows of 264 elements (528 B:
// 256 weights + 8 pad), columns of every row permuted (groups of four within each 16: 0,2,1,3).  Staging is then a straight
// 16-byte-per-lane copy of 152 064 bytes with all of a thread's loads in flight at once.
constexpr uint32_t IMG_VEC16 = (uint32_t)((LDS_W2 + LDS_W3) / 16);   // 9504 uint4
template <int THREADS>
__device__ __forceinline__ void stage_image(unsigned char* dst, const uint16_t* __restrict__ img, uint32_t tid) {
    const uint4* src = reinterpret_cast<const uint4*>(img);
    uint4* d = reinterpret_cast<uint4*>(dst);
    constexpr uint32_t NT = (uint32_t)THREADS;
    constexpr uint32_t PER = (IMG_VEC16 + NT - 1u) / NT;               // 19 (512 threads) or 38 vectors per thread
    uint4 v[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t c = k * NT + tid;
        v[k] = c < IMG_VEC16 ? src[c] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t c = k * NT + tid;
        if (c < IMG_VEC16) d[c] = v[k];
    }
}

// LDS carve-up of one network (LDS_TOTAL bytes from `lds`) and the per-lane operand row pointers of a wave
struct LdsNet {
    unsigned char* w2; unsigned char* w3; float* b2; unsigned char* w1;
};
__device__ __forceinline__ LdsNet lds_net(unsigned char* lds) {
    return LdsNet{lds, lds + LDS_W2, reinterpret_cast<float*>(lds + LDS_W2 + LDS_W3), lds + LDS_W2 + LDS_W3 + LDS_B2};
}

// stage one network's weights into LDS (all THREADS threads of the workgroup; the caller synchronises afterwards)
template <int THREADS>
__device__ __forceinline__ void stage_net(unsigned char* lds, const float* __restrict__ w1, const float* __restrict__ b1,
                                          const uint16_t* __restrict__ w23, const float* __restrict__ b2, uint32_t tid) {
    const LdsNet l = lds_net(lds);
    stage_image<THREADS>(lds, w23, tid);                             // W2 rows then W3 rows, contiguous in LDS
    if (tid < (uint32_t)HID) {
        l.b2[tid] = TANH_PRESCALE * b2[tid];
        stage_w1_row(l.w1, tid, w1, b1);
    }
}

// One 32-env tile through all three layers: xb = the lane's layer-1 B operand (split_inputs of its env's observation), the three
// row pointers = this lane's operand rows in the LDS images (see the callers).  Returns Y^T without b3: y[r] = output row
// (r&3) + 8(r>>2) + 4*half of env `col`.  (stamps: diagnostic build only - s_memtime after the prologue and after the main loop.)
// STORE (the learner's forward, q1learner.hpp): the activations leave as they are consumed - h1_dst / h2_dst point at THIS lane's slot
// of the tile's T-format arrays (f16x8 units; element (t, u) at [(2 t + u) * 64]): tanh(H1) / tanh(H2) as the very B operands the next
// layer's MFMAs read, 32 fully coalesced 16-byte stores per layer and tile.
// cache policy of an activation store (H1POL of mlp_tile_t): 0 = plain (write-back in the XCD's L2), 1 = sc1, 2 = nt, 3 = sc0 sc1
template <int POL>
__device__ __forceinline__ void act_store(f16x8* p, const f16x8 v) {
    if constexpr (POL == 0) {
        *p = v;
    } else {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        union { f16x8 h; u32x4 u; } o;
        o.h = v;
        if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(o.u) : "memory");
        else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(o.u) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(o.u) : "memory");
    }
}

// KEEP (the fused forward + backward kernel of the learner, q1learner_fused.hpp): tanh(H2) additionally stays with the caller, in h2k[t][u] - the
// operand vectors the data-gradient phase multiplies by (1 - h2^2) a few microseconds later, without a trip through memory.
// STORE_H2 = false (with STORE): only tanh(H1) is stored - the fused kernel's product mode needs tanh(H2) nowhere else.
template <bool STORE, bool KEEP = false, bool STORE_H2 = STORE, int H1POL = 0>
__device__ __forceinline__ f32x16 mlp_tile_t(const f16x8 xb, const unsigned char* w1row, const unsigned char* wrow, const unsigned char* w3row,
                                             const float* l_b2, uint32_t half, uint64_t* stamps, f16x8* h1_dst, f16x8* h2_dst, f16x8 (*h2k)[2] = nullptr) {
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[8];                                               // H2^T pre-activations, start at b2
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(l_b2 + t * 32 + 8 * q + 4 * half);   // rows 8q + 4h + (0..3)
            acc[t][4 * q + 0] = b.x; acc[t][4 * q + 1] = b.y; acc[t][4 * q + 2] = b.z; acc[t][4 * q + 3] = b.w;
        }

    // Software pipeline over the eight 32-row tiles of H1: tanh of tile t1+1 is interleaved (sched_group_barrier) with the
    // sixteen MFMAs of tile t1; every W2 operand register is re-requested from LDS for the next K-step right after the
    // MFMA that consumed it has been issued (operands are read at issue), i.e. eight MFMAs ahead of its next use.
    f16x8 cur0, cur1, a_l1, a[8];
    f32x16 dn;
    {
        const f16x8 a0 = *reinterpret_cast<const f16x8*>(w1row);
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(w1row + 1024u);
        a_l1 = *reinterpret_cast<const f16x8*>(w1row + 2048u);
        const f32x16 d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xb, zero16, 0, 0, 0);
        dn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xb, zero16, 0, 0, 0);
#pragma unroll
        for (int t2 = 0; t2 < 8; ++t2) a[t2] = *reinterpret_cast<const f16x8*>(wrow + (size_t)t2 * 32u * ROW_BYTES);
        cur0 = activate(d0, 0); cur1 = activate(d0, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (stamps) stamps[0] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int t1 = 0; t1 < 8; ++t1) {
        const uint32_t q0 = 2u * (uint32_t)t1;
        if constexpr (STORE) { act_store<H1POL>(h1_dst + (2 * t1) * 64, cur0); act_store<H1POL>(h1_dst + (2 * t1 + 1) * 64, cur1); }
        // phase 1: even K-step, first half of tanh(tile t1+1)
#pragma unroll
        for (int t2 = 0; t2 < 8; ++t2) {
            acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t2], cur0, acc[t2], 0, 0, 0);
            a[t2] = *reinterpret_cast<const f16x8*>(wrow + (size_t)t2 * 32u * ROW_BYTES + (q0 + 1u) * 32u);
        }
        const f16x8 nxt0 = activate(dn, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // phase 2: odd K-step, second half of the tanh, then layer 1 of tile t1+2 (one MFMA)
        const f16x8 a_l1n = *reinterpret_cast<const f16x8*>(w1row + (((uint32_t)t1 + 3u) & 7u) * 1024u);
#pragma unroll
        for (int t2 = 0; t2 < 8; ++t2) {
            acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t2], cur1, acc[t2], 0, 0, 0);
            a[t2] = *reinterpret_cast<const f16x8*>(wrow + (size_t)t2 * 32u * ROW_BYTES + ((q0 + 2u) & 15u) * 32u);
        }
        const f16x8 nxt1 = activate(dn, 1);
        dn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l1, xb, zero16, 0, 0, 0);   // tile t1 + 2 (the last two passes wrap around, unused)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur0 = nxt0; cur1 = nxt1; a_l1 = a_l1n;
    }
    if (stamps) stamps[1] = __builtin_amdgcn_s_memtime();

    // layer 3: tanh(H2 tile) and its two K-steps, W3 operands requested two tiles ahead
    f32x16 y = zero16;
    f16x8 w3a = *reinterpret_cast<const f16x8*>(w3row), w3b = *reinterpret_cast<const f16x8*>(w3row + 32u);
#pragma unroll
    for (int t2 = 0; t2 < 8; ++t2) {
        const f16x8 f0 = activate(acc[t2], 0), f1 = activate(acc[t2], 1);
        if constexpr (STORE_H2) { h2_dst[(2 * t2) * 64] = f0; h2_dst[(2 * t2 + 1) * 64] = f1; }
        if constexpr (KEEP) { h2k[t2][0] = f0; h2k[t2][1] = f1; }
        y = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3a, f0, y, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3b, f1, y, 0, 0, 0);
        if (t2 < 7) {
            w3a = *reinterpret_cast<const f16x8*>(w3row + (uint32_t)(2 * t2 + 2) * 32u);
            w3b = *reinterpret_cast<const f16x8*>(w3row + (uint32_t)(2 * t2 + 3) * 32u);
        }
    }
    return y;
}

__device__ __forceinline__ f32x16 mlp_tile(const f16x8 xb, const unsigned char* w1row, const unsigned char* wrow, const unsigned char* w3row,
                                           const float* l_b2, uint32_t half, uint64_t* stamps) {
    return mlp_tile_t<false>(xb, w1row, wrow, w3row, l_b2, half, stamps, nullptr, nullptr);
}

// w1: float[HID][OBS] (torch Linear(6,256).weight), b1: float[HID], w23: the f16 LDS image of W2 (Linear(256,256).weight) and W3
// (Linear(256,out).weight in rows 0..out-1 of a 32-row tile) described above, b2: float[HID], b3: float[out_dim];
// obs float[n][6]; out float[n][out_dim].
// THREADS = 512: 8 waves, TWO per SIMD, each walking its own 32-env tiles (large batches); THREADS = 256: one wave per SIMD
// (batches with at most one tile per SIMD, where a second wave would have nothing to do and the register budget is better spent).
struct Net {                                         // one network: device pointers (see q1env_policy_forward)
    const float* w1; const float* b1; const uint16_t* w23; const float* b2; const float* b3;
    float* out; int out_dim;
};

template <int THREADS>
// (register budget of 256 per lane in both variants - 512 threads x 1 block or 256 threads x "2" blocks: with more the compiler
// parks the accumulators in AGPRs and every tanh input costs a v_accvgpr_read; LDS admits one block per CU either way)
__global__ void __launch_bounds__(THREADS, 512 / THREADS)
mlp_forward_kernel(int n, const float* __restrict__ obs, Net net_a, Net net_b, int nets) {
    // nets == 2: the first half of the grid evaluates net_a, the second half net_b, over the same observations - the policy
    // and the value network of one sampler tick in ONE launch (weight staging of both overlaps, half the launches).
    const uint32_t bgrid = nets == 2 ? gridDim.x / 2u : gridDim.x;
    const bool second = nets == 2 && blockIdx.x >= bgrid;
    const uint32_t bid = second ? blockIdx.x - bgrid : blockIdx.x;
    const Net net = second ? net_b : net_a;
    const float* __restrict__ w1 = net.w1; const float* __restrict__ b1 = net.b1; const uint16_t* __restrict__ w23 = net.w23;
    const float* __restrict__ b2 = net.b2; const float* __restrict__ b3 = net.b3; float* __restrict__ out = net.out;
    const int OUT = net.out_dim;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* l_w2 = lds;
    unsigned char* l_w3 = lds + LDS_W2;
    float* l_b2 = reinterpret_cast<float*>(lds + LDS_W2 + LDS_W3);
    unsigned char* l_w1 = lds + LDS_W2 + LDS_W3 + LDS_B2;           // layer-1 operand image: [k][half][8] f16

    const uint32_t tid = threadIdx.x;
    stage_image<THREADS>(lds, w23, tid);                             // W2 rows then W3 rows, contiguous in LDS
    if (tid < (uint32_t)HID) {
        l_b2[tid] = TANH_PRESCALE * b2[tid];
        stage_w1_row(l_w1, tid, w1, b1);
    }
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, half = lane >> 5;
    __syncthreads();

    // Every wave owns whole 32-env tiles (tile = 32 consecutive envs), grid-stride.  Two waves share a SIMD: while one is in
    // a VALU stretch (tanh) or waits for LDS, the other one's MFMAs keep the matrix pipe busy - with a single wave per SIMD
    // the in-order issue serialises the two pipes unless the instruction stream alternates perfectly.
    const uint32_t ntiles = ((uint32_t)n + 31u) / 32u;
    const uint32_t tstride = bgrid * (uint32_t)(THREADS / 64);
    uint32_t tile = bid * (uint32_t)(THREADS / 64) + wave;
    // observations of the NEXT tile are requested while the current one is computed
    float xn[3];
    {
        const uint32_t e0 = tile * 32u + col;
#pragma unroll
        for (int s = 0; s < 3; ++s) xn[s] = (tile < ntiles && e0 < (uint32_t)n) ? obs[(size_t)e0 * OBS + 2u * s + half] : 0.0f;
    }
    const unsigned char* w1row = l_w1 + (size_t)col * 32u + half * 16u;               // + tile * 1024
    const unsigned char* wrow = l_w2 + (size_t)col * ROW_BYTES + half * 16u;          // + t2 * 32 rows, + K-step * 32 B
    const unsigned char* w3row = l_w3 + (size_t)col * ROW_BYTES + half * 16u;         // + K-step * 32 B
#ifdef Q1POL_TRACE                                                   // diagnostic build (tools/trace_mlp.py): phase times of wave 0
    uint64_t tr_pro = 0, tr_loop = 0, tr_l3 = 0, tr_chunks = 0;
    const uint64_t tr_real0 = wall_clock64(), tr_mem0 = __builtin_amdgcn_s_memtime();   // 100 MHz reference vs s_memtime
#define Q1POL_STAMP(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#else
#define Q1POL_STAMP(var)
#endif
    for (; tile < ntiles; tile += tstride) {
        const uint32_t env = tile * 32u + col;
        const bool live = env < (uint32_t)n;
        Q1POL_STAMP(ts0);
        const f16x8 xb = split_inputs(xn, half);                    // B operand of layer 1
        {
            const uint32_t en = env + tstride * 32u;
            const bool more = tile + tstride < ntiles && en < (uint32_t)n;
#pragma unroll
            for (int s = 0; s < 3; ++s) xn[s] = more ? obs[(size_t)en * OBS + 2u * s + half] : 0.0f;
        }

#ifdef Q1POL_TRACE
        uint64_t stamps[2];
        const f32x16 y = mlp_tile(xb, w1row, wrow, w3row, l_b2, half, stamps);
        const uint64_t ts1 = stamps[0], ts2 = stamps[1];
#else
        const f32x16 y = mlp_tile(xb, w1row, wrow, w3row, l_b2, half, nullptr);
#endif
        // y[r] = output row (r&3) + 8(r>>2) + 4*half of env `col`; OUT <= 32 rows are real (10 policy logits with the continuous
        // mouse, 2K + 2S+1 with a discrete one, 1 for the value net).  OUT is wave-uniform: the group test is a scalar branch.
        if (live) {
            float* dst = out + (size_t)env * (uint32_t)OUT;
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
#ifdef Q1POL_TRACE
        __builtin_amdgcn_sched_barrier(0);
        const uint64_t ts3 = __builtin_amdgcn_s_memtime();
        tr_pro += ts1 - ts0; tr_loop += ts2 - ts1; tr_l3 += ts3 - ts2; tr_chunks += 1;
#endif
    }
#ifdef Q1POL_TRACE
    if (blockIdx.x == 0 && tid == 0) {                  // s_memtime ticks summed over this wave's tiles
        out[0] = (float)tr_pro; out[1] = (float)tr_loop; out[2] = (float)tr_l3; out[3] = (float)tr_chunks;
        out[4] = (float)(wall_clock64() - tr_real0); out[5] = (float)(__builtin_amdgcn_s_memtime() - tr_mem0);
    }
#endif
}

}  // namespace q1pol
