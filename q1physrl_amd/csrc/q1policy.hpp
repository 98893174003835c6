// q1policy.hpp - fused forward pass of one policy / value network of the reference's shape (RLlib fcnet of the published
// checkpoint, SURVEY.md section 2: obs 6 -> 256 tanh -> 256 tanh -> OUT, OUT = 10 logits or 1 value) for gfx950.
//
// This is the GEMM-shaped neighbour of the env hot path (the sampler tick is: this, q1env_policy_sample, q1env_step,
// q1env_reset_philox), so it is the one place matrix cores are used:
//   layer 1 (K = 6)    float32 VALU, computed on the fly directly in the MFMA operand layout, tanh, rounded to bf16
//   layer 2 (256x256)  v_mfma_f32_32x32x16_bf16, float32 accumulate; computed TRANSPOSED (H2^T = W2 . H1^T) so that after the
//                      MFMAs each lane owns one env (column) and 128 of its 256 hidden units (rows)
//   layer 3 (K = 256)  float32 VALU on the accumulator registers + one cross-half shuffle; no LDS round trip of H2
// One workgroup (4 waves, one per SIMD) keeps the whole network in LDS - W2 as bf16 [n][k] with rows padded to 528 B so the
// 16-byte operand reads are bank-conflict free, W1 / biases / W3 as float32: 156 KB of the CU's 160 KB - and walks
// 128-env chunks grid-stride, so the weights are fetched once per CU, not once per chunk.
//
// MFMA operand layout used (v_mfma_f32_32x32x16_bf16): A: lane l holds A[row = l & 31][k = 8*(l >> 5) + j], j = 0..7;
// B: lane l holds B[k = 8*(l >> 5) + j][col = l & 31]; C/D: col = l & 31, row = (r & 3) + 8*(r >> 2) + 4*(l >> 5), r = 0..15.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace q1pol {

constexpr int HID = 256;
constexpr int OBS = 6;
constexpr int W2_ROW_BYTES = HID * 2 + 16;            // 528: padded row stride of the bf16 W2 copy in LDS
constexpr int MAX_OUT = 12;                           // W3 rows are padded to 12 floats (3 x 16 B) in LDS
constexpr size_t LDS_W2 = (size_t)HID * W2_ROW_BYTES;                // 135168
constexpr size_t LDS_W1 = (size_t)HID * 8 * 4;                       // [k][8]: 6 weights + bias + pad = 8192
constexpr size_t LDS_B2 = (size_t)HID * 4;                           // 1024
constexpr size_t LDS_W3 = (size_t)HID * MAX_OUT * 4;                 // 12288
constexpr size_t LDS_TOTAL = LDS_W2 + LDS_W1 + LDS_B2 + LDS_W3 + 64; // 156736 B <= 163840

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {      // round-to-nearest-even, finite inputs
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ float fast_tanh(float x) {                // 1 - 2 / (e^{2x} + 1): exact limits at +-inf
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f); // e^{2x} = 2^{2x log2 e}
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// w1: float[HID][OBS] (torch Linear(6,256).weight), b1: float[HID], w2: bf16 bits [HID n][HID k] (Linear(256,256).weight),
// b2: float[HID], w3: float[out_dim][HID] (Linear(256,out).weight), b3: float[out_dim]; obs float[n][6]; out float[n][out_dim].
template <int OUT>
__global__ void __launch_bounds__(256, 1)
mlp_forward_kernel(int n, const float* __restrict__ obs, const float* __restrict__ w1, const float* __restrict__ b1,
                   const uint16_t* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
                   const float* __restrict__ b3, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* l_w2 = lds;
    float* l_w1 = reinterpret_cast<float*>(lds + LDS_W2);            // [k][8] = w1[k][0..5], b1[k], 0
    float* l_b2 = reinterpret_cast<float*>(lds + LDS_W2 + LDS_W1);
    float* l_w3 = reinterpret_cast<float*>(lds + LDS_W2 + LDS_W1 + LDS_B2);   // [n][MAX_OUT]
    float* l_b3 = reinterpret_cast<float*>(lds + LDS_W2 + LDS_W1 + LDS_B2 + LDS_W3);

    // ---- stage the network into LDS once per workgroup (16-byte global loads, coalesced)
    const uint32_t tid = threadIdx.x;
    for (uint32_t c = tid; c < HID * (HID * 2 / 16); c += 256) {     // 256 rows x 32 chunks of 16 B
        const uint32_t row = c >> 5, ch = c & 31u;
        const uint4 v = reinterpret_cast<const uint4*>(w2)[c];
        *reinterpret_cast<uint4*>(l_w2 + (size_t)row * W2_ROW_BYTES + ch * 16) = v;
    }
    {
        const uint32_t k = tid;                                       // HID == blockDim.x == 256
        float4 lo = make_float4(w1[k * OBS + 0], w1[k * OBS + 1], w1[k * OBS + 2], w1[k * OBS + 3]);
        float4 hi = make_float4(w1[k * OBS + 4], w1[k * OBS + 5], b1[k], 0.0f);
        reinterpret_cast<float4*>(l_w1)[k * 2] = lo;
        reinterpret_cast<float4*>(l_w1)[k * 2 + 1] = hi;
        l_b2[k] = b2[k];
#pragma unroll
        for (int o = 0; o < MAX_OUT; ++o) l_w3[k * MAX_OUT + o] = (o < OUT) ? w3[(size_t)o * HID + k] : 0.0f;
        if (k < MAX_OUT) l_b3[k] = (k < (uint32_t)OUT) ? b3[k] : 0.0f;
    }
    __syncthreads();

    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, half = lane >> 5;
    const uint32_t nchunks = ((uint32_t)n + 127u) / 128u;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t env = chunk * 128u + wave * 32u + col;
        const bool live = env < (uint32_t)n;
        float x[OBS];
#pragma unroll
        for (int i = 0; i < OBS; ++i) x[i] = live ? obs[(size_t)env * OBS + i] : 0.0f;

        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

#pragma unroll 2
        for (int kk = 0; kk < HID / 16; ++kk) {
            // layer 1 for this lane's env and its 8 hidden units k = 16*kk + 8*half + j, straight into the B operand
            bf16x8 bfrag;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = (uint32_t)kk * 16u + half * 8u + (uint32_t)j;
                const float4 lo = reinterpret_cast<const float4*>(l_w1)[k * 2];
                const float4 hi = reinterpret_cast<const float4*>(l_w1)[k * 2 + 1];
                float s = hi.z;
                s = fmaf(x[0], lo.x, s); s = fmaf(x[1], lo.y, s); s = fmaf(x[2], lo.z, s); s = fmaf(x[3], lo.w, s);
                s = fmaf(x[4], hi.x, s); s = fmaf(x[5], hi.y, s);
                bfrag[j] = (short)f32_to_bf16_bits(fast_tanh(s));
            }
            // layer 2, transposed: acc[t] (+)= W2[n = 32t + row][k] . H1^T[k][env]
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t nrow = (uint32_t)t * 32u + col;        // A operand: row = lane & 31
                const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(l_w2 + (size_t)nrow * W2_ROW_BYTES + ((uint32_t)kk * 16u + half * 8u) * 2u);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[t], 0, 0, 0);
            }
        }

        // layer 3 on the accumulators: this lane owns env `col` and hidden units n = 32t + (r&3) + 8(r>>2) + 4*half
        float o_acc[OUT];
#pragma unroll
        for (int o = 0; o < OUT; ++o) o_acc[o] = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t nn = (uint32_t)t * 32u + (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * half;
                const float h2 = fast_tanh(acc[t][r] + l_b2[nn]);
                const float4* wrow = reinterpret_cast<const float4*>(l_w3 + nn * MAX_OUT);
                float w[MAX_OUT];
                *reinterpret_cast<float4*>(w) = wrow[0];
                if (OUT > 4) *reinterpret_cast<float4*>(w + 4) = wrow[1];
                if (OUT > 8) *reinterpret_cast<float4*>(w + 8) = wrow[2];
#pragma unroll
                for (int o = 0; o < OUT; ++o) o_acc[o] = fmaf(h2, w[o], o_acc[o]);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of later rows from being hoisted
            }                                                          // (unbounded hoisting spilled the accumulators)
        }
#pragma unroll
        for (int o = 0; o < OUT; ++o) o_acc[o] += __shfl_xor(o_acc[o], 32, 64);      // the two halves own disjoint hidden units
        if (half == 0 && live) {
#pragma unroll
            for (int o = 0; o < OUT; ++o) out[(size_t)env * OUT + o] = o_acc[o] + l_b3[o];
        }
    }
}

}  // namespace q1pol
