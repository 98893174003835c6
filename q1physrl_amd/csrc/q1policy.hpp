// q1policy.hpp - fused forward pass of one policy / value network of the reference's shape (RLlib fcnet of the published
// checkpoint, SURVEY.md section 2: obs 6 -> 256 tanh -> 256 tanh -> OUT, OUT = 10 logits or 1 value) for gfx950.
//
// This is the GEMM-shaped neighbour of the env hot path (the sampler tick is: this, q1env_policy_sample, q1env_step,
// q1env_reset_philox), so it is the one place matrix cores are used - for all three layers, computed TRANSPOSED
// (hidden units are MFMA rows, envs are MFMA columns), so that the output of one layer is already distributed the way the
// next layer's B operand needs it and activations never leave registers:
//   layer 1  H1^T = W1ext . Xext^T      v_mfma_f32_32x32x2_f32 (exact float32; K = 8: six inputs, a 1 for the bias, a 0);
//                                       W1ext ([input][k], 8 KB) is read from LDS, 4 dwords per lane per 32-row tile
//   layer 2  H2^T = W2 . tanh(H1)^T     v_mfma_f32_32x32x16_bf16, float32 accumulate, accumulators start at the bias b2
//   layer 3  Y^T  = W3 . tanh(H2)^T     v_mfma_f32_32x32x16_bf16 (rows = outputs, padded to 32), + b3 in float32
// The C/D register layout of a 32x32 tile gives lane (c, h) the rows (r&3) + 8(r>>2) + 4h, r = 0..15, of column c, while a
// B operand wants 8 consecutive K indices per lane.  Instead of shuffling activations, the K index of the NEXT layer's
// weights is permuted once, when they are staged into LDS: within every 16 hidden units the four groups of four are stored
// in the order (0, 2, 1, 3).  Then lane (c, h)'s accumulator registers 8u..8u+7 ARE its B operand of K-step 2t + u.
// One workgroup (4 waves, one per SIMD) keeps W2 / W3 (bf16, rows padded by 16 B so the 16-byte operand reads are
// bank-conflict free), W1ext and b2 in LDS - 157.5 KB of the CU's 160 KB - and walks 128-env chunks grid-stride, so the weights
// are fetched once per CU, not once per chunk.  tanh = 1 - 2/(2^(2x log2 e) + 1) on v_exp_f32 / v_rcp_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace q1pol {

constexpr int HID = 256;
constexpr int OBS = 6;
constexpr int ROW_BYTES = HID * 2 + 16;               // 528: padded row stride of the bf16 weight rows in LDS
constexpr size_t LDS_W2 = (size_t)HID * ROW_BYTES;    // 135168
constexpr size_t LDS_W3 = (size_t)32 * ROW_BYTES;     // 16896 (output rows padded to one 32-row tile)
constexpr size_t LDS_B2 = (size_t)HID * 4;            // 1024
constexpr size_t LDS_W1 = (size_t)8 * HID * 4;        // 8192: W1ext as [input i = 0..7][k]
constexpr size_t LDS_TOTAL = LDS_W2 + LDS_W3 + LDS_B2 + LDS_W1;    // 161280 B <= 163840

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fast_tanh(float x) {                // 1 - 2 / (e^{2x} + 1): exact limits at +-inf
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f); // e^{2x} = 2^{2x log2 e}
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// tanh of accumulator registers 8u .. 8u+7 -> the lane's bf16 B operand of K-step 2t + u.  Two values at a time: the
// multiply, the +1 and the final 1 - 2r are packed float32 instructions (v_pk_mul/add/fma_f32), only v_exp_f32 and
// v_rcp_f32 are per element; the pair is converted with one v_cvt_pk_bf16_f32 (RNE).
__device__ __forceinline__ bf16x8 activate(const f32x16& acc, int u) {
    union { bf16x8 v; bf16x2 p[4]; } o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 x = {acc[8 * u + 2 * j], acc[8 * u + 2 * j + 1]};
        x = x * 2.8853900817779268f;                                  // 2 log2(e)
        f32x2 t = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        t = t + 1.0f;
        const f32x2 r = {__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
        const f32x2 y = 1.0f - 2.0f * r;
        o.p[j] = __builtin_convertvector(y, bf16x2);
    }
    return o.v;
}

// One 32-row tile of layer 1 on the float32 matrix path: rows k = 32t + col, A operand of step s = W1ext[k][2s + half].
__device__ __forceinline__ void layer1_operands(const float* l_w1, uint32_t t, uint32_t col, uint32_t half, float (&a)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = l_w1[(2u * s + half) * HID + t * 32u + col];
}
__device__ __forceinline__ f32x16 layer1_tile(const float (&a)[4], const float (&x1)[4]) {
    f32x16 d1;
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], x1[s], d1, 0, 0, 0);
    return d1;
}

// The host hands W2 and W3 over as ONE bf16 image that is already in LDS layout: (256 + 32) rows of 264 elements (528 B:
// 256 weights + 8 pad), columns of every row permuted (groups of four within each 16: 0,2,1,3).  Staging is then a straight
// 16-byte-per-lane copy of 152 064 bytes with all of a thread's loads in flight at once.
constexpr uint32_t IMG_VEC16 = (uint32_t)((LDS_W2 + LDS_W3) / 16);   // 9504 uint4
__device__ __forceinline__ void stage_image(unsigned char* dst, const uint16_t* __restrict__ img, uint32_t tid) {
    const uint4* src = reinterpret_cast<const uint4*>(img);
    uint4* d = reinterpret_cast<uint4*>(dst);
    constexpr uint32_t PER = (IMG_VEC16 + 255u) / 256u;              // 38 vectors per thread
    uint4 v[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t c = k * 256u + tid;
        v[k] = c < IMG_VEC16 ? src[c] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t c = k * 256u + tid;
        if (c < IMG_VEC16) d[c] = v[k];
    }
}

// w1: float[HID][OBS] (torch Linear(6,256).weight), b1: float[HID], w23: the bf16 LDS image of W2 (Linear(256,256).weight) and W3
// (Linear(256,out).weight in rows 0..out-1 of a 32-row tile) described above, b2: float[HID], b3: float[out_dim];
// obs float[n][6]; out float[n][out_dim].
template <int OUT>
__global__ void __launch_bounds__(256, 1)
mlp_forward_kernel(int n, const float* __restrict__ obs, const float* __restrict__ w1, const float* __restrict__ b1,
                   const uint16_t* __restrict__ w23, const float* __restrict__ b2, const float* __restrict__ b3,
                   float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* l_w2 = lds;
    unsigned char* l_w3 = lds + LDS_W2;
    float* l_b2 = reinterpret_cast<float*>(lds + LDS_W2 + LDS_W3);
    float* l_w1 = reinterpret_cast<float*>(lds + LDS_W2 + LDS_W3 + LDS_B2);   // [i][k]: inputs 0..5, bias (input 6 == 1), 0

    const uint32_t tid = threadIdx.x;
    stage_image(lds, w23, tid);                                      // W2 rows then W3 rows, contiguous in LDS
    l_b2[tid] = b2[tid];                                             // HID == blockDim.x
#pragma unroll
    for (int i = 0; i < OBS; ++i) l_w1[i * HID + tid] = w1[tid * OBS + i];
    l_w1[6 * HID + tid] = b1[tid];
    l_w1[7 * HID + tid] = 0.0f;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, half = lane >> 5;
    __syncthreads();

    const uint32_t nchunks = ((uint32_t)n + 127u) / 128u;
    // observations of the NEXT chunk are requested while the current one is computed (one wave per SIMD: a global load
    // issued at the top of a chunk would otherwise expose its whole HBM latency once per chunk)
    float xn[3];
    {
        const uint32_t e0 = blockIdx.x * 128u + wave * 32u + col;
#pragma unroll
        for (int s = 0; s < 3; ++s) xn[s] = (blockIdx.x < nchunks && e0 < (uint32_t)n) ? obs[(size_t)e0 * OBS + 2u * s + half] : 0.0f;
    }
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t env = chunk * 128u + wave * 32u + col;
        const bool live = env < (uint32_t)n;
        float x1[4];                                                 // B operands: Xext[env][2s + half]
#pragma unroll
        for (int s = 0; s < 3; ++s) x1[s] = xn[s];
        x1[3] = half ? 0.0f : 1.0f;
        {
            const uint32_t en = env + gridDim.x * 128u;
            const bool more = chunk + gridDim.x < nchunks && en < (uint32_t)n;
#pragma unroll
            for (int s = 0; s < 3; ++s) xn[s] = more ? obs[(size_t)en * OBS + 2u * s + half] : 0.0f;
        }

        f32x16 acc[8];                                               // H2^T pre-activations, start at b2
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = *reinterpret_cast<const float4*>(l_b2 + t * 32 + 8 * q + 4 * half);   // rows 8q + 4h + (0..3)
                acc[t][4 * q + 0] = b.x; acc[t][4 * q + 1] = b.y; acc[t][4 * q + 2] = b.z; acc[t][4 * q + 3] = b.w;
            }

        // Software pipeline over the eight 32-row tiles of H1: while the sixteen bf16 MFMAs of tile t1 run on the matrix
        // pipe, the VALU computes tanh of tile t1+1 (one wave per SIMD: nothing else could hide either behind the other).
        bf16x8 cur0, cur1;
        float a1n[4];                                                // layer-1 A operands, fetched one tile ahead as well
        {
            float a1[4];
            layer1_operands(l_w1, 0u, col, half, a1);
            layer1_operands(l_w1, 1u, col, half, a1n);
            const f32x16 d1 = layer1_tile(a1, x1);
            cur0 = activate(d1, 0); cur1 = activate(d1, 1);
        }
        // A operands (W2 rows) are fetched from LDS one K-step AHEAD of the MFMAs that consume them: with a single wave
        // per SIMD an LDS read issued right before its MFMA exposes its full latency 144 times per tile (measured: 48 %
        // of the wave's cycles in s_waitcnt).
        const unsigned char* wrow = l_w2 + (size_t)col * ROW_BYTES + half * 16u;      // + t2 * 32 rows, + K-step * 32 B
        bf16x8 a_even[8], a_odd[8];
#pragma unroll
        for (int t2 = 0; t2 < 8; ++t2) a_even[t2] = *reinterpret_cast<const bf16x8*>(wrow + (size_t)t2 * 32u * ROW_BYTES);
#pragma unroll 1
        for (int t1 = 0; t1 < 8; ++t1) {
            const uint32_t q0 = 2u * (uint32_t)t1;
            // phase A: request the odd K-step's operands      (sched_barrier: the compiler must not sink these reads
#pragma unroll                                          //   back down next to their MFMAs)
            for (int t2 = 0; t2 < 8; ++t2)
                a_odd[t2] = *reinterpret_cast<const bf16x8*>(wrow + (size_t)t2 * 32u * ROW_BYTES + (q0 + 1u) * 32u);
            float a1f[4];                                            // operands of tile t1 + 2, for the next iteration
            layer1_operands(l_w1, (uint32_t)(t1 + 2) & 7u, col, half, a1f);
            __builtin_amdgcn_sched_barrier(0);
            // phase B: even K-step on the matrix pipe; layer 1 + tanh of the NEXT tile on the VALU meanwhile
            const f32x16 dn = layer1_tile(a1n, x1);                  // tile t1 + 1 (the last pass recomputes tile 0, unused)
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2) acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_even[t2], cur0, acc[t2], 0, 0, 0);
            const bf16x8 nxt0 = activate(dn, 0);
            __builtin_amdgcn_sched_barrier(0);
            // phase C: request the next even K-step's operands
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2)
                a_even[t2] = *reinterpret_cast<const bf16x8*>(wrow + (size_t)t2 * 32u * ROW_BYTES + ((q0 + 2u) & 15u) * 32u);
            __builtin_amdgcn_sched_barrier(0);
            // phase D: odd K-step; second half of the next tile's tanh
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2) acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_odd[t2], cur1, acc[t2], 0, 0, 0);
            const bf16x8 nxt1 = activate(dn, 1);
            __builtin_amdgcn_sched_barrier(0);
            cur0 = nxt0; cur1 = nxt1;
#pragma unroll
            for (int q = 0; q < 4; ++q) a1n[q] = a1f[q];
        }

        // layer 3: all sixteen W3 operands are requested up front (64 VGPRs; the file has room with one wave per SIMD)
        bf16x8 w3f[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) w3f[q] = *reinterpret_cast<const bf16x8*>(l_w3 + (size_t)col * ROW_BYTES + ((uint32_t)q * 16u + half * 8u) * 2u);
        f32x16 y, y2;                                                // two chains: consecutive MFMAs do not wait on each other
#pragma unroll
        for (int r = 0; r < 16; ++r) { y[r] = 0.0f; y2[r] = 0.0f; }
#pragma unroll
        for (int t2 = 0; t2 < 8; ++t2) {
            const bf16x8 f0 = activate(acc[t2], 0), f1 = activate(acc[t2], 1);
            y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3f[2 * t2], f0, y, 0, 0, 0);
            y2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3f[2 * t2 + 1], f1, y2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] += y2[r];
        // y[r] = output row (r&3) + 8(r>>2) + 4*half of env `col`: rows 0..7 come from registers 0..3 of the two halves,
        // rows 8..9 from registers 4..5 of half 0
        if (live) {
            float* dst = out + (size_t)env * OUT;
            if (OUT >= 8) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r + 4 * half] = y[r] + b3[r + 4 * half];
                if (half == 0) {
#pragma unroll
                    for (int r = 4; r < 4 + (OUT - 8); ++r) dst[r + 4] = y[r] + b3[r + 4];
                }
            } else if (half == 0) {
#pragma unroll
                for (int r = 0; r < OUT; ++r) dst[r] = y[r] + b3[r];
            }
        }
    }
}

}  // namespace q1pol
