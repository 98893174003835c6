// q1learner_fused.hpp - the large-minibatch SGD step's forward, PPO loss gradient and data gradients as ONE kernel (round 6; VERDICT r3 - r5
// "32 768-sample learner step"): what learner_forward_kernel and learner_backward_kernel<true> of q1learner.hpp do in two launches with the
// 68 MB of float16 activations written by the first and read back by the second.  Counterpart of the torch modules + autograd of
// q1physrl_amd/ppo.py for the reference's PPO configuration (q1physrl/train.py:60-64, data/params.yml:4-13); included by q1env_learner.hip.
//
// One workgroup = eight waves = eight 32-sample tiles of one network (two waves per SIMD, 256 registers each), exactly one tile per wave:
//   phase F   the forward image (W2 x 2 log2 e, W3, layer-1 operands, b2: 157.5 KB) is staged into LDS; every wave runs its tile through
//             q1pol::mlp_tile_t - the sampler's forward, bit for bit - storing tanh(H1) (and, without the products below, tanh(H2)) in T-format for
//             the weight-gradient kernel and KEEPING tanh(H2) and the outputs in registers;
//   restage   barrier; the per-sample loss inputs are requested; the transposed images (W2^T 132 KB, W3^T one K-step: 12 KB) replace the
//             forward image in LDS; barrier;
//   phase B   the wave's OWN h1 vectors are requested back first (a wave's memory operations complete in order: behind the dZ2 stores they
//             would wait 5 - 15 us for those stores' acknowledgements on the XCDs whose write path is backed up), then the tile's PPO loss
//             gradient from the outputs in registers (q1ppo_loss.hpp, the same function and bits as learner_backward_kernel<true>),
//             dZ2 = W3^T dY (1 - h2^2) with h2 from registers, dZ1 = W2^T dZ2 (1 - h1^2); [x | 1], dY, dZ2 (and dZ1) transposed to N-format on
//             the matrix pipe and stored for the weight-gradient kernel, as before.
// DW1 = false: every array the weight-gradient kernel reads holds the same bits as after the two-launch sequence, so the whole step is
//   bit-identical to round 4's four-launch step (tests/test_hip_learner.py holds it to that; two whole training runs end on round 5's numbers).
// DW1 = true ("products"): neither dZ1 nor tanh(H2) is stored at all - 2 x 33.5 MB per network written and read back for a 256 x 7 and an
//   out x 256 product.  Each wave multiplies its tile's dZ1^T by [x | 1] and its dY^T by tanh(H2) right where the operands sit in registers -
//   one more pair of matrix instructions per 32-unit tile each - and stores the partial products (float32: 8 KB of dW1 / db1, 10 KB (policy) or
//   2 KB (value network) of dW3 per tile); learner_wgrad_shared_kernel adds its split's tiles up in tile order (deterministic; float32
//   additions per tile instead of one accumulation chain, so dW1, db1 and dW3 differ from the four-launch step's in the last bits - everything
//   else is still identical).  The operand slots of both products are permuted so that every lane stores the same 16-byte piece: a branch
//   around a store costs the register allocator 80 - 160 registers in these fully unrolled phases (tests/test_fused_learner_isa.py).
// A 32 768-sample step (profiles/r6_learner_fused.txt): 98.8 -> 81 - 83 us, 442 -> 318 MB.  The saturation report is per WORKGROUP: per wave it
// was 2 x 2 048 same-address atomics per launch, which kept the dispatch open 14 us after its last wave had ended.
#pragma once
#include "q1learner.hpp"

namespace q1learn {

constexpr int W3C_ROW_BYTES = 32 + 16;                        // compact W3^T row: outputs 0..15 (one K-step) + 16 B pad
constexpr size_t LDS_W3C = (size_t)HID * W3C_ROW_BYTES;       // 12288
constexpr size_t LDS_FZ_RED = q1pol::LDS_TOTAL;               // float[8][5] statistics rows of the eight waves, behind either phase's images
constexpr size_t LDS_FZ = LDS_FZ_RED + 256;                   // 161536 <= 163840
static_assert(LDS_W2T + LDS_W3C <= LDS_FZ_RED, "the backward images must fit under the statistics rows");
constexpr uint32_t DW1_TILE_FLOATS = 8u * 32u * 8u;           // per-tile partial products of dW1 / db1: float[unit tile 8][lane 32][reg 8]

struct FzNet {
    const float* w1; const float* b1; const uint16_t* w23; const float* b2; const float* b3;     // forward: masters of layer 1 / biases, forward image
    const uint16_t* w2t; const uint16_t* w3t;                                                  // backward images (q1learner.hpp BwdNet)
    f16x8* h1T; f16x8* h2T; f16x8* dz2N; f16x8* dz1N; f16x8* xN; f16x8* dyN;
    float* dw1p;              // DW1: float[tile][DW1_TILE_FLOATS]
    float4* dw3a; float* dw3b;  // DW1: per-tile products dY^T h2 (q1learner.hpp WgNet): float4[tile][8][64] (rows 0..7; policy network only), float[tile][8][64]
    uint32_t* sat;
};

#ifndef Q1_FZ_EXP         // timing experiments of the diagnostic build (tools/exp_fused_stamps.py; results are WRONG with any bit set): 1 = the forward phase
#define Q1_FZ_EXP 0       // stores nothing, 4 = hardware exp / log in the loss, 8 = neighbouring XCDs swap their tiles, 16 = the backward phase's stores are
#endif                    // folded into a checksum instead (the arithmetic that produces them stays), 32 = ... are non-temporal, 64 / 128 / 192 = the forward phase's tanh(H1) stores are sc1 / nt / sc0 sc1,
                          // 256 = no saturation report, 512 = every wave leaves its end time in one word (atomic max) for the next launch to read
template <class T>
__device__ __forceinline__ void fz_store(T* dst, const T& v, uint32_t& chk) {
    if constexpr ((Q1_FZ_EXP & 16) != 0) {
        union { T t; uint32_t u[sizeof(T) / 4]; } o;
        o.t = v;
#pragma unroll
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) chk ^= o.u[k];
    } else if constexpr ((Q1_FZ_EXP & 32) != 0) {
        typedef uint32_t uvec __attribute__((ext_vector_type(sizeof(T) / 4)));
        union { T t; uvec u; } o;
        o.t = v;
        __builtin_nontemporal_store(o.u, reinterpret_cast<uvec*>(dst));
    } else {
        *dst = v;
    }
}

template <bool DW1>
__global__ void __launch_bounds__(512, 1)
learner_fwdbwd_kernel(int n, const float* __restrict__ obs, const int64_t* __restrict__ idx, const int64_t* __restrict__ idx_cursor, FzNet net_a, FzNet net_b,
                      LossArgs la, BcArgs bca) {
    using namespace q1pol;
    if (idx && idx_cursor) idx += *idx_cursor;
    const uint32_t bgrid = gridDim.x / 2u;
    const bool second = blockIdx.x >= bgrid;
    const uint32_t bid = second ? blockIdx.x - bgrid : blockIdx.x;
    const FzNet net = second ? net_b : net_a;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t col = lane & 31u, half = lane >> 5;
    const uint32_t ntiles = ((uint32_t)n + 31u) / 32u;
#if defined(Q1_FZ_EXP) && (Q1_FZ_EXP & 8)               // timing experiment: the workgroup on XCD x takes the tiles its neighbour on XCD x ^ 1 would have
    const uint32_t tile = (bid ^ 1u) * 8u + wave;
#else
    const uint32_t tile = bid * 8u + wave;
#endif
    const bool tile_live = tile < ntiles;                     // wave-uniform; a wave without a tile still stages and meets the barriers
    const uint32_t s = tile * 32u + col;
    const bool live = tile_live && s < (uint32_t)n;
    const size_t tbase = (size_t)tile * TILE_VECS + lane;
#ifdef Q1_FZ_STAMPS       // diagnostic build (tools/exp_fused_stamps.py): 100 MHz stamps of every wave, left in the array the mode does not use (dW1 products / dZ1)
    const uint64_t stamp0 = wall_clock64();
    float stamps[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // the latest wave end of the PREVIOUS launch of this kernel (the diagnostic build launches it twice in a row): how long do the end of one dispatch and
    // the start of the next take?
    unsigned long long* g_last_end = reinterpret_cast<unsigned long long*>(la.stats_rows + 9000);
    const float gap_us = (Q1_FZ_EXP & 512) ? (float)(long long)(stamp0 - *g_last_end) * 0.01f : 0.0f;
#define Q1_FZ_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); stamps[k] = (float)(wall_clock64() - stamp0) * 0.01f; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define Q1_FZ_STAMP(k) do { } while (0)
#endif
    uint32_t chk = 0;

    // ---------------------------------------------------------------------------------------------------------------- phase F
    // the sample's observation row (both lanes of a sample's pair read all of it; six named scalars, not an array: indexed by `half` an array
    // ends up in scratch memory)
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f, o4 = 0.0f, o5 = 0.0f;
    const size_t src = tile_src(idx, s, live);
    {
        StageRegs<512, IMG_VEC16> r;
        stage_issue<512, IMG_VEC16>(r, net.w23, tid);         // the image's loads go out first; the dependent gather lands under them
        if (live) {
            const float* orow = obs + src * OBS;
            if (la.wide) {
                const float2 a = reinterpret_cast<const float2*>(orow)[0], b = reinterpret_cast<const float2*>(orow)[1], c = reinterpret_cast<const float2*>(orow)[2];
                o0 = a.x; o1 = a.y; o2 = b.x; o3 = b.y; o4 = c.x; o5 = c.y;
            } else {
                o0 = orow[0]; o1 = orow[1]; o2 = orow[2]; o3 = orow[3]; o4 = orow[4]; o5 = orow[5];
            }
        }
        stage_commit<512, IMG_VEC16>(lds, r, tid);
    }
    {
        const LdsNet l = lds_net(lds);
        if (tid < (uint32_t)HID) {
            l.b2[tid] = TANH_PRESCALE * net.b2[tid];
            stage_w1_row(l.w1, tid, net.w1, net.b1);
        }
    }
    __syncthreads();
    Q1_FZ_STAMP(0);
    f16x8 h2k[8][2];
    f32x16 y = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 8; ++t) { h2k[t][0] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0}; h2k[t][1] = h2k[t][0]; }
    if (tile_live) {
        const LdsNet l = lds_net(lds);
        const float x[3] = {half ? o1 : o0, half ? o3 : o2, half ? o5 : o4};                      // learner_forward_kernel: obs[src * 6 + 2 k + half]
        const f16x8 xb = split_inputs(x, half);
        const unsigned char* w1row = l.w1 + (size_t)col * 32u + half * 16u;
        const unsigned char* wrow = l.w2 + (size_t)col * ROW_BYTES + half * 16u;
        const unsigned char* w3row = l.w3 + (size_t)col * ROW_BYTES + half * 16u;
        y = mlp_tile_t<!(Q1_FZ_EXP & 1), true, !(Q1_FZ_EXP & 1) && !DW1, (Q1_FZ_EXP >> 6) & 3>(xb, w1row, wrow, w3row, l.b2, half, nullptr, net.h1T + tbase, net.h2T + tbase, h2k);
    }
    Q1_FZ_STAMP(1);

    // ---------------------------------------------------------------------------------------------------------------- restage
    // per-sample loss inputs (the gathers that hang on src), requested before the images: half 0 fetches the key bits and the mouse action,
    // half 1 the behaviour policy's row, old log-probability and advantage; they swap below (as learner_backward_kernel<true> does)
    float in_r[10], in_sc[2];
#pragma unroll
    for (int c = 0; c < 10; ++c) in_r[c] = 0.0f;
    in_sc[0] = in_sc[1] = 0.0f;
    float b3v[6];                                             // b3 of the output rows this lane's registers 0..5 hold
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (int)half;
        b3v[r] = row < (second ? 1 : 10) ? net.b3[row] : 0.0f;
    }
    if (live) {
        if (!second) {
            if (half) {
                const float* row = la.old_logits + src * (size_t)la.old_stride;
                if (la.wide) {
#pragma unroll
                    for (int c = 0; c < 10; c += 2) {
                        const float2 a = *reinterpret_cast<const float2*>(row + c);
                        in_r[c] = a.x; in_r[c + 1] = a.y;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 10; ++c) in_r[c] = row[c];
                }
                in_sc[0] = la.logp_old[src]; in_sc[1] = la.adv[src];
            } else {
                in_sc[0] = __uint_as_float((uint32_t)la.keys[src]); in_sc[1] = la.mouse[src];
            }
        } else {
            in_sc[0] = la.value_old[src]; in_sc[1] = la.vtarg[src];
        }
    }
    __syncthreads();                                          // every wave is done with the forward image
    Q1_FZ_STAMP(2);
    unsigned char* l_w2t = lds;
    unsigned char* l_w3c = lds + LDS_W2T;
    float (*red)[5] = reinterpret_cast<float (*)[5]>(lds + LDS_FZ_RED);
    {
        StageRegs<512, (uint32_t)(LDS_W2T / 16)> r;
        stage_issue<512, (uint32_t)(LDS_W2T / 16)>(r, net.w2t, tid);
        // W3^T: bytes 0..31 of every 80-byte image row (outputs 0..15: the in-kernel loss is written for 10 / 1 outputs), two 16-byte pieces per row
        const uint4 w3v = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(net.w3t) + (size_t)(tid >> 1) * W3T_ROW_BYTES)[tid & 1u];
        if (bca.step && blockIdx.x == 0 && tid == 0) {        // the optimizer's bias corrections of step count + 1 (q1learner.hpp BcArgs)
            const long long t = *bca.step + 1;
            bca.bc[0] = (float)(1.0 - pow((double)bca.beta1, (double)t));
            bca.bc[1] = (float)(1.0 - pow((double)bca.beta2, (double)t));
        }
        stage_commit<512, (uint32_t)(LDS_W2T / 16)>(l_w2t, r, tid);
        *reinterpret_cast<uint4*>(l_w3c + (size_t)(tid >> 1) * W3C_ROW_BYTES + (tid & 1u) * 16u) = w3v;
    }
    __syncthreads();
    Q1_FZ_STAMP(3);

    // ---------------------------------------------------------------------------------------------------------------- phase B
    float amax = 0.0f;
    float st[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (tile_live) {
        f16x8 e0, e1;                                         // selection operands of the transposition (learner_backward_kernel)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            e0[e] = (8u * half + (uint32_t)e == col) ? (_Float16)1.0f : (_Float16)0.0f;
            e1[e] = (8u * half + (uint32_t)e + 16u == col) ? (_Float16)1.0f : (_Float16)0.0f;
        }
        const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned char* w2trow = l_w2t + (size_t)col * ROW_BYTES + half * 16u;
        const unsigned char* w3crow = l_w3c + (size_t)col * W3C_ROW_BYTES + half * 16u;
        const float klc = *la.kl_coeff_dev;
        // ---- the wave's own h1 vectors back, requested FIRST: memory operations of a wave complete in order, so requested behind the dZ2 stores
        //      (where they are needed) they would wait for those stores' acknowledgements - 5 .. 15 us on the XCDs whose write path is backed up
        //      (tools/exp_fused_stamps.py)
        f16x8 hv[8][2];
#pragma unroll
        for (int t = 0; t < 8; ++t) { hv[t][0] = net.h1T[tbase + (2u * t) * 64u]; hv[t][1] = net.h1T[tbase + (2u * t + 1u) * 64u]; }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the outputs as the forward kernel would have stored them (y + b3), and the halves' gathers, swapped
        float v6[6], p6[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) v6[r] = y[r] + b3v[r];
#pragma unroll
        for (int r = 0; r < 6; ++r) p6[r] = __shfl_xor(v6[r], 32, 64);
        float y0[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y0[e] = 0.0f;
        if (!second) {
            TileVals tv;
            float pr[10], ps[2];
#pragma unroll
            for (int c = 0; c < 10; ++c) pr[c] = __shfl_xor(in_r[c], 32, 64);
            ps[0] = __shfl_xor(in_sc[0], 32, 64); ps[1] = __shfl_xor(in_sc[1], 32, 64);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                tv.lg[c] = half ? p6[c] : v6[c];                                       // rows 0..3: half 0's registers 0..3
                tv.lg[4 + c] = half ? v6[c] : p6[c];                                   // rows 4..7: half 1's registers 0..3
            }
            tv.lg[8] = half ? p6[4] : v6[4]; tv.lg[9] = half ? p6[5] : v6[5];          // rows 8, 9: half 0's registers 4, 5
#pragma unroll
            for (int c = 0; c < 10; ++c) tv.ol[c] = half ? in_r[c] : pr[c];
            tv.kb = __float_as_uint(half ? ps[0] : in_sc[0]);
            tv.mouse = half ? ps[1] : in_sc[1];
            tv.logp_old = half ? in_sc[0] : ps[0];
            tv.adv = half ? in_sc[1] : ps[1];
            float g[10];
#pragma unroll
            for (int c = 0; c < 10; ++c) g[c] = 0.0f;
            const PpoSample in{tv.kb, tv.mouse, tv.logp_old, tv.adv};
            const PpoSums ps2 = ppo_policy_grad<true, true, (Q1_FZ_EXP & 4) != 0>(la.p, tv.lg, tv.ol, in, la.clip, la.ent_coeff, klc, la.inv_b, g, 10, half);
            if (live && half == 0u) { st[0] += ps2.ent; st[1] += ps2.kl; st[2] += -ps2.surr; st[3] += -ps2.surr + klc * ps2.kl - la.ent_coeff * ps2.ent; }
#pragma unroll
            for (int e = 0; e < 8; ++e) y0[e] = live ? (half ? (e < 2 ? g[8 + e] : 0.0f) : g[e]) : 0.0f;
        } else if (live) {
            float vf;
            const float v = half ? p6[0] : v6[0];
            const float dvf = ppo_value_grad(v, in_sc[0], in_sc[1], la.vf_clip, vf);
            if (half == 0u) { y0[0] = la.vf_coeff * dvf * la.inv_bv; st[3] += la.vf_coeff * vf; st[4] += vf; }
        }
        f16x8 dyb0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            amax = fmaxf(fabsf(y0[e]), amax);
            dyb0[e] = (_Float16)fminf(fmaxf(y0[e], -65504.0f), 65504.0f);
        }
        Q1_FZ_STAMP(4);
        // DW1: dY once more as the A operand of the per-tile dW3 products, output 9 moved from slot 9 to slot 12 (see the dZ2 phase)
        f16x8 dq0 = zero8, dq1 = zero8;
        if constexpr (DW1) {
            f16x8 dyq = dyb0;
            if (half) { dyq[4] = dyb0[1]; dyq[1] = (_Float16)0.0f; }
            const f32x16 dqt = transpose_tile(dyq, zero8, e0, e1);
            dq0 = cvt8(dqt, 0); dq1 = cvt8(dqt, 1);
        }
        // ---- [x | 1] and dY in N-format for the weight-gradient kernel
        f16x8 xd0 = zero8, xd1 = zero8;                      // DW1: [x | 1] once more, inputs 4..6 moved to operand slots 8..10 (see the dZ1 epilogue)
        {
            f16x8 x0;
            const float one = live ? 1.0f : 0.0f;
            const float xv[4] = {half ? o4 : o0, half ? o5 : o1, half ? one : o2, half ? 0.0f : o3};          // element e = input 4 half + e of [x | 1]
#pragma unroll
            for (int e = 0; e < 8; ++e) x0[e] = e < 4 ? (_Float16)fminf(fmaxf(xv[e], -65504.0f), 65504.0f) : (_Float16)0.0f;
            const f32x16 dx = transpose_tile(x0, zero8, e0, e1);
            const size_t sb = (size_t)tile * 128u + lane;
            fz_store(net.xN + sb, cvt8(dx, 0), chk); fz_store(net.xN + sb + 64u, cvt8(dx, 1), chk);
            if constexpr (DW1) {
                // a second transposition with inputs 0..3 in slots 0..3 and inputs 4, 5, the constant 1 in slots 8..10 (all in the half-0 lanes'
                // vectors): as the A operand of the dW1 product its rows 0..3 (registers 0..3 of the half-0 lanes) are then inputs 0..3 and its
                // rows 4..7 (registers 0..3 of the half-1 lanes) inputs 4, 5, the bias, nothing - every lane stores ONE 16-byte piece
                f16x8 xq;
                const float qv[8] = {o0, o1, o2, o3, o4, o5, one, 0.0f};
#pragma unroll
                for (int e = 0; e < 8; ++e) xq[e] = half ? (_Float16)0.0f : (_Float16)fminf(fmaxf(qv[e], -65504.0f), 65504.0f);
                const f32x16 dq = transpose_tile(xq, zero8, e0, e1);
                xd0 = cvt8(dq, 0); xd1 = cvt8(dq, 1);
            }
            const f32x16 dd = transpose_tile(dyb0, zero8, e0, e1);
            fz_store(net.dyN + sb, cvt8(dd, 0), chk); fz_store(net.dyN + sb + 64u, cvt8(dd, 1), chk);
        }
        Q1_FZ_STAMP(5);
        // ---- dH2^T = W3^T dY^T, dZ2 = dH2 (1 - h2^2)
        f16x8 dzb[8][2];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(w3crow + (size_t)t * 32u * W3C_ROW_BYTES);
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, dyb0, zero16, 0, 0, 0);
            times_dtanh(acc, h2k[t][0], h2k[t][1]);
            dzb[t][0] = cvt8_sat(acc, 0, amax);
            dzb[t][1] = cvt8_sat(acc, 1, amax);
            const f32x16 dt = transpose_tile(dzb[t][0], dzb[t][1], e0, e1);
            fz_store(net.dz2N + tbase + (2u * (uint32_t)t) * 64u, cvt8(dt, 0), chk);                  // (store_n)
            fz_store(net.dz2N + tbase + (2u * (uint32_t)t + 1u) * 64u, cvt8(dt, 1), chk);
            if constexpr (DW1) {
                __builtin_amdgcn_sched_barrier(0);            // (the two halves of the iteration one after the other: their temporaries do not fit side by side)
                // dW3 of this tile's 32 samples and 32 units: rows = outputs (slots 0..8 and 12 of dyq), columns = units (lane c = unit sigma(c)) - registers
                // 0..3 of lane (c, h) are outputs 4 h .. 4 h + 3, register 4 is output 8 (h = 0) / 9 (h = 1); a one-output network has its only row in
                // register 0 of the half-0 lanes.  tanh(H2) itself is stored nowhere.
                const f32x16 ht = transpose_tile(h2k[t][0], h2k[t][1], e0, e1);
                f32x16 p3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(dq0, cvt8(ht, 0), zero16, 0, 0, 0);
                p3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(dq1, cvt8(ht, 1), p3, 0, 0, 0);
                const size_t pb = ((size_t)tile * 8u + (uint32_t)t) * 64u + lane;
                // (no branch on the network: a one-output network's dw3a is a 16-byte dummy every lane overwrites - a branch per iteration costs
                // the register allocator ~80 registers here)
                fz_store(net.dw3a + (second ? 0 : pb), make_float4(p3[0], p3[1], p3[2], p3[3]), chk);
                fz_store(net.dw3b + pb, second ? p3[0] : p3[4], chk);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        Q1_FZ_STAMP(6);
        // ---- dH1^T = W2^T dZ2^T in two passes of four row tiles, each followed by its dZ1 epilogue (learner_backward_kernel).  The wave's own
        //      h1 vectors come back four tiles at a time, requested at the head of the pass that ends with them (64 matrix instructions later):
        //      with all eight tiles held the kernel does not fit its 256 registers
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
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], dzb[q >> 1][q & 1], acc1[t], 0, 0, 0);
                        if (q < 15) a[t] = *reinterpret_cast<const f16x8*>(w2trow + (size_t)(4 * pass + t) * 32u * ROW_BYTES + (uint32_t)(q + 1) * 32u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            Q1_FZ_STAMP(7 + 2 * pass);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                times_dtanh(acc1[t], hv[4 * pass + t][0], hv[4 * pass + t][1]);
                const f16x8 z0 = cvt8_sat(acc1[t], 0, amax), z1 = cvt8_sat(acc1[t], 1, amax);
                const f32x16 d = transpose_tile(z0, z1, e0, e1);
                if constexpr (DW1) {
                    // rows = inputs, columns = the tile's units (lane c = unit sigma(c)): registers 0..3 of lane (c, h) are d loss / d (W1[unit][4 h ..
                    // 4 h + 3]) of this tile's 32 samples, with "W1[unit][6]" = b1[unit] and slot 7 empty
                    f32x16 p = __builtin_amdgcn_mfma_f32_32x32x16_f16(xd0, cvt8(d, 0), zero16, 0, 0, 0);
                    p = __builtin_amdgcn_mfma_f32_32x32x16_f16(xd1, cvt8(d, 1), p, 0, 0, 0);
                    fz_store(reinterpret_cast<float4*>(net.dw1p + (size_t)tile * DW1_TILE_FLOATS + ((size_t)(4 * pass + t) * 32u + col) * 8u + 4u * half),
                             make_float4(p[0], p[1], p[2], p[3]), chk);
                } else {
                    fz_store(net.dz1N + tbase + (2u * (uint32_t)(4 * pass + t)) * 64u, cvt8(d, 0), chk);       // (store_n)
                    fz_store(net.dz1N + tbase + (2u * (uint32_t)(4 * pass + t) + 1u) * 64u, cvt8(d, 1), chk);
                }
                __builtin_amdgcn_sched_barrier(0);            // one tile's epilogue at a time (interleaved, the four of them do not fit the registers)
            }
            Q1_FZ_STAMP(8 + 2 * pass);
        }
    }
#ifdef Q1_FZ_STAMPS
    if (tile_live && lane == 0) {                            // (DW1: the dZ1 array is the unused one)
        stamps[11] = (float)(wall_clock64() - stamp0) * 0.01f;
        stamps[12] = gap_us;
        stamps[13] = (float)(stamp0 & 0xFFFFFFull);           // (absolute start, 10-ns ticks modulo 2^24: when did this wave begin, relative to the others?)
        float* dst = DW1 ? reinterpret_cast<float*>(net.dz1N + (size_t)tile * TILE_VECS) : net.dw1p + (size_t)tile * DW1_TILE_FLOATS;
        for (int k = 0; k < 14; ++k) dst[k] = stamps[k];
    }
    if ((Q1_FZ_EXP & 512) && lane == 0) atomicMax(g_last_end, (unsigned long long)wall_clock64());
    if (Q1_FZ_EXP & 16) la.stats_rows[2560u * 2u + (size_t)blockIdx.x * 512u / 64u + wave] = __uint_as_float(chk ^ __shfl_xor(chk, 17, 64));
#endif
    // ---- statistics rows and the saturation report: ONE row / one pair of atomics per workgroup.  (The first version reported per wave, as
    //      learner_backward_kernel did: 2 x 2 048 atomics on two words per launch, which the memory side executes one after the other - the dispatch
    //      lasted 14 us longer than its last wave, profiles/r6_learner_fused.txt.)
    uint32_t (*red_sat)[2] = reinterpret_cast<uint32_t (*)[2]>(lds + LDS_FZ_RED + 160);
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) st[k] += __shfl_down(st[k], off, 64);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red[wave][k] = st[k];
    }
    if (net.sat) {
        const uint64_t over = __ballot(amax > 65504.0f);
        float wmax = amax;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
        if (lane == 0) { red_sat[wave][0] = (uint32_t)__popcll(over); red_sat[wave][1] = __float_as_uint(wmax); }
    }
    __syncthreads();
    if (tid < 5u) {
        float a = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) a += red[w][tid];
        la.stats_rows[(size_t)blockIdx.x * 5u + tid] = a;
    }
    if (!(Q1_FZ_EXP & 256) && net.sat && tid == 64u) {
        uint32_t cnt = 0, mx = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { cnt += red_sat[w][0]; mx = max(mx, red_sat[w][1]); }       // (bits of non-negative floats order like the floats)
        sat_report(net.sat, cnt, mx);
    }
}

}  // namespace q1learn
