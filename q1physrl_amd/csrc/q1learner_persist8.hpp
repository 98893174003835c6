// q1learner_persist8.hpp - the persistent PPO learner with EIGHT waves per workgroup (two per SIMD).  Same decomposition, exchange protocol,
// buffers, numerics recipe and entry point as q1learner_persist.hpp (read that file first); what changes is who does what inside a workgroup.
//
// Why: the four-wave kernel is bound by the instruction streams of one wave per SIMD, and the issue limit of this machine is per WAVE - a
// wave issues one instruction every ~5.1 cycles whatever its type, and a second wave on the same SIMD issues at the same rate beside it
// (tools/ubench_halfexec.hip, profiles/r5_tick_floor.txt).  So every phase is cut in two halves that run on a wave pair:
//   wave W = 4 B + t:  t = the 32-sample tile, B = 0 "A wave" (orientation [unit][sample]: lane = sample), B = 1 "B wave" (lane = unit)
//   P1   A: H1^T tile -> published row-major;           B: H1 tile -> published transposed, kept for dZ1
//   P2   the 16 K-steps are SPLIT: wave (t, B) requests only K-steps 8 B .. 8 B + 7 of its tile's H1 rows (no operand is pulled twice
//        through the CU's L2 port) and forms the partial sums of BOTH orientations; the partials the partner needs cross through LDS
//        (32 KB exchange buffer); A: tanh, H2 tile, partial logits, publish;  B: tanh, H2^T
//   loss A waves (two lanes per sample, as before); B waves request their weight-gradient tile's optimizer state meanwhile
//   B3   A: dZ2 row-major -> published;                 B: dZ2^T -> LDS, db2 partial
//   G2   EIGHT 32-input tiles of dW2 rows U, one per wave (16 optimizer elements per lane instead of 32); dW3 / db2 / db3 + statistics ride on
//        waves 5 / 6 / 7
//   B2   K split again: partial dH1 of wave (t, 0) crosses through LDS to wave (t, 1), which holds H1 and forms dZ1^T; dW1 / db1 on wave 2
// 256 registers per lane (two waves per SIMD), 151 KB of LDS.
// RESULT (MI355X, round 5): equivalent (the persistent learner's tests pass with Q1_LEARNER_WAVES=8) and SLOWER than the four-wave kernel,
// 21.7 against 20.2 us per step: rows + layer 1 1.3 -> 1.0 and the weight gradients 5.3 -> 4.0 us gain, but every operand wait grows (eight
// waves behind the same 64 B / clock L2 port: 1.1 -> 1.7 us), each barrier's workgroup part costs more, and the K-split phases pay two
// workgroup barriers + an LDS round trip for 8 matrix products saved (P2 3.1 -> 3.5, B2 3.7 -> 4.0).  The per-wave issue limit measured by
// tools/ubench_halfexec.hip holds for independent register-only chains; these phases are latency chains (LDS, L2) that a second wave on the
// SIMD stretches.  Kept selectable (Q1_LEARNER_WAVES=8) as the record of the experiment; the product default is the four-wave kernel.
#pragma once
#include "q1learner_persist.hpp"

namespace q1pl {

constexpr uint32_t L_XCH = L_ST + 8192;                      // float [4 tiles][2 directions][16 registers][64 lanes]: 32 KB
constexpr uint32_t LDS_BYTES8 = L_XCH + 4 * 2 * 16 * 64 * 4; // 151 040

#define Q1PL_LD(i, off) "global_load_dwordx4 %" #i ", %[p], off offset:" #off Q1PL_SC "\n\t"
// 8 operand vectors (32 bytes apart) + this thread's two chunks (256 bytes apart) of W2's column block, one wait
__device__ __forceinline__ void ld8_2(const uint16_t* p, f16x8 (&o)[8], const uint16_t* q, f16x8 (&c)[2], bool loc) {
    if (loc) {
        asm volatile(Q1PL_LD(0, 0) Q1PL_LD(1, 32) Q1PL_LD(2, 64) Q1PL_LD(3, 96) Q1PL_LD(4, 128) Q1PL_LD(5, 160) Q1PL_LD(6, 192) Q1PL_LD(7, 224)
                     "global_load_dwordx4 %8, %[q], off" Q1PL_SC "\n\tglobal_load_dwordx4 %9, %[q], off offset:256" Q1PL_SC "\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(c[0]), "=&v"(c[1])
                     : [p] "v"(p), [q] "v"(q) : "memory");
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) o[s] = *reinterpret_cast<const f16x8*>(p + 16 * s);
        c[0] = *reinterpret_cast<const f16x8*>(q);
        c[1] = *reinterpret_cast<const f16x8*>(q + 128);
    }
}
#undef Q1PL_LD

template <int NI>
__device__ __forceinline__ void persistent_learner_body8(const Args& a, unsigned char* lds) {
    constexpr uint32_t NT = 512;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, W = tid >> 6, t = W & 3u, roleB = W >> 2, c = lane & 31u, h = lane >> 5;
    constexpr uint32_t ni = (uint32_t)NI;
    const uint32_t g = blockIdx.x >> 3;
    const Net net = a.net[NI];
    constexpr int OUT = NI == 0 ? 10 : 1;
    float* const fl = reinterpret_cast<float*>(lds + L_FL);
    float* const b2p = fl;                 // [32]
    float* const red2 = fl + 32;           // [4][32]
    int* const s_ok = reinterpret_cast<int*>(fl + 212);
    float* const statbuf = fl + 288;       // [3][128]
    float* const xch = reinterpret_cast<float*>(lds + L_XCH);
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float c2 = TANH_PRESCALE;
    const uint32_t U0 = 32u * g;
    const uint32_t bsm = 32u * t + c;                           // A waves: the sample this lane pair differentiates the loss of

    // ---------------------------------------------------------------- prologue
    for (uint32_t off = tid * 16u; off < L_FL; off += NT * 16u) *reinterpret_cast<uint4*>(lds + off) = uint4{0, 0, 0, 0};
    __syncthreads();
    float* const sW1 = reinterpret_cast<float*>(lds + L_ST);
    float* const sB1 = sW1 + 3 * 192;
    float* const sB2 = sB1 + 3 * 32;
    float* const sW3 = sB2 + 3 * 32;
    float* const sB3 = sW3 + 3 * 320;
    const size_t E_B2 = 65536, E_W1 = 65536 + 256, E_B1 = E_W1 + 1536, E_W3 = E_B1 + 256, E_B3 = E_W3 + (size_t)OUT * 256;
    auto small_state = [&](bool to_lds) {
        for (uint32_t e = tid; e < 192u; e += NT) {
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
        for (uint32_t e = tid; e < (uint32_t)OUT * 32u; e += NT) {
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
    // the W2 slice's optimizer state in owner-lane order: wave W owns the 32-input tile k = 32 W + c, slot r = unit U0 + row(r, h)
    float* const st_w = net.w2st + (((size_t)g * 3 + 0) * 8 + W) * 1024 + lane;
    float* const st_m = net.w2st + (((size_t)g * 3 + 1) * 8 + W) * 1024 + lane;
    float* const st_v = net.w2st + (((size_t)g * 3 + 2) * 8 + W) * 1024 + lane;
    const uint32_t kown = 32u * W + c;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t e = (size_t)(U0 + rrow(r, h)) * HID + kown;
        st_w[64 * r] = net.w2[e]; st_m[64 * r] = net.m[e]; st_v[64 * r] = net.v[e];
    }
    {
        const uint32_t u = tid >> 4, k0 = (tid & 15u) * 16u;                           // 16 threads per owned unit, 16 inputs each
        const float* src = net.w2 + (size_t)(U0 + u) * HID + k0;
        for (uint32_t k = 0; k < 16u; ++k) {
            const float wv = src[k];
            *reinterpret_cast<_Float16*>(lds + L_W2OWN + u * LD_W + 2u * (k0 + k)) = (_Float16)(c2 * wv);
            union { _Float16 hh; uint16_t b; } o; o.hh = (_Float16)wv;
            __hip_atomic_store(net.w2tx + (size_t)(k0 + k) * HID + U0 + u, o.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            *reinterpret_cast<_Float16*>(lds + L_W3 + (uint32_t)o * LD_32 + 2u * u) = wv;
            *reinterpret_cast<_Float16*>(lds + L_W3T + u * LD_16 + 2u * (uint32_t)o) = wv;
        }
    }
    if (tid < MB) *reinterpret_cast<_Float16*>(lds + L_XT + 6u * LD_B + 2u * tid) = (_Float16)1.0f;
    if (g == 0 && tid < 16u) __hip_atomic_store(net.b3x + tid, (int)tid < OUT ? net.b3[tid] : 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool loc = false;
    {
        if (tid == 0) __hip_atomic_store(net.bar + 16 + g, xcc_id() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bar_arrive(net.bar + 8, false);
        if (!bar_wait(net.bar + 8, (uint32_t)G, false, a.status, 3u, 0u, a.timeout_ticks, s_ok)) return;
        uint32_t same = 1u;
        const uint32_t mine = __hip_atomic_load(net.bar + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t k = 1; k < (uint32_t)G; ++k) same &= __hip_atomic_load(net.bar + 16 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine ? 1u : 0u;
        loc = a.allow_local != 0 && same != 0u && mine != 0u;
        if (tid == 0 && g == 0) a.status[2 + ni] = loc ? mine : 0u;
    }
    const float klc = *a.klc_dev;
    const long long step0 = *a.step_count;
    double pw1 = pow((double)a.beta1, (double)step0), pw2 = pow((double)a.beta2, (double)step0);
    float st_acc[3] = {0.0f, 0.0f, 0.0f};                       // (wave 7's lane 0 of workgroup 0 keeps the running statistics)
    float amax = 0.0f;
    uint32_t nsat = 0;
    uint32_t bar_n = 0;
    int64_t win = 0, in_epoch = 0;
    auto row_at = [&](int64_t window, uint32_t b) -> int64_t { return a.idx ? a.idx[window + (int64_t)b] : window + (int64_t)b; };
    int64_t srcX = row_at(0, tid & (MB - 1)), srcL = row_at(0, bsm);
    float oxn[6];
    auto request_obs = [&]() {
        const float2* o2 = reinterpret_cast<const float2*>(a.obs + (size_t)srcX * 6);
        const float2 p0 = o2[0], p1 = o2[1], p2 = o2[2];
        oxn[0] = p0.x; oxn[1] = p0.y; oxn[2] = p1.x; oxn[3] = p1.y; oxn[4] = p2.x; oxn[5] = p2.y;
    };
    request_obs();
    __syncthreads();

    const bool profiling = a.prof != nullptr && blockIdx.x == 0 && tid == 0;
    unsigned long long pacc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = profiling ? wall_clock64() : 0;
#define Q1PL_STAMP(k) do { if (profiling) { const uint64_t now_ = wall_clock64(); pacc[k] += now_ - tprev; tprev = now_; } } while (0)
    // this lane's slot of the partial-sum exchange: [tile][direction][register][lane]; direction 0 = written by the A wave (read by B), 1 = by B
    float* const xch_mine = xch + ((size_t)(t * 2u + roleB) * 16u) * 64u + lane;
    float* const xch_other = xch + ((size_t)(t * 2u + (roleB ^ 1u)) * 16u) * 64u + lane;

    for (int64_t step = 0; step < a.steps; ++step) {
        const uint32_t par = (uint32_t)(step & 1);
        const bool last = step + 1 == a.steps;
        uint16_t* const h1x = net.h1x + (size_t)par * MB * HID;
        uint16_t* const h1tx = net.h1tx + (size_t)par * MB * HID;
        pw1 *= (double)a.beta1;
        pw2 *= (double)a.beta2;
        if (tid < MB) {
            _Float16 hi[6], lo[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float xs = fminf(fmaxf(oxn[i], -65504.0f), 65504.0f);
                hi[i] = (_Float16)xs;
                lo[i] = (_Float16)(xs - (float)hi[i]);
            }
            _Float16* row = reinterpret_cast<_Float16*>(lds + L_XH + tid * LD_16);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                row[i] = hi[i]; row[8 + i] = lo[i];
                *reinterpret_cast<_Float16*>(lds + L_XT + (uint32_t)i * LD_B + 2u * tid) = hi[i];
                *reinterpret_cast<_Float16*>(lds + L_XT + (uint32_t)(8 + i) * LD_B + 2u * tid) = lo[i];
            }
            row[6] = (_Float16)1.0f; row[7] = (_Float16)0.0f; row[14] = (_Float16)1.0f; row[15] = (_Float16)0.0f;
        }
        __syncthreads();
        const float lr_bc1 = a.lr / (float)(1.0 - pw1), rs_bc2 = 1.0f / sqrtf((float)(1.0 - pw2));

        // ------------------------------------------------------------ P1: A waves publish H1 row-major, B waves transposed (and keep it)
        float h1B[16];                                          // B waves: tanh(H1)[b = 32 t + row(r)][u = c]
        {
            const f16x8 a1 = lds16(lds, L_W1 + c * LD_16 + 16u * h);
            const f16x8 x1 = lds16(lds, L_XH + (32u * t + c) * LD_16 + 16u * h);
            if (!roleB) {
                const f32x16 dA = mm(a1, x1, zero16);           // [u][b]: lane = sample, registers = units
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    pub8(h1x + (size_t)(32u * t + c) * HID + U0 + 8u * q + 4u * h, pack4(act(dA[4 * q]), act(dA[4 * q + 1]), act(dA[4 * q + 2]), act(dA[4 * q + 3])), loc);
#pragma unroll
                for (int r = 0; r < 16; ++r) h1B[r] = 0.0f;
            } else {
                const f32x16 dB = mm(x1, a1, zero16);           // [b][u]: lane = unit, registers = samples
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { t4[j] = act(dB[4 * q + j]); h1B[4 * q + j] = r16(t4[j]); }
                    pub8(h1tx + (size_t)(U0 + c) * MB + 32u * t + 8u * q + 4u * h, pack4(t4[0], t4[1], t4[2], t4[3]), loc);
                }
            }
        }
        Q1PL_STAMP(0);
        bar_arrive(net.bar, loc);
        const int64_t srcL_now = srcL;
        if (!last) {
            if (++in_epoch == a.spe) { in_epoch = 0; win += a.epoch_stride - (a.spe - 1) * MB; } else { win += MB; }
            srcX = row_at(win, tid & (MB - 1)); srcL = row_at(win, bsm);
        }
        if (!bar_wait(net.bar, (uint32_t)G * ++bar_n, loc, a.status, 0u, (uint32_t)step, a.timeout_ticks, s_ok)) return;
        Q1PL_STAMP(1);

        // ------------------------------------------------------------ P2: K-split partial sums of both orientations, exchange, tanh
        float h2X[16];                                          // A waves: tanh(H2)[u = row(r)][b = c];  B waves: tanh(H2)[b = row(r)][u = c]
        uint32_t in_kb = 0u;
        float in_a = 0.0f, in_b = 0.0f, in_c = 0.0f;
        float oldrow[10];
        {
            f16x8 bH[8], wc[2];
            const uint32_t jg = tid >> 4, ch = tid & 15u;       // column block: 32 rows x 32 chunks of 16 bytes, two chunks (ch, ch + 16) per thread
            ld8_2(h1x + (size_t)(32u * t + c) * HID + 128u * roleB + 8u * h, bH, net.w2tx + (size_t)(U0 + jg) * HID + 8u * ch, wc, loc);
            if (!roleB) {                                       // the loss's per-sample inputs (A waves), behind this phase's operands
                const size_t sl = (size_t)srcL_now;
                if (ni == 0) { in_kb = (uint32_t)a.keys[sl]; in_a = a.mouse_u[sl]; in_b = a.logp_old[sl]; in_c = a.adv[sl]; }
                else { in_a = a.value_old[sl]; in_b = a.vtarg[sl]; }
#pragma unroll
                for (int o = 0; o < 10; ++o) oldrow[o] = ni == 0 ? a.old_logits[sl * (size_t)a.old_stride + (size_t)o] : 0.0f;
            }
            f32x16 accA = zero16, accB = zero16;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const f16x8 aW = lds16(lds, L_W2OWN + c * LD_W + 32u * (8u * roleB + (uint32_t)s) + 16u * h);
                accA = mm(aW, bH[s], accA);
                accB = mm(bH[s], aW, accB);
            }
            *reinterpret_cast<f16x8*>(lds + L_W2COL + jg * LD_W + 16u * ch) = wc[0];
            *reinterpret_cast<f16x8*>(lds + L_W2COL + jg * LD_W + 16u * (ch + 16u)) = wc[1];
            // the partial the partner needs: A hands over its [b][u] partial, B its [u][b] partial
#pragma unroll
            for (int r = 0; r < 16; ++r) xch_mine[64 * r] = roleB ? accA[r] : accB[r];
            __syncthreads();
            if (!roleB) {
                const float bsel = 0.0f; (void)bsel;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bq = *reinterpret_cast<const float4*>(b2p + 8 * q + 4 * h);
                    const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
                    float tA[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { tA[j] = act((accA[4 * q + j] + xch_other[64 * (4 * q + j)]) + bb[j]); h2X[4 * q + j] = r16(tA[j]); }
                    *reinterpret_cast<uint64_t*>(lds + L_H2W + t * 32u * LD_32 + c * LD_32 + 2u * (8u * q + 4u * h)) = pack4(tA[0], tA[1], tA[2], tA[3]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f32x16 accY = zero16;                           // [o][b]: lane = sample, registers = outputs
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    accY = mm(lds16(lds, L_W3 + c * LD_32 + 32u * (uint32_t)s + 16u * h), lds16(lds, L_H2W + t * 32u * LD_32 + c * LD_32 + 32u * (uint32_t)s + 16u * h), accY);
                float* yrow = net.yp + ((size_t)g * MB + 32u * t + c) * 16u;
                pub8f(yrow + 4u * h, accY[0], accY[1], loc); pub8f(yrow + 4u * h + 2u, accY[2], accY[3], loc);
                pub8f(yrow + 8u + 4u * h, accY[4], accY[5], loc); pub8f(yrow + 10u + 4u * h, accY[6], accY[7], loc);
            } else {
                const float bB = b2p[c];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float tB[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { tB[j] = act((xch_other[64 * (4 * q + j)] + accB[4 * q + j]) + bB); h2X[4 * q + j] = r16(tB[j]); }
                    *reinterpret_cast<uint64_t*>(lds + L_H2T + c * LD_B + 2u * (32u * t + 8u * q + 4u * h)) = pack4(tB[0], tB[1], tB[2], tB[3]);
                }
            }
        }
        Q1PL_STAMP(2);
        bar_arrive(net.bar, loc);
        if (!bar_wait(net.bar, (uint32_t)G * ++bar_n, loc, a.status, 1u, (uint32_t)step, a.timeout_ticks, s_ok)) return;
        Q1PL_STAMP(3);

        // ------------------------------------------------------------ loss (A waves, two lanes per sample); B waves request their G2 state
        float wS[16], mS[16], vS[16];                           // this wave's weight-gradient tile: masters and moments
        if (!roleB) {
            float s3[3] = {0.0f, 0.0f, 0.0f};
            float gl[10];
            float y[12];
            {
                f32x4 part[5][3];
                const float* yb = net.yp + ((size_t)(4u * h) * MB + bsm) * 16u;
                ld_rows(yb, yb + (size_t)MB * 16, yb + (size_t)2 * MB * 16, yb + (size_t)3 * MB * 16, net.b3x, part, loc);
                float half_[12];
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) half_[4 * v + e] = ((part[0][v][e] + part[1][v][e]) + part[2][v][e]) + part[3][v][e];
#pragma unroll
                for (int o = 0; o < 12; ++o) {
                    const float other = __shfl_xor(half_[o], 32, 64);
                    const float lo_ = h ? other : half_[o], hi_ = h ? half_[o] : other;
                    y[o] = part[4][o >> 2][o & 3] + (lo_ + hi_);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { wS[r] = st_w[64 * r]; mS[r] = st_m[64 * r]; vS[r] = st_v[64 * r]; }
            if (!last && tid < MB) request_obs();
            Q1PL_STAMP(10);
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
            Q1PL_STAMP(11);
            _Float16 row16[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) row16[o] = (_Float16)0.0f;
#pragma unroll
            for (int o = 0; o < 10; ++o) {
                if (o < OUT) {
                    row16[o] = (_Float16)sat16(gl[o], amax, nsat);
                    if (!h) *reinterpret_cast<_Float16*>(lds + L_DYT + (uint32_t)o * LD_B + 2u * bsm) = row16[o];
                }
            }
            if (!h) {
                *reinterpret_cast<f16x8*>(lds + L_DY + bsm * LD_16) = *reinterpret_cast<const f16x8*>(row16);
                *reinterpret_cast<f16x8*>(lds + L_DY + bsm * LD_16 + 16u) = *reinterpret_cast<const f16x8*>(row16 + 8);
                statbuf[bsm] = s3[0]; statbuf[MB + bsm] = s3[1]; statbuf[2 * MB + bsm] = s3[2];
            }
            Q1PL_STAMP(12);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { wS[r] = st_w[64 * r]; mS[r] = st_m[64 * r]; vS[r] = st_v[64 * r]; }
        }
        __syncthreads();
        Q1PL_STAMP(4);

        // ------------------------------------------------------------ B3: dZ2 of the owned units: A row-major (published), B transposed (LDS)
        {
            const f16x8 aT = lds16(lds, L_W3T + c * LD_16 + 16u * h);
            const f16x8 bY = lds16(lds, L_DY + (32u * t + c) * LD_16 + 16u * h);
            if (!roleB) {
                const f32x16 dA = mm(aT, bY, zero16);           // [u][b]: lane = sample (h2X's layout on an A wave)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float zA[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) zA[j] = sat16(dA[4 * q + j] * (1.0f - h2X[4 * q + j] * h2X[4 * q + j]), amax, nsat);
                    pub8(net.dz2x + (size_t)(32u * t + c) * HID + U0 + 8u * q + 4u * h, pack4(zA[0], zA[1], zA[2], zA[3]), loc);
                }
            } else {
                const f32x16 dB = mm(bY, aT, zero16);           // [b][u]: lane = unit (h2X's layout on a B wave)
                float sb = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float zB[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { zB[j] = sat16(dB[4 * q + j] * (1.0f - h2X[4 * q + j] * h2X[4 * q + j]), amax, nsat); sb += r16(zB[j]); }
                    *reinterpret_cast<uint64_t*>(lds + L_DZ2T + c * LD_B + 2u * (32u * t + 8u * q + 4u * h)) = pack4(zB[0], zB[1], zB[2], zB[3]);
                }
                sb += __shfl_xor(sb, 32, 64);
                if (h == 0) red2[t * 32u + c] = sb;
            }
        }
        bar_arrive(net.bar, loc);
        Q1PL_STAMP(5);

        // ------------------------------------------------------------ G2: one 32-input tile of dW2 rows U per wave + Adam + new images
        {
            f16x8 hT[8];
            ld8(h1tx + (size_t)kown * MB + 8u * h, hT, loc);
            Q1PL_STAMP(14);
            f32x16 acc = zero16;                                // [u][k]: lane = input k, registers = owned units
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = mm(lds16(lds, L_DZ2T + c * LD_B + 32u * (uint32_t)s + 16u * h), hT[s], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int r = 4 * q + j;
                    const f32x2 gr = f32x2{acc[r], acc[r + 1]} * net.inv_scale;
                    f32x2 mm2 = {mS[r], mS[r + 1]}, vv2 = {vS[r], vS[r + 1]};
                    const f32x2 wn = adam2(f32x2{wS[r], wS[r + 1]}, gr, mm2, vv2, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                    wS[r] = wn.x; wS[r + 1] = wn.y; mS[r] = mm2.x; mS[r + 1] = mm2.y; vS[r] = vv2.x; vS[r + 1] = vv2.y;
                    const f32x2 wi = c2 * wn;
                    *reinterpret_cast<_Float16*>(lds + L_W2OWN + rrow(r, h) * LD_W + 2u * kown) = (_Float16)wi.x;
                    *reinterpret_cast<_Float16*>(lds + L_W2OWN + rrow(r + 1, h) * LD_W + 2u * kown) = (_Float16)wi.y;
                    if (last) { net.gw2[(size_t)(U0 + rrow(r, h)) * HID + kown] = gr.x; net.gw2[(size_t)(U0 + rrow(r + 1, h)) * HID + kown] = gr.y; }
                }
                pub8(net.w2tx + (size_t)kown * HID + U0 + 8u * q + 4u * h, pack4(wS[4 * q], wS[4 * q + 1], wS[4 * q + 2], wS[4 * q + 3]), loc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { st_w[64 * r] = wS[r]; st_m[64 * r] = mS[r]; st_v[64 * r] = vS[r]; }
            Q1PL_STAMP(15);
        }
        if (W == 5u) {                                          // dW3[:, U]: lane = owned unit, registers = outputs
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
                acc = mm(lds16(lds, L_DYT + c * LD_B + 32u * (uint32_t)s + 16u * h), lds16(lds, L_H2T + c * LD_B + 32u * (uint32_t)s + 16u * h), acc);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t o = rrow(r, h);
                if ((int)o < OUT) {
                    const float gr = acc[r] * net.inv_scale;
                    w3v[r] = adam1(w3v[r], gr, m3v[r], v3v[r], a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                    *reinterpret_cast<_Float16*>(lds + L_W3 + o * LD_32 + 2u * c) = (_Float16)w3v[r];
                    *reinterpret_cast<_Float16*>(lds + L_W3T + c * LD_16 + 2u * o) = (_Float16)w3v[r];
                    if (last) net.gw3[(size_t)o * HID + U0 + c] = gr;
                    const uint32_t i3 = o * 32u + c;
                    sW3[i3] = w3v[r]; sW3[320 + i3] = m3v[r]; sW3[640 + i3] = v3v[r];
                }
            }
        }
        if (W == 6u && h == 0u) {                               // db2[U]
            float b2v = sB2[c], mv = sB2[32 + c], vv = sB2[64 + c];
            const float gr = (((red2[c] + red2[32u + c]) + red2[64u + c]) + red2[96u + c]) * net.inv_scale;
            b2v = adam1(b2v, gr, mv, vv, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
            b2p[c] = c2 * b2v;
            sB2[c] = b2v; sB2[32 + c] = mv; sB2[64 + c] = vv;
            if (last) net.gb2[U0 + c] = gr;
        }
        f32x16 acc_b3 = zero16;                                 // [o][i']: lane (c = 6, h) holds db3[o = row(r, h)]
        if (g == 0 && W == 7u) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                acc_b3 = mm(lds16(lds, L_DYT + c * LD_B + 32u * (uint32_t)s + 16u * h), lds16(lds, L_XT + c * LD_B + 32u * (uint32_t)s + 16u * h), acc_b3);
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
        Q1PL_STAMP(6);
        if (!bar_wait(net.bar, (uint32_t)G * ++bar_n, loc, a.status, 2u, (uint32_t)step, a.timeout_ticks, s_ok)) return;
        Q1PL_STAMP(7);

        // ------------------------------------------------------------ B2: dH1 of the owned units, K split over the wave pair; dZ1; dW1 / db1 / db3
        {
            f16x8 zr[8];
            ld8(net.dz2x + (size_t)(32u * t + c) * HID + 128u * roleB + 8u * h, zr, loc);
            f32x16 acc = zero16;                                // [b][j]: lane = owned unit j, registers = samples (h1B's layout)
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = mm(zr[s], lds16(lds, L_W2COL + c * LD_W + 32u * (8u * roleB + (uint32_t)s) + 16u * h), acc);
            if (!roleB) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xch_mine[64 * r] = acc[r];
            }
            __syncthreads();
            if (roleB) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float z[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        z[j] = sat16((xch_other[64 * (4 * q + j)] + acc[4 * q + j]) * (1.0f - h1B[4 * q + j] * h1B[4 * q + j]), amax, nsat);
                    *reinterpret_cast<uint64_t*>(lds + L_DZ1T + c * LD_B + 2u * (32u * t + 8u * q + 4u * h)) = pack4(z[0], z[1], z[2], z[3]);
                }
            }
        }
        __syncthreads();
        if (W == 2u) {                                          // dW1[U] / db1[U]
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
                acc = mm(lds16(lds, L_XT + c * LD_B + 32u * (uint32_t)s + 16u * h), lds16(lds, L_DZ1T + c * LD_B + 32u * (uint32_t)s + 16u * h), acc);
            _Float16* row = reinterpret_cast<_Float16*>(lds + L_W1 + c * LD_16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (h == 0u || j < 2) {
                    const float gr = (acc[j] + acc[4 + j]) * net.inv_scale;
                    w1v[j] = adam1(w1v[j], gr, m1v[j], v1v[j], a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                    const _Float16 wv = (_Float16)(c2 * w1v[j]);
                    row[base + j] = wv; row[8 + base + j] = wv;
                    if (last) net.gw1[(size_t)(U0 + c) * 6 + (size_t)(base + j)] = gr;
                    const uint32_t i1 = c * 6u + (uint32_t)(base + j);
                    sW1[i1] = w1v[j]; sW1[192 + i1] = m1v[j]; sW1[384 + i1] = v1v[j];
                }
            }
            if (h) {
                const float gr = acc[2] * net.inv_scale;
                b1v = adam1(b1v, gr, mb1, vb1, a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                const float bs = c2 * b1v;
                const _Float16 bhi = (_Float16)bs;
                row[6] = bhi; row[14] = (_Float16)(bs - (float)bhi);
                if (last) net.gb1[U0 + c] = gr;
                sB1[c] = b1v; sB1[32 + c] = mb1; sB1[64 + c] = vb1;
            }
        }
        if (g == 0 && W == 7u && c == 6u) {                     // db3 / b3 (after barrier 3: every workgroup has read this step's b3)
            float bv[8], mv[8], vv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t o = rrow(r, h), oc = (int)o < OUT ? o : 0u;
                bv[r] = sB3[oc]; mv[r] = sB3[16 + oc]; vv[r] = sB3[32 + oc];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t o = rrow(r, h);
                if ((int)o < OUT) {
                    const float gr = acc_b3[r] * net.inv_scale;
                    bv[r] = adam1(bv[r], gr, mv[r], vv[r], a.beta1, a.beta2, a.eps, lr_bc1, rs_bc2);
                    sB3[o] = bv[r]; sB3[16 + o] = mv[r]; sB3[32 + o] = vv[r];
                    if (last) net.gb3[o] = gr;
                    pub4f(net.b3x + o, bv[r], loc);
                }
            }
        }
        __syncthreads();
        Q1PL_STAMP(8);
    }
#undef Q1PL_STAMP
    if (profiling)
        for (int k = 0; k < 20; ++k) a.prof[k] = pacc[k];

    // ---------------------------------------------------------------- epilogue
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t e = (size_t)(U0 + rrow(r, h)) * HID + kown;
        net.w2[e] = st_w[64 * r]; net.m[e] = st_m[64 * r]; net.v[e] = st_v[64 * r];
    }
    small_state(false);
    if (g == 0 && tid == 448u) {                                // (wave 7's lane 0 kept the running statistics)
        if (ni == 0) {
            a.stats_acc[0] += st_acc[0]; a.stats_acc[1] += st_acc[1]; a.stats_acc[2] += st_acc[2];
            *a.step_count = step0 + a.steps;
        } else {
            a.stats_acc[4] += st_acc[0];
        }
    }
    if (a.saturation) {
        if (nsat) atomicAdd(a.saturation + 2u * ni, nsat);
        if (amax > 0.0f) atomicMax(a.saturation + 2u * ni + 1u, __float_as_uint(amax));
    }
}

__global__ void __launch_bounds__(512, 1)
persistent_learner_kernel8(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t role = blockIdx.x & 7u;
    if (role == 0u) persistent_learner_body8<0>(a, lds);
    else if (role == 1u) persistent_learner_body8<1>(a, lds);
}

}  // namespace q1pl
