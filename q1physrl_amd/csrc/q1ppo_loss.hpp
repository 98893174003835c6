// q1ppo_loss.hpp - the PPO loss of a minibatch and its gradient w.r.t. the policy outputs as one kernel (device code shared by
// q1env_policy.hip: q1env_ppo_loss_grad, and q1env_learner.hip: the native learner step).  A template, so that both translation units
// may instantiate it.
#pragma once
#include "q1policy_glue.hpp"

// PPO loss of one minibatch and its gradient with respect to the policy outputs (learner-side glue): the
// closed forms of q1physrl_amd/ppo.py::ppo_loss (RLlib 0.8.4 PPOLoss over the reference's Q1PhysActionDist, action_dist.py:46-243)
// differentiated by hand, one lane per sample - replaces ~100 elementwise launches + their autograd twins per SGD step:
//   keys k:  d = l1 - l0, p = sigmoid(d), a = action bit:  logp -= softplus(a ? -d : d)            d logp / dd = a - p
//            H += softplus(d) - d p                                                                   dH / dd = -d p (1 - p)
//            KL(old || new) += p_o (logp_o1 - logp_n1) + (1 - p_o)(logp_o0 - logp_n0)                dKL / dd = p - p_o
//   mouse:   u = S ndtri((x - low) / (high - low)), z = (u - mean) / std  (mean, log_std clamped; the clamp gates the gradient)
//            logp += N(mean, std).logpdf(u) - N(0, S).logpdf(u) - log(high - low)                    d/dmean = z / std, d/dlog_std = z^2 - 1
//            H  += log(high - low) - (log S - log_std + (std^2 + mean^2) / (2 S^2) - 1/2)
//            KL += log_std - log_std_o + (std_o^2 + (mean_o - mean)^2) / (2 std^2) - 1/2
//   ratio = exp(logp - logp_old), surrogate = min(adv ratio, adv clip(ratio, 1 -+ c))              d/dlogp = adv ratio if adv ratio <= adv clip(..)
//   vf = max((v - vt)^2, (v_old + clip(v - v_old, +-vc) - vt)^2)
//   total = mean(-surrogate + kl_coeff KL + vf_coeff vf - ent_coeff H);  dlogits / dvalue are d total / d(logits, value).
// partials[block][5] = sums of (entropy, kl, -surrogate, total, vf) over the block's samples (no atomics; add them up and divide by B).
// GATHER (the native learner, q1learner.hpp): sample i of the minibatch is row idx[i] of the per-sample inputs (old_logits with
// old_stride, keys, mouse, logp_old, adv, value_old, vtarg - the whole trajectory batch, never copied); logits / value / dlogits /
// dvalue are minibatch-local rows i.  out_scale / out_scale_v multiply dlogits / dvalue (the learner asks for B x pi_upscale / (B / value_downscale) x the averaged
// gradient so that it sits in float16's normal range; 1 otherwise); the statistics are unaffected.
// Per-sample inputs of the policy part and what it returns besides the gradient row
struct PpoSample { uint32_t kb; float mouse, logp_old, adv; };
struct PpoSums { float ent, kl, surr; };

// The policy part of one sample: entropy, KL, surrogate, and g[c] = inv_b x d(-surrogate + klc KL - ent_coeff H) / d row[c] for the
// row's row_stride outputs.  FIXED = the reference's own action structure (4 keys + continuous mouse, 10 outputs): every index into
// g is then a compile-time constant, so g may live in registers (the native learner's backward kernel computes its own dY with this,
// q1learner.hpp); FIXED = false reads the structure from Params (the stand-alone kernel below).  Same operations in the same order
// either way: both users produce the same bits.
// PAIR (with FIXED; the native learner's backward kernel): lanes l and l ^ 32 evaluate the SAME sample, so the four keys' terms - two
// exponentials, two log1p and two divisions each: half of the function's instructions - are computed once per pair, keys 0, 1 by the
// lane with half = 0 and keys 2, 3 by its partner, and swapped (six values per key); both lanes then run the accumulations over k = 0..3
// in the order of the one-lane form, on the same values: the same bits.  All 64 lanes must be active in the call.
// FAST (the persistent learner, q1learner_persist.hpp - a kernel bound by its instruction count): the hardware's exponential and
// logarithm (v_exp_f32 / v_log_f32 behind __expf / __logf, ~1e-6 relative) in place of the library's expf / log1pf, and in.mouse holds
// the squashed-Gaussian pre-image u = S ndtri((x - low) / (high - low)) ALREADY (computed by the caller, off its critical path).  The
// results differ from the exact forms by ~1e-6 relative - three orders below the float16 rounding the gradient row gets next.
template <bool FIXED, bool PAIR = false, bool FAST = false>
__device__ __forceinline__ PpoSums ppo_policy_grad(const Params& p, const float* __restrict__ row, const float* __restrict__ old, const PpoSample& in,
                                                   float clip, float ent_coeff, float klc, float inv_b, float* __restrict__ g, int row_stride,
                                                   uint32_t half = 0u) {
    static_assert(!PAIR || FIXED, "the pair form is written for the fixed action structure");
    const int nk = FIXED ? 4 : p.num_keys;
    const int yaw_mode = FIXED ? 1 : p.yaw_mode;
    const uint32_t kb = in.kb;
    float logp = 0.0f, ent = 0.0f, kl = 0.0f;
    float dlp[4], dh[4], dk[4];
    auto expf = [](float z) { return FAST ? __expf(z) : ::expf(z); };                       // (shadows the library function below)
    auto softplus = [&](float z) { return (z > 0.0f ? z : 0.0f) + (FAST ? __logf(1.0f + __expf(-fabsf(z))) : log1pf(::expf(-fabsf(z)))); };
    if constexpr (PAIR) {
        float m[2][6], x[2][6];                                                     // [key of this half][logp term, entropy term, kl term, dlp, dh, dk]
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                               // this lane's keys: k = 2 half + j
            const float l0 = half ? row[4 + 2 * j] : row[2 * j], l1 = half ? row[5 + 2 * j] : row[2 * j + 1];
            const float o0 = half ? old[4 + 2 * j] : old[2 * j], o1 = half ? old[5 + 2 * j] : old[2 * j + 1];
            const float d = l1 - l0, d_o = o1 - o0;
            const float pn = 1.0f / (1.0f + expf(-d)), po = 1.0f / (1.0f + expf(-d_o));
            const float a = (float)((kb >> (2u * half + (uint32_t)j)) & 1u);
            const float sp_pos = softplus(d), sp_neg = softplus(-d);
            m[j][0] = a != 0.0f ? sp_neg : sp_pos;
            m[j][1] = sp_pos - d * pn;
            m[j][2] = po * (sp_neg - softplus(-d_o)) + (1.0f - po) * (sp_pos - softplus(d_o));
            m[j][3] = a - pn; m[j][4] = -d * pn * (1.0f - pn); m[j][5] = pn - po;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 6; ++q) x[j][q] = __shfl_xor(m[j][q], 32, 64);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = k & 1;
            const bool mine = (uint32_t)(k >> 1) == half;                           // keys 0, 1 belong to half 0, keys 2, 3 to half 1
            logp -= mine ? m[j][0] : x[j][0];
            ent += mine ? m[j][1] : x[j][1];
            kl += mine ? m[j][2] : x[j][2];
            dlp[k] = mine ? m[j][3] : x[j][3]; dh[k] = mine ? m[j][4] : x[j][4]; dk[k] = mine ? m[j][5] : x[j][5];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= nk) break;
            const float d = row[2 * k + 1] - row[2 * k], d_o = old[2 * k + 1] - old[2 * k];
            const float pn = 1.0f / (1.0f + expf(-d)), po = 1.0f / (1.0f + expf(-d_o));
            const float a = (float)((kb >> k) & 1u);
            const float sp_pos = softplus(d), sp_neg = softplus(-d);            // -log p(0), -log p(1)
            logp -= a != 0.0f ? sp_neg : sp_pos;
            ent += sp_pos - d * pn;
            kl += po * (sp_neg - softplus(-d_o)) + (1.0f - po) * (sp_pos - softplus(d_o));
            dlp[k] = a - pn; dh[k] = -d * pn * (1.0f - pn); dk[k] = pn - po;
        }
    }
    float dlp_m = 0.0f, dlp_s = 0.0f, dh_m = 0.0f, dh_s = 0.0f, dk_m = 0.0f, dk_s = 0.0f;
    bool in_m = false, in_s = false;
    // discrete mouse: Categorical over M = 2S+1 logits (see sample_categorical):
    //   logp += l_a - lse;  H_c = -sum p_j log p_j;  KL_c = sum po_j (log po_j - log p_j)
    //   d logp / d l_j = [j == a] - p_j;  d H_c / d l_j = -p_j (log p_j + H_c);  d KL_c / d l_j = p_j - po_j
    const int cat_m = yaw_mode == 2 ? 2 * (int)p.yaw_steps + 1 : 0;
    float cat_lse = 0.0f, cat_lse_o = 0.0f, cat_h = 0.0f;
    int cat_a = 0;
    if (!FIXED && yaw_mode == 2) {
        const float* l = row + 2 * nk;
        const float* lo = old + 2 * nk;
        float mx = l[0], mxo = lo[0];
        for (int j = 1; j < cat_m; ++j) { mx = fmaxf(mx, l[j]); mxo = fmaxf(mxo, lo[j]); }
        float sn = 0.0f, so = 0.0f;
        for (int j = 0; j < cat_m; ++j) { sn += expf(l[j] - mx); so += expf(lo[j] - mxo); }
        cat_lse = mx + logf(sn);
        cat_lse_o = mxo + logf(so);
        cat_a = min(max((int)in.mouse, 0), cat_m - 1);
        logp += l[cat_a] - cat_lse;
        float kc = 0.0f;
        for (int j = 0; j < cat_m; ++j) {
            const float lpn = l[j] - cat_lse, lpo = lo[j] - cat_lse_o;
            cat_h -= expf(lpn) * lpn;
            kc += expf(lpo) * (lpo - lpn);
        }
        ent += cat_h;
        kl += kc;
    }
    if (yaw_mode == 1) {
        const float S = SQUASH_SCALE, low = -p.action_range_f32, high = p.action_range_f32;
        const float m_raw = row[2 * nk], s_raw = row[2 * nk + 1];
        in_m = m_raw >= -3.0f && m_raw <= 3.0f;
        in_s = s_raw >= -20.0f && s_raw <= 2.0f;
        const float mean = fminf(fmaxf(m_raw, -3.0f), 3.0f), ls = fminf(fmaxf(s_raw, -20.0f), 2.0f);
        const float mean_o = fminf(fmaxf(old[2 * nk], -3.0f), 3.0f), ls_o = fminf(fmaxf(old[2 * nk + 1], -20.0f), 2.0f);
        const float inv_std = expf(-ls), std = expf(ls), std_o = expf(ls_o);
        const float u = FAST ? in.mouse : S * normcdfinvf((in.mouse - low) / (high - low));
        const float z = (u - mean) * inv_std, zq = u / S;
        logp += (-0.5f * z * z - ls - 0.9189385332046727f) - ((-0.5f * zq * zq - LOG_SQUASH_SCALE - 0.9189385332046727f) + p.log_range_f32);
        dlp_m = z * inv_std; dlp_s = z * z - 1.0f;
        ent += p.log_range_f32 - (LOG_SQUASH_SCALE - ls + (std * std + mean * mean) / (2.0f * S * S) - 0.5f);
        dh_m = -mean / (S * S); dh_s = 1.0f - std * std / (S * S);
        const float dm = mean_o - mean, q = (std_o * std_o + dm * dm) * inv_std * inv_std;
        kl += ls - ls_o + 0.5f * q - 0.5f;
        dk_m = -dm * inv_std * inv_std; dk_s = 1.0f - q;
    }
    const float ratio = expf(logp - in.logp_old), ad = in.adv;
    const float s1 = ad * ratio, s2 = ad * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
    const float surr = fminf(s1, s2);
    const float c_lp = s1 <= s2 ? -s1 : 0.0f;                                    // d(-surrogate) / dlogp
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= nk) break;
        const float gd = (c_lp * dlp[k] + klc * dk[k] - ent_coeff * dh[k]) * inv_b;
        g[2 * k] = -gd; g[2 * k + 1] = gd;
    }
    if (yaw_mode == 1) {
        g[2 * nk] = in_m ? (c_lp * dlp_m + klc * dk_m - ent_coeff * dh_m) * inv_b : 0.0f;
        g[2 * nk + 1] = in_s ? (c_lp * dlp_s + klc * dk_s - ent_coeff * dh_s) * inv_b : 0.0f;
    }
    if (!FIXED && yaw_mode == 2) {
        const float* l = row + 2 * nk;
        const float* lo = old + 2 * nk;
        for (int j = 0; j < cat_m; ++j) {
            const float lpn = l[j] - cat_lse, pn = expf(lpn), po = expf(lo[j] - cat_lse_o);
            const float dlp_j = (j == cat_a ? 1.0f : 0.0f) - pn, dh_j = -pn * (lpn + cat_h), dk_j = pn - po;
            g[2 * nk + j] = (c_lp * dlp_j + klc * dk_j - ent_coeff * dh_j) * inv_b;
        }
    }
    if (!FIXED)
        for (int c = 2 * nk + (yaw_mode == 1 ? 2 : cat_m); c < row_stride; ++c) g[c] = 0.0f;
    return PpoSums{ent, kl, surr};
}

// The value part of one sample: vf (returned through the reference) and d vf / d v
__device__ __forceinline__ float ppo_value_grad(float v, float vo, float vt, float vf_clip, float& vf) {
    const float dv = v - vo, vc = vo + fminf(fmaxf(dv, -vf_clip), vf_clip);
    const float e1 = (v - vt) * (v - vt), e2 = (vc - vt) * (vc - vt);
    vf = fmaxf(e1, e2);
    return e1 >= e2 ? 2.0f * (v - vt) : ((dv >= -vf_clip && dv <= vf_clip) ? 2.0f * (vc - vt) : 0.0f);
}

template <bool GATHER>
__global__ void __launch_bounds__(256)
ppo_loss_grad_kernel(Params p, int batch, const float* __restrict__ logits, const float* __restrict__ old_logits, int row_stride, int old_stride,
                     const uint8_t* __restrict__ keys, const float* __restrict__ mouse, const float* __restrict__ logp_old,
                     const float* __restrict__ adv, const float* __restrict__ value, const float* __restrict__ value_old,
                     const float* __restrict__ vtarg, const int64_t* __restrict__ idx, const int64_t* __restrict__ idx_cursor, float clip, float vf_clip,
                     float vf_coeff, float ent_coeff,
                     const float* __restrict__ kl_coeff_dev, float out_scale, float out_scale_v, float* __restrict__ dlogits, float* __restrict__ dvalue,
                     float* __restrict__ partials) {
    __shared__ float red[4][5];
    if (GATHER && idx_cursor) idx += *idx_cursor;       // rows idx[cursor .. cursor + batch) (q1env_learner_batch.idx_cursor_dev)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < (uint32_t)batch;
    const float klc = *kl_coeff_dev;
    const float inv_b = out_scale / (float)batch;
    float st[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
        const size_t src = GATHER ? (size_t)idx[i] : (size_t)i;
        const float* row = logits + (size_t)i * row_stride;
        const float* old = old_logits + src * (size_t)old_stride;
        float* g = dlogits + (size_t)i * row_stride;
        const PpoSample in{keys[src], p.yaw_mode != 0 ? mouse[src] : 0.0f, logp_old[src], adv[src]};
        const PpoSums ps = ppo_policy_grad<false>(p, row, old, in, clip, ent_coeff, klc, inv_b, g, row_stride);
        float vf;
        const float dvf = ppo_value_grad(value[i], value_old[src], vtarg[src], vf_clip, vf);
        dvalue[i] = vf_coeff * dvf * (out_scale_v / (float)batch);
        st[0] = ps.ent; st[1] = ps.kl; st[2] = -ps.surr; st[3] = -ps.surr + klc * ps.kl + vf_coeff * vf - ent_coeff * ps.ent; st[4] = vf;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) st[k] += __shfl_down(st[k], off, 64);
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = st[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) partials[(size_t)blockIdx.x * 5 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

