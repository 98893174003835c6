// q1env_learner.hip - the native PPO learner step of libq1env.so (q1env_learner_*; device code in q1learner.hpp and q1ppo_loss.hpp):
// forward + loss gradient + backward + weight gradients of the policy and the value network on the matrix cores, float32 master
// weights and gradients in torch layouts.  Counterpart of the torch modules + autograd in q1physrl_amd/ppo.py (reference: RLlib's
// PPOTrainer as configured by q1physrl/train.py:60-64 and data/params.yml:4-13).
#include "q1env_host.hpp"
#include "q1policy.hpp"
#include "q1policy_glue.hpp"
#include "q1ppo_loss.hpp"
#include "q1learner.hpp"
#include "q1learner_fused.hpp"

using namespace q1;

namespace {

// q1env_learner_step's float16 loss scales: the policy network's per-sample gradients travel multiplied by B x pi_upscale, the value
// network's by B / value_downscale (q1learner.hpp "Gradient scaling").  Defaults from round 3's training regressions (profiles/
// r3_train_ppo_native_*.json): pi_upscale = 256 - the policy head starts with weights of 1e-2 x (RLlib's
// normc_initializer(0.01)), so dZ2 = W3^T dY and dZ1 behind it sat at 1e-3 .. 1e-6, partly in float16's subnormal range, and one seed
// of five plateaued at 5 230 instead of ~5 650 until the scale lifted them; value_downscale = 1 (B / 64 cost two of three seeds the
// same way from the other side).  Outliers saturate (cvt8_sat).  Changed per handle by q1env_learner_set_loss_scale, not by the environment.
static float learner_pi_upscale(const q1env* h) { return h->pi_upscale > 0.0f ? h->pi_upscale : 256.0f; }             // (q1env_learner_set_loss_scale)
static float learner_value_downscale(const q1env* h) { return h->value_downscale > 0.0f ? h->value_downscale : 1.0f; }
constexpr size_t IMG_FWD_BYTES = (q1pol::LDS_W2 + q1pol::LDS_W3);           // 152064: float16[288][264]
constexpr size_t IMG_W2T_BYTES = q1learn::LDS_W2T;                          // 135168
constexpr size_t IMG_W3T_BYTES = q1learn::LDS_W3T;                          // 20480

constexpr int64_t FUSED_MIN_BATCH = 2048;      // q1env_learner_sgd_step takes the fused forward + backward kernel from here on (automatic mode)
constexpr size_t STATS_ROWS = 2048;      // >= the backward kernel's workgroups (at most one per CU)

struct NetWs {
    uint16_t* w23; uint16_t* w2t; uint16_t* w3t;
    q1learn::f16x8* h1T; q1learn::f16x8* h2T; q1learn::f16x8* dz2N; q1learn::f16x8* dz1N;
    q1learn::f16x8* xN; q1learn::f16x8* dyN;
    float* partial;
    float* dw1p;                  // the fused forward + backward kernel's per-tile dW1 / db1 products (q1learner_fused.hpp, DW1)
    float4* dw3a; float* dw3b;    // ... and its per-tile dW3 products (dw3a: the policy network only)
};
struct Ws {
    NetWs net[2];
    float* logits; float* value; float* dlogits; float* dvalue;
    float* stats_rows;            // float[STATS_ROWS][5]: the fused step's statistics rows (one per workgroup of the backward kernel)
    size_t bytes;
};

// carve the caller's workspace (256-byte aligned pieces); base may be NULL to size it
Ws carve_ws(void* base, int64_t mb, int out_pi, int splits) {
    Ws w{};
    char* b = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* q = b ? b + off : nullptr; off += align_up(bytes, 256); return q; };
    const size_t tiles = (size_t)((mb + 31) / 32), act = tiles * q1learn::TILE_VECS * 16u;
    for (int k = 0; k < 2; ++k) {
        NetWs& n = w.net[k];
        n.w23 = (uint16_t*)take(IMG_FWD_BYTES); n.w2t = (uint16_t*)take(IMG_W2T_BYTES); n.w3t = (uint16_t*)take(IMG_W3T_BYTES);
        n.h1T = (q1learn::f16x8*)take(act); n.h2T = (q1learn::f16x8*)take(act);
        n.dz2N = (q1learn::f16x8*)take(act); n.dz1N = (q1learn::f16x8*)take(act);
        n.xN = (q1learn::f16x8*)take(tiles * 128u * 16u); n.dyN = (q1learn::f16x8*)take(tiles * 128u * 16u);
        n.partial = (float*)take((size_t)splits * q1learn::PARTIAL_STRIDE * 4u);
    }
    w.logits = (float*)take((size_t)mb * out_pi * 4u); w.value = (float*)take((size_t)mb * 4u);
    w.dlogits = (float*)take((size_t)mb * out_pi * 4u); w.dvalue = (float*)take((size_t)mb * 4u);
    w.stats_rows = (float*)take(STATS_ROWS * 5u * 4u);
    for (int k = 0; k < 2; ++k) w.net[k].dw1p = (float*)take(tiles * q1learn::DW1_TILE_FLOATS * 4u);      // (behind everything round 4 laid out)
    w.net[0].dw3a = (float4*)take(tiles * 8u * 64u * 16u); w.net[1].dw3a = (float4*)take(256);      // (value network: a dummy the kernel's branch-free store hits)
    for (int k = 0; k < 2; ++k) w.net[k].dw3b = (float*)take(tiles * 8u * 64u * 4u);
    w.bytes = off;
    return w;
}

int check_nets(const char* who, const q1env_learner_net* pi, const q1env_learner_net* vf, bool need_grads) {
    for (const q1env_learner_net* m : {pi, vf}) {
        if (!m || !m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3) return fail(Q1ENV_ERR_INVALID_ARG, std::string(who) + ": null weight pointer");
        if (need_grads && (!m->gw1 || !m->gb1 || !m->gw2 || !m->gb2 || !m->gw3 || !m->gb3)) return fail(Q1ENV_ERR_INVALID_ARG, std::string(who) + ": null gradient pointer");
        if (m->out_dim < 1 || m->out_dim > 32) return fail(Q1ENV_ERR_INVALID_ARG, std::string(who) + ": out_dim must be in 1..32");
    }
    return 0;
}

int check_shape(const char* who, int64_t mb, int splits) {
    if (mb <= 0 || mb > ((int64_t)1 << 24)) return fail(Q1ENV_ERR_INVALID_ARG, std::string(who) + ": minibatch out of range");
    if (splits < 1 || splits > 512) return fail(Q1ENV_ERR_INVALID_ARG, std::string(who) + ": splits must be in 1..512");
    return 0;
}

int ensure_learner_attrs(q1env* h) {
    if (!h->learner_attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)q1learn::learner_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1pol::LDS_TOTAL));
        HIP_TRY(hipFuncSetAttribute((const void*)q1learn::learner_backward_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1learn::LDS_BWD));
        HIP_TRY(hipFuncSetAttribute((const void*)q1learn::learner_backward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1learn::LDS_BWD));
        HIP_TRY(hipFuncSetAttribute((const void*)q1learn::learner_fwdbwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1learn::LDS_FZ));
        HIP_TRY(hipFuncSetAttribute((const void*)q1learn::learner_fwdbwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1learn::LDS_FZ));
        h->learner_attr_set = true;
    }
    return 0;
}

int launch_forward(q1env* h, const Ws& w, int64_t mb, const q1env_learner_net* pi, const q1env_learner_net* vf, const float* obs, const int64_t* idx,
                   const int64_t* idx_cursor,
                   float* logits, float* value) {
    if (int r = ensure_learner_attrs(h)) return r;
    const q1learn::FwdNet na{pi->w1, pi->b1, w.net[0].w23, pi->b2, pi->b3, logits, pi->out_dim, w.net[0].h1T, w.net[0].h2T};
    const q1learn::FwdNet nb{vf->w1, vf->b1, w.net[1].w23, vf->b2, vf->b3, value, vf->out_dim, w.net[1].h1T, w.net[1].h2T};
    const unsigned cus = (unsigned)(h->num_cus > 1 ? h->num_cus / 2 : 1);                  // CUs per network, one workgroup each
    const unsigned tiles = (unsigned)((mb + 31) / 32);
    unsigned blocks = (tiles + 7u) / 8u;
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(q1learn::learner_forward_kernel, dim3(blocks * 2u), dim3(512), q1pol::LDS_TOTAL, h->stream, (int)mb, obs, idx, idx_cursor, na, nb, 2);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_backward(q1env* h, const Ws& w, int64_t mb, int splits, const q1env_learner_net* pi, const q1env_learner_net* vf, const float* obs,
                    const int64_t* idx, const int64_t* idx_cursor, const float* dlogits, const float* dvalue, float grad_scale, float grad_scale_v,
                    bool reduce = true, uint32_t* sat = nullptr, const q1learn::LossArgs* fused = nullptr, const q1learn::BcArgs* bca = nullptr,
                    unsigned* grid_out = nullptr) {
    if (int r = ensure_learner_attrs(h)) return r;
    const q1learn::BwdNet ba{w.net[0].w2t, w.net[0].w3t, dlogits, pi->out_dim, pi->out_dim, w.net[0].h1T, w.net[0].h2T,
                             w.net[0].dz2N, w.net[0].dz1N, w.net[0].xN, w.net[0].dyN, sat};
    const q1learn::BwdNet bb{w.net[1].w2t, w.net[1].w3t, dvalue, vf->out_dim, vf->out_dim, w.net[1].h1T, w.net[1].h2T,
                             w.net[1].dz2N, w.net[1].dz1N, w.net[1].xN, w.net[1].dyN, sat ? sat + 2 : nullptr};
    const unsigned cus = (unsigned)(h->num_cus > 1 ? h->num_cus / 2 : 1);
    const unsigned tiles = (unsigned)((mb + 31) / 32);
    unsigned blocks = (tiles + 3u) / 4u;
    if (blocks > cus) blocks = cus;
    if (grid_out) *grid_out = blocks * 2u;
    if (blocks * 2u > STATS_ROWS) return fail(Q1ENV_ERR_INVALID_ARG, "learner: more workgroups than statistics rows");
    if (fused)          // dY computed in the kernel (q1env_learner_sgd_step; the reference's action structure)
        hipLaunchKernelGGL(q1learn::learner_backward_kernel<true>, dim3(blocks * 2u), dim3(256), q1learn::LDS_BWD, h->stream, (int)mb, obs, idx, idx_cursor, ba,
                           bb, 2, *fused, bca ? *bca : q1learn::BcArgs{nullptr, nullptr, 0.0f, 0.0f});
    else
        hipLaunchKernelGGL(q1learn::learner_backward_kernel<false>, dim3(blocks * 2u), dim3(256), q1learn::LDS_BWD, h->stream, (int)mb, obs, idx, idx_cursor, ba,
                           bb, 2, q1learn::LossArgs{}, q1learn::BcArgs{nullptr, nullptr, 0.0f, 0.0f});
    const q1learn::WgNet wa{w.net[0].dz2N, w.net[0].dz1N, w.net[0].h1T, w.net[0].h2T, w.net[0].xN, w.net[0].dyN, w.net[0].partial, nullptr, nullptr, nullptr};
    const q1learn::WgNet wb{w.net[1].dz2N, w.net[1].dz1N, w.net[1].h1T, w.net[1].h2T, w.net[1].xN, w.net[1].dyN, w.net[1].partial, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(q1learn::learner_wgrad_kernel<false>, dim3(2u * (unsigned)splits, 2, 2), dim3(256), 0, h->stream, (int)mb, wa, wb, splits);
    if (!reduce) { HIP_TRY(hipGetLastError()); return 0; }          // q1env_learner_adam sums the partials itself
    const q1learn::Grads ga{pi->gw1, pi->gb1, pi->gw2, pi->gb2, pi->gw3, pi->gb3, pi->out_dim};
    const q1learn::Grads gb{vf->gw1, vf->gb1, vf->gw2, vf->gb2, vf->gw3, vf->gb3, vf->out_dim};
    const unsigned elems = (unsigned)q1learn::PARTIAL_FLOATS;            // one thread per slot of the partial-sum slab, in the slab's order
    hipLaunchKernelGGL(q1learn::learner_reduce_kernel, dim3((elems + 255u) / 256u, 2), dim3(256), 0, h->stream, (const float*)w.net[0].partial,
                       (const float*)w.net[1].partial, ga, gb, splits, 1.0f / grad_scale, 1.0f / grad_scale_v, 0);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The fused step's first two launches (round 6): forward + loss gradient + data gradients in one kernel (q1learner_fused.hpp), then the
// weight-gradient kernel.  dw1: dZ1 is replaced by the per-tile dW1 / db1 products.  One workgroup per eight 32-sample tiles and network.
int launch_fwdbwd(q1env* h, const Ws& w, int64_t mb, int splits, const q1env_learner_net* pi, const q1env_learner_net* vf, const float* obs, const int64_t* idx,
                  const int64_t* idx_cursor, uint32_t* sat, const q1learn::LossArgs& la, const q1learn::BcArgs& bca, bool dw1, unsigned* grid_out) {
    if (int r = ensure_learner_attrs(h)) return r;
    const q1learn::FzNet fa{pi->w1, pi->b1, w.net[0].w23, pi->b2, pi->b3, w.net[0].w2t, w.net[0].w3t, w.net[0].h1T, w.net[0].h2T,
                            w.net[0].dz2N, w.net[0].dz1N, w.net[0].xN, w.net[0].dyN, w.net[0].dw1p, w.net[0].dw3a, w.net[0].dw3b, sat};
    const q1learn::FzNet fb{vf->w1, vf->b1, w.net[1].w23, vf->b2, vf->b3, w.net[1].w2t, w.net[1].w3t, w.net[1].h1T, w.net[1].h2T,
                            w.net[1].dz2N, w.net[1].dz1N, w.net[1].xN, w.net[1].dyN, w.net[1].dw1p, w.net[1].dw3a, w.net[1].dw3b, sat ? sat + 2 : nullptr};
    const unsigned tiles = (unsigned)((mb + 31) / 32);
    const unsigned blocks = (tiles + 7u) / 8u;
    if (blocks * 2u > STATS_ROWS) return fail(Q1ENV_ERR_INVALID_ARG, "learner: more workgroups than statistics rows");
    *grid_out = blocks * 2u;
    const q1learn::WgNet wa{w.net[0].dz2N, w.net[0].dz1N, w.net[0].h1T, w.net[0].h2T, w.net[0].xN, w.net[0].dyN, w.net[0].partial, w.net[0].dw1p, w.net[0].dw3a, w.net[0].dw3b};
    const q1learn::WgNet wb{w.net[1].dz2N, w.net[1].dz1N, w.net[1].h1T, w.net[1].h2T, w.net[1].xN, w.net[1].dyN, w.net[1].partial, w.net[1].dw1p, nullptr, w.net[1].dw3b};
    if (dw1) {
        hipLaunchKernelGGL(q1learn::learner_fwdbwd_kernel<true>, dim3(blocks * 2u), dim3(512), q1learn::LDS_FZ, h->stream, (int)mb, obs, idx, idx_cursor, fa, fb, la, bca);
#if defined(Q1_FZ_STAMPS) && (Q1_FZ_EXP & 512)       // diagnostic build: the same launch once more (idempotent), so that the second one's stamps see the first one's end
        hipLaunchKernelGGL(q1learn::learner_fwdbwd_kernel<true>, dim3(blocks * 2u), dim3(512), q1learn::LDS_FZ, h->stream, (int)mb, obs, idx, idx_cursor, fa, fb, la, bca);
#endif
        if (h->wgrad_variant == 1)          // (experiment, tools/time_learner.py --step-mode fused_dw1_q: column quarters, two workgroups per CU - 35.2 us against 27.6)
            hipLaunchKernelGGL((q1learn::learner_wgrad_kernel<true, 2>), dim3(2u * (unsigned)splits, 4, 2), dim3(256), 0, h->stream, (int)mb, wa, wb, splits);
        else if (h->wgrad_variant == 2)     // (experiment: the round-4 kernel on the product arrays)
            hipLaunchKernelGGL((q1learn::learner_wgrad_kernel<true, 4>), dim3(2u * (unsigned)splits, 2, 2), dim3(256), 0, h->stream, (int)mb, wa, wb, splits);
        else
            hipLaunchKernelGGL(q1learn::learner_wgrad_shared_kernel, dim3(2u * (unsigned)splits, 2, 2), dim3(256), 0, h->stream, (int)mb, wa, wb, splits);
    } else {
        hipLaunchKernelGGL(q1learn::learner_fwdbwd_kernel<false>, dim3(blocks * 2u), dim3(512), q1learn::LDS_FZ, h->stream, (int)mb, obs, idx, idx_cursor, fa, fb, la, bca);
        hipLaunchKernelGGL(q1learn::learner_wgrad_kernel<false>, dim3(2u * (unsigned)splits, 2, 2), dim3(256), 0, h->stream, (int)mb, wa, wb, splits);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int q1env_learner_set_loss_scale(q1env_t* h, float pi_upscale, float value_downscale) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_set_loss_scale: null handle");
    auto pow2 = [](float x) { int e = 0; return x > 0.0f && frexpf(x, &e) == 0.5f; };
    if ((pi_upscale != 0.0f && !pow2(pi_upscale)) || (value_downscale != 0.0f && !pow2(value_downscale)))
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_set_loss_scale: scales must be 0 (default) or exact powers of two");
    h->pi_upscale = pi_upscale;
    h->value_downscale = value_downscale;
    return Q1ENV_OK;
}

int q1env_learner_set_step_mode(q1env_t* h, int mode) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_set_step_mode: null handle");
    if (mode == 4 || mode == 5) { h->learner_step_mode = 3; h->wgrad_variant = mode - 3; return Q1ENV_OK; }      // (measurement only: mode 3 with an older weight-gradient kernel)
    h->wgrad_variant = 0;
    if (mode < 0 || mode > 3)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_set_step_mode: mode must be 0 (automatic), 1 (four launches), 2 (fused forward + backward) or 3 (fused, dW1 products)");
    h->learner_step_mode = mode;
    return Q1ENV_OK;
}

uint64_t q1env_learner_workspace_bytes(int64_t minibatch, int out_dim_pi, int splits) {
    if (minibatch <= 0 || out_dim_pi < 1 || out_dim_pi > 32 || splits < 1 || splits > 512) return 0;
    return (uint64_t)carve_ws(nullptr, minibatch, out_dim_pi, splits).bytes;
}

int q1env_learner_images(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int64_t minibatch, int splits) {
    if (!h || !ws_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_images: null argument");
    if (int r = check_nets("q1env_learner_images", pi, vf, false)) return r;
    if (int r = check_shape("q1env_learner_images", minibatch, splits)) return r;
    DeviceGuard guard(h->device);
    const Ws w = carve_ws(ws_dev, minibatch, pi->out_dim, splits);
    const q1learn::ImgNet na{pi->w2, pi->w3, pi->out_dim, w.net[0].w23, w.net[0].w2t, w.net[0].w3t};
    const q1learn::ImgNet nb{vf->w2, vf->w3, vf->out_dim, w.net[1].w23, w.net[1].w2t, w.net[1].w3t};
    const unsigned elems = 288u * 264u + 256u * 264u + 256u * 40u;
    hipLaunchKernelGGL(q1learn::learner_images_kernel, dim3((elems + 255u) / 256u, 2), dim3(256), 0, h->stream, na, nb);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

int q1env_learner_forward(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int64_t minibatch, int splits,
                          const float* obs_dev, const int64_t* idx_dev, float* logits_out, float* value_out) {
    if (!h || !ws_dev || !obs_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_forward: null argument");
    if (int r = check_nets("q1env_learner_forward", pi, vf, false)) return r;
    if (int r = check_shape("q1env_learner_forward", minibatch, splits)) return r;
    DeviceGuard guard(h->device);
    const Ws w = carve_ws(ws_dev, minibatch, pi->out_dim, splits);
    if (int r = launch_forward(h, w, minibatch, pi, vf, obs_dev, idx_dev, nullptr, logits_out ? logits_out : w.logits, value_out ? value_out : w.value)) return r;
    return Q1ENV_OK;
}

int q1env_learner_backward(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int64_t minibatch, int splits,
                           const float* obs_dev, const int64_t* idx_dev, const float* dlogits_dev, const float* dvalue_dev, float grad_scale) {
    if (!h || !ws_dev || !obs_dev || !dlogits_dev || !dvalue_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_backward: null argument");
    if (!(grad_scale > 0.0f)) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_backward: grad_scale must be > 0");
    if (int r = check_nets("q1env_learner_backward", pi, vf, true)) return r;
    if (int r = check_shape("q1env_learner_backward", minibatch, splits)) return r;
    DeviceGuard guard(h->device);
    const Ws w = carve_ws(ws_dev, minibatch, pi->out_dim, splits);
    return launch_backward(h, w, minibatch, splits, pi, vf, obs_dev, idx_dev, nullptr, dlogits_dev, dvalue_dev, grad_scale, grad_scale);
}

int q1env_learner_step(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int splits, const q1env_learner_batch* b) {
    if (!h || !ws_dev || !b) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_step: null argument");
    if (!b->obs_dev || !b->old_logits_dev || !b->keys_dev || !b->logp_old_dev || !b->adv_dev || !b->value_old_dev || !b->vtarg_dev || !b->kl_coeff_dev ||
        !b->stats_partials_dev)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_step: null pointer in q1env_learner_batch");
    if (int r = check_nets("q1env_learner_step", pi, vf, true)) return r;
    if (int r = check_shape("q1env_learner_step", b->minibatch, splits)) return r;
    if (h->p.yaw_mode != 0 && !b->mouse_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_step: mouse actions required");
    if (pi->out_dim != policy_row_width(h->p)) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_step: pi->out_dim must be " + std::to_string(policy_row_width(h->p)));
    if (vf->out_dim != 1) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_step: vf->out_dim must be 1");
    if (b->old_stride < pi->out_dim) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_step: old_stride smaller than the policy row");
    DeviceGuard guard(h->device);
    const int64_t mb = b->minibatch;
    const Ws w = carve_ws(ws_dev, mb, pi->out_dim, splits);
    if (int r = launch_forward(h, w, mb, pi, vf, b->obs_dev, b->idx_dev, b->idx_cursor_dev, w.logits, w.value)) return r;
    // per-sample (un-averaged) gradients x learner_pi_upscale(h) for the policy, / learner_value_downscale(h) (default 1) for the value
    // network: float16's normal range (q1learner.hpp)
    const float scale = (float)mb * learner_pi_upscale(h), scale_v = (float)mb / learner_value_downscale(h);
    if (b->idx_dev)
        hipLaunchKernelGGL(ppo_loss_grad_kernel<true>, grid_for((int)mb, 256), dim3(256), 0, h->stream, h->p, (int)mb, (const float*)w.logits,
                           b->old_logits_dev, pi->out_dim, b->old_stride, b->keys_dev, b->mouse_dev, b->logp_old_dev, b->adv_dev, (const float*)w.value,
                           b->value_old_dev, b->vtarg_dev, b->idx_dev, b->idx_cursor_dev, b->clip_param, b->vf_clip_param, b->vf_loss_coeff, b->entropy_coeff,
                           b->kl_coeff_dev, scale, scale_v, w.dlogits, w.dvalue, b->stats_partials_dev);
    else
        hipLaunchKernelGGL(ppo_loss_grad_kernel<false>, grid_for((int)mb, 256), dim3(256), 0, h->stream, h->p, (int)mb, (const float*)w.logits,
                           b->old_logits_dev, pi->out_dim, b->old_stride, b->keys_dev, b->mouse_dev, b->logp_old_dev, b->adv_dev, (const float*)w.value,
                           b->value_old_dev, b->vtarg_dev, (const int64_t*)nullptr, (const int64_t*)nullptr, b->clip_param, b->vf_clip_param, b->vf_loss_coeff, b->entropy_coeff,
                           b->kl_coeff_dev, scale, scale_v, w.dlogits, w.dvalue, b->stats_partials_dev);
    HIP_TRY(hipGetLastError());
    return launch_backward(h, w, mb, splits, pi, vf, b->obs_dev, b->idx_dev, b->idx_cursor_dev, w.dlogits, w.dvalue, scale, scale_v, b->skip_reduce == 0,
                           b->saturation_dev);
}

uint64_t q1env_learner_adam_state_bytes(int out_dim_pi) {
    if (out_dim_pi < 1 || out_dim_pi > 32) return 0;
    const size_t per_pi = 65536u + 256u + 1536u + 256u + (size_t)out_dim_pi * 257u, per_vf = 65536u + 256u + 1536u + 256u + 257u;
    return (uint64_t)(256u + align_up(2 * per_pi * 4u, 256) + align_up(2 * per_vf * 4u, 256));
}

}  // extern "C"

namespace {

// q1env_learner_adam (fused_tick = false: the bookkeeping kernel runs in front) and the tail of q1env_learner_sgd_step (fused_tick: the
// bias corrections were left by the backward kernel, workgroup (0, 0) of the Adam kernel does the rest; stats = the backward kernel's
// `stat_rows` statistics rows)
int launch_adam(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int64_t minibatch, int splits,
                float grad_scale, float lr, float beta1, float beta2, float eps, void* adam_state_dev, const float* stats_partials_dev, bool fused_tick,
                int stat_rows, bool dw1 = false) {
    if (!h || !ws_dev || !adam_state_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_adam: null argument");
    if (!(grad_scale > 0.0f) || !(lr >= 0.0f) || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps > 0.0f))
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_adam: bad hyper-parameter");
    if (int r = check_nets("q1env_learner_adam", pi, vf, true)) return r;
    if (int r = check_shape("q1env_learner_adam", minibatch, splits)) return r;
    // the state block is sized by q1env_learner_adam_state_bytes(out_dim_pi), which assumes a scalar value head
    if (vf->out_dim != 1) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_adam: vf->out_dim must be 1");
    if (pi->out_dim < 1 || pi->out_dim > 32) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_adam: pi->out_dim must be in 1..32");
    DeviceGuard guard(h->device);
    const Ws w = carve_ws(ws_dev, minibatch, pi->out_dim, splits);
    char* st = (char*)adam_state_dev;
    long long* step = (long long*)st;
    float* bc = (float*)(st + 8);
    const size_t per_pi = 65536u + 256u + 1536u + 256u + (size_t)pi->out_dim * 257u, per_vf = 65536u + 256u + 1536u + 256u + (size_t)vf->out_dim * 257u;
    float* m_pi = (float*)(st + 256), *v_pi = m_pi + per_pi;
    float* m_vf = (float*)(st + 256 + align_up(2 * per_pi * 4u, 256)), *v_vf = m_vf + per_vf;
    q1learn::AdamTick tick{nullptr, nullptr, 0, nullptr, 0, 0.0f, nullptr, 0};
    if (fused_tick)
        tick = q1learn::AdamTick{step, (long long*)(st + 72), (long long)minibatch, stats_partials_dev, stat_rows, 1.0f / (float)minibatch, (float*)(st + 16), dw1 ? 1 : 0};
    else
        hipLaunchKernelGGL(q1learn::adam_tick_kernel, dim3(1), dim3(64), 0, h->stream, step, bc, beta1, beta2, stats_partials_dev,
                           (int)((minibatch + 255) / 256), 1.0f / (float)minibatch, (float*)(st + 16), (long long*)(st + 72), (long long)minibatch);
    const q1learn::AdamNet na{const_cast<float*>(pi->w1), const_cast<float*>(pi->b1), const_cast<float*>(pi->w2), const_cast<float*>(pi->b2),
                              const_cast<float*>(pi->w3), const_cast<float*>(pi->b3),
                              q1learn::Grads{pi->gw1, pi->gb1, pi->gw2, pi->gb2, pi->gw3, pi->gb3, pi->out_dim}, m_pi, v_pi, w.net[0].w23, w.net[0].w2t, w.net[0].w3t};
    const q1learn::AdamNet nb{const_cast<float*>(vf->w1), const_cast<float*>(vf->b1), const_cast<float*>(vf->w2), const_cast<float*>(vf->b2),
                              const_cast<float*>(vf->w3), const_cast<float*>(vf->b3),
                              q1learn::Grads{vf->gw1, vf->gb1, vf->gw2, vf->gb2, vf->gw3, vf->gb3, vf->out_dim}, m_vf, v_vf, w.net[1].w23, w.net[1].w2t, w.net[1].w3t};
    const unsigned elems = (unsigned)q1learn::PARTIAL_FLOATS;            // one thread per slot of the partial-sum slab, in the slab's order
    hipLaunchKernelGGL(q1learn::learner_adam_kernel, dim3((elems + 255u) / 256u, 2), dim3(256), 0, h->stream, (const float*)w.net[0].partial,
                       (const float*)w.net[1].partial, na, nb, splits, 1.0f / (grad_scale * learner_pi_upscale(h)), learner_value_downscale(h) / grad_scale,
                       q1learn::AdamHyper{lr, beta1, beta2, eps}, (const float*)bc, tick);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}

}  // namespace

extern "C" {

int q1env_learner_adam(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int64_t minibatch, int splits,
                       float grad_scale, float lr, float beta1, float beta2, float eps, void* adam_state_dev, const float* stats_partials_dev) {
    return launch_adam(h, pi, vf, ws_dev, minibatch, splits, grad_scale, lr, beta1, beta2, eps, adam_state_dev, stats_partials_dev, false, 0);
}

int q1env_learner_sgd_step(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* ws_dev, int splits, const q1env_learner_batch* b,
                           float lr, float beta1, float beta2, float eps, void* adam_state_dev) {
    if (!h || !ws_dev || !b || !adam_state_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_step: null argument");
    if (!b->obs_dev || !b->old_logits_dev || !b->keys_dev || !b->logp_old_dev || !b->adv_dev || !b->value_old_dev || !b->vtarg_dev || !b->kl_coeff_dev)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_step: null pointer in q1env_learner_batch");
    if (!(lr >= 0.0f) || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps > 0.0f))
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_step: bad hyper-parameter");
    if (int r = check_nets("q1env_learner_sgd_step", pi, vf, true)) return r;
    if (int r = check_shape("q1env_learner_sgd_step", b->minibatch, splits)) return r;
    if (h->p.yaw_mode != 0 && !b->mouse_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_step: mouse actions required");
    if (pi->out_dim != policy_row_width(h->p)) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_step: pi->out_dim must be " + std::to_string(policy_row_width(h->p)));
    if (vf->out_dim != 1) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_step: vf->out_dim must be 1");
    if (b->old_stride < pi->out_dim) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_step: old_stride smaller than the policy row");
    // the in-kernel loss gradient is written for the reference's action structure (4 keys + continuous mouse: 10 outputs); any other
    // Config takes the same path as q1env_learner_step + q1env_learner_adam
    const bool fixed = h->p.num_keys == 4 && h->p.yaw_mode == 1 && pi->out_dim == 10;
    if (!fixed) {
        if (!b->stats_partials_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_step: stats_partials_dev is required for this action structure");
        q1env_learner_batch bb = *b;
        bb.skip_reduce = 1;
        if (int r = q1env_learner_step(h, pi, vf, ws_dev, splits, &bb)) return r;
        return q1env_learner_adam(h, pi, vf, ws_dev, b->minibatch, splits, (float)b->minibatch, lr, beta1, beta2, eps, adam_state_dev, b->stats_partials_dev);
    }
    DeviceGuard guard(h->device);
    const int64_t mb = b->minibatch;
    const Ws w = carve_ws(ws_dev, mb, pi->out_dim, splits);
    // the kernel sequence (q1env_learner_set_step_mode): the fused forward + backward kernel from FUSED_MIN_BATCH samples on
    int mode = h->learner_step_mode;
    // (one statistics row per workgroup of eight tiles and network: a minibatch beyond STATS_ROWS / 2 such workgroups - 262 144 samples - keeps the
    // four launches, whose backward kernel walks its tiles grid-stride)
    const bool fused_fits = ((mb + 31) / 32 + 7) / 8 * 2 <= (int64_t)STATS_ROWS;
    if (mode == 0) mode = (mb >= FUSED_MIN_BATCH && fused_fits) ? 3 : 1;
    if (mode != 1 && !fused_fits) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_step: the fused kernel serves minibatches up to 262 144 samples (step mode 0 or 1 beyond)");
    if (mode == 1)
        if (int r = launch_forward(h, w, mb, pi, vf, b->obs_dev, b->idx_dev, b->idx_cursor_dev, w.logits, w.value)) return r;
    const float scale = (float)mb * learner_pi_upscale(h), scale_v = (float)mb / learner_value_downscale(h);
    q1learn::LossArgs la{};
    la.p = h->p;
    la.logits = w.logits; la.value = w.value; la.old_logits = b->old_logits_dev; la.old_stride = b->old_stride;
    la.keys = b->keys_dev; la.mouse = b->mouse_dev; la.logp_old = b->logp_old_dev; la.adv = b->adv_dev; la.value_old = b->value_old_dev; la.vtarg = b->vtarg_dev;
    la.kl_coeff_dev = b->kl_coeff_dev;
    la.clip = b->clip_param; la.vf_clip = b->vf_clip_param; la.vf_coeff = b->vf_loss_coeff; la.ent_coeff = b->entropy_coeff;
    la.inv_b = scale / (float)mb; la.inv_bv = scale_v / (float)mb;
    la.stats_rows = w.stats_rows;
    la.wide = ((uintptr_t)b->old_logits_dev % 8u == 0 && b->old_stride % 2 == 0 && (uintptr_t)b->obs_dev % 8u == 0) ? 1 : 0;    // (w.logits is 256-byte aligned, rows of 40 B)
    char* st = (char*)adam_state_dev;
    const q1learn::BcArgs bca{(const long long*)st, (float*)(st + 8), beta1, beta2};
    unsigned rows = 0;
    if (mode == 1) {
        if (int r = launch_backward(h, w, mb, splits, pi, vf, b->obs_dev, b->idx_dev, b->idx_cursor_dev, nullptr, nullptr, scale, scale_v, false, b->saturation_dev, &la,
                                    &bca, &rows))
            return r;
    } else {
        if (int r = launch_fwdbwd(h, w, mb, splits, pi, vf, b->obs_dev, b->idx_dev, b->idx_cursor_dev, b->saturation_dev, la, bca, mode == 3, &rows)) return r;
    }
    return launch_adam(h, pi, vf, ws_dev, mb, splits, (float)mb, lr, beta1, beta2, eps, adam_state_dev, w.stats_rows, true, (int)rows, mode == 3);
}

}  // extern "C"
