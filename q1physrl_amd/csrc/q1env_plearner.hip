// q1env_plearner.hip - the PERSISTENT PPO learner of libq1env.so (q1env_learner_sgd_epochs; device code in q1learner_persist.hpp): a whole
// update's SGD steps at the reference's minibatch size as ONE dispatch.  Its own translation unit: it is compiled with its own flags
// (q1physrl_amd/build.py TU_FLAGS: accumulators in the ordinary vector registers; no atomic optimizer - see there).
#include "q1env_host.hpp"
#include "q1policy.hpp"
#include "q1policy_glue.hpp"
#include "q1ppo_loss.hpp"
#include "q1learner_persist.hpp"
#include "q1learner_persist32.hpp"

using namespace q1;

namespace {
// the float16 loss scales of q1env_learner_step (q1env_learner.hip: the same defaults, the same setter)
static float learner_pi_upscale(const q1env* h) { return h->pi_upscale > 0.0f ? h->pi_upscale : 256.0f; }
static float learner_value_downscale(const q1env* h) { return h->value_downscale > 0.0f ? h->value_downscale : 1.0f; }
int check_nets(const char* who, const q1env_learner_net* pi, const q1env_learner_net* vf, bool need_grads) {
    for (const q1env_learner_net* m : {pi, vf}) {
        if (!m || !m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3) return fail(Q1ENV_ERR_INVALID_ARG, std::string(who) + ": null weight pointer");
        if (need_grads && (!m->gw1 || !m->gb1 || !m->gw2 || !m->gb2 || !m->gw3 || !m->gb3)) return fail(Q1ENV_ERR_INVALID_ARG, std::string(who) + ": null gradient pointer");
        if (m->out_dim < 1 || m->out_dim > 32) return fail(Q1ENV_ERR_INVALID_ARG, std::string(who) + ": out_dim must be in 1..32");
    }
    return 0;
}
}  // namespace

// ---- persistent learner (q1learner_persist.hpp): steps x { forward, loss gradient, backward, weight gradients, Adam } of 128-sample
// minibatches as ONE dispatch of 2 x 8 co-operating workgroups
namespace {
struct PWs { uint16_t* h1x; uint16_t* h1tx; uint16_t* dz2x; uint16_t* w2tx; float* yp; float* w2st; float* b3x; uint32_t* bar; size_t bytes; };
constexpr size_t STATUS_SNAP_OFF = 192;        // int64 inside the 256-byte status line: the step count at the start of the launch (q1pl::Args::step0_snap)
size_t carve_pws(void* base, int64_t batch_rows, PWs out[2], uint32_t** status, float** mouse_u) {
    char* b = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* q = b ? b + off : nullptr; off += align_up(bytes, 256); return q; };
    if (status) *status = (uint32_t*)take(256); else (void)take(256);
    { float* u = (float*)take((size_t)batch_rows * 4); if (mouse_u) *mouse_u = u; }
    for (int k = 0; k < 2; ++k) {
        PWs w{};
        const size_t start = off;
        w.bar = (uint32_t*)take(256);
        w.b3x = (float*)take(256);
        // (activations are exchanged as float16 by the default kernel and as float32 by q1env_learner_sgd_epochs_f32: sized for the latter)
        w.h1x = (uint16_t*)take((size_t)2 * q1pl32::XB_ACT);
        w.h1tx = (uint16_t*)take((size_t)2 * q1pl32::XB_ACT);
        w.dz2x = (uint16_t*)take((size_t)q1pl32::XB_ACT);
        w.w2tx = (uint16_t*)take((size_t)2 * q1pl::HID * q1pl::HID * 2);
        w.yp = (float*)take((size_t)q1pl::G * q1pl::MB * 16 * 4);
        w.w2st = (float*)take((size_t)q1pl::G * 3 * 4 * 2048 * 4);
        w.bytes = off - start;         // the group's exchange workspace = [bar, end of w2st): what its buffer resource addresses
        if (out) out[k] = w;
    }
    return off;
}
}  // namespace

extern "C" {

uint64_t q1env_learner_persistent_bytes(int64_t batch_rows) { return batch_rows > 0 ? (uint64_t)carve_pws(nullptr, batch_rows, nullptr, nullptr, nullptr) : 0; }

int q1env_learner_persistent_layout(int64_t batch_rows, int net, uint64_t* offsets9) {
    if (batch_rows <= 0 || net < 0 || net > 1 || !offsets9) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_persistent_layout: bad argument");
    PWs pw[2];
    uint32_t* status = nullptr;
    float* mouse_u = nullptr;
    char* const base = (char*)(uintptr_t)4096;                  // (any non-null base: only differences are reported)
    carve_pws(base, batch_rows, pw, &status, &mouse_u);
    const PWs& w = pw[net];
    const void* parts[8] = {w.bar, w.b3x, w.h1x, w.h1tx, w.dz2x, w.w2tx, w.yp, w.w2st};
    for (int k = 0; k < 8; ++k) offsets9[k] = (uint64_t)((const char*)parts[k] - base);
    offsets9[8] = (uint64_t)w.bytes;
    return Q1ENV_OK;
}

int q1env_learner_set_exchange_mode(q1env_t* h, int mode) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_set_exchange_mode: null handle");
    if (mode < 0 || mode > 2) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_set_exchange_mode: mode must be 0 (automatic), 1 (agent scope) or 2 (automatic, census made to fail)");
    h->plearner_mode = mode;
    return Q1ENV_OK;
}

int q1env_learner_set_profiling(q1env_t* h, int wave_of_group) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_set_profiling: null handle");
    if (wave_of_group < -1 || wave_of_group > 31) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_set_profiling: -1 (off) or workgroup (0..7) + 8 x wave (0..3)");
    h->plearner_prof = wave_of_group;
    return Q1ENV_OK;
}

}  // extern "C"

namespace {
int sgd_epochs_impl(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* pws_dev, const q1env_learner_batch* b,
                    int64_t batch_rows, int64_t idx_rows, int64_t steps, int64_t steps_per_epoch, int64_t epoch_stride, float lr, float beta1,
                    float beta2, float eps, void* adam_state_dev, double timeout_s, bool f32) {
    if (!h || !pws_dev || !b || !adam_state_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: null argument");
    if (!b->obs_dev || !b->old_logits_dev || !b->keys_dev || !b->mouse_dev || !b->logp_old_dev || !b->adv_dev || !b->value_old_dev || !b->vtarg_dev ||
        !b->kl_coeff_dev)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: null pointer in q1env_learner_batch");
    if (!(lr >= 0.0f) || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps > 0.0f))
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: bad hyper-parameter");
    if (int r = check_nets("q1env_learner_sgd_epochs", pi, vf, true)) return r;
    if (b->minibatch != q1pl::MB) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: the persistent learner is built for minibatches of 128 samples (RLlib's sgd_minibatch_size); use q1env_learner_sgd_step for other sizes");
    if (batch_rows < q1pl::MB) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: batch_rows must be the number of rows of the train batch arrays (>= 128)");
    if (steps <= 0 || steps_per_epoch <= 0 || epoch_stride < steps_per_epoch * q1pl::MB) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: bad step schedule");
    // the LAST position the schedule reads must exist: in idx_dev (idx_rows entries) or, without an index list, among the rows themselves
    {
        const int64_t last = ((steps - 1) / steps_per_epoch) * epoch_stride + ((steps - 1) % steps_per_epoch + 1) * q1pl::MB;
        const int64_t have = b->idx_dev ? idx_rows : batch_rows;
        if (b->idx_dev && idx_rows <= 0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: idx_rows must be the number of entries of idx_dev");
        if (last > have)
            return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: the schedule reads position " + std::to_string(last - 1) + " but " +
                                                   (b->idx_dev ? "idx_dev has " : "the train batch has ") + std::to_string(have) + " entries");
    }
    if (!(h->p.num_keys == 4 && h->p.yaw_mode == 1 && pi->out_dim == 10) || vf->out_dim != 1)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: written for the reference's action structure (4 keys + continuous mouse: 10 policy outputs, scalar value); use q1env_learner_sgd_step");
    if (b->old_stride < pi->out_dim) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: old_stride smaller than the policy row");
    if (h->num_cus < 2 * q1pl::G) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_sgd_epochs: needs 16 compute units");
    DeviceGuard guard(h->device);
    if (!h->plearner_attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)q1pl::persistent_learner_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1pl::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void*)q1pl::persistent_learner_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1pl::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void*)q1pl32::persistent_learner_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1pl32::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void*)q1pl32::persistent_learner_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1pl32::LDS_BYTES));
        h->plearner_attr_set = true;
    }
    PWs pw[2];
    uint32_t* status = nullptr;
    float* mouse_u = nullptr;
    carve_pws(pws_dev, batch_rows, pw, &status, &mouse_u);
    // status words and BOTH groups' arrival counters start at zero in every launch (a counter left at its previous final value would
    // let every wait of the new launch pass at once: no synchronisation at all - found the hard way, by a diverging training run)
    HIP_TRY(hipMemsetAsync(status, 0, 256, h->stream));
    HIP_TRY(hipMemsetAsync(pw[0].bar, 0, 256, h->stream));
    HIP_TRY(hipMemsetAsync(pw[1].bar, 0, 256, h->stream));
#ifdef Q1_CHECK
    {   // the assertion build's globals (q1learner_persist.hpp): where a failed assertion is reported, how far an exchange offset may reach
        const uint32_t xbytes = (uint32_t)pw[0].bytes;
        HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(q1pl::g_chk_status), &status, sizeof(status), 0, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(q1pl::g_chk_xbytes), &xbytes, sizeof(xbytes), 0, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));              // (the sources are locals)
    }
#endif
    char* st = (char*)adam_state_dev;
    const size_t per_pi = 65536u + 256u + 1536u + 256u + (size_t)pi->out_dim * 257u, per_vf = 65536u + 256u + 1536u + 256u + 257u;
    float* m_pi = (float*)(st + 256), *v_pi = m_pi + per_pi;
    float* m_vf = (float*)(st + 256 + align_up(2 * per_pi * 4u, 256)), *v_vf = m_vf + per_vf;
    q1pl::Args a{};
    a.p = h->p;
    const float mbf = (float)q1pl::MB;
    auto fill = [&](q1pl::Net& n, const q1env_learner_net* s, float* m, float* v, const PWs& w, float inv_b, float inv_scale) {
        n.w1 = const_cast<float*>(s->w1); n.b1 = const_cast<float*>(s->b1); n.w2 = const_cast<float*>(s->w2); n.b2 = const_cast<float*>(s->b2);
        n.w3 = const_cast<float*>(s->w3); n.b3 = const_cast<float*>(s->b3);
        n.gw1 = s->gw1; n.gb1 = s->gb1; n.gw2 = s->gw2; n.gb2 = s->gb2; n.gw3 = s->gw3; n.gb3 = s->gb3;
        n.m = m; n.v = v; n.out_dim = s->out_dim;
        n.h1x = w.h1x; n.h1tx = w.h1tx; n.dz2x = w.dz2x; n.w2tx = w.w2tx; n.yp = w.yp; n.w2st = w.w2st; n.b3x = w.b3x; n.bar = w.bar; n.xbase = (const char*)w.bar;
        n.inv_b = inv_b; n.inv_scale = inv_scale;
    };
    // the float16 loss scales of q1env_learner_step: per-sample gradients x pi_upscale (policy) / value_downscale (value); none in float32
    fill(a.net[0], pi, m_pi, v_pi, pw[0], f32 ? 1.0f : learner_pi_upscale(h), 1.0f / (mbf * (f32 ? 1.0f : learner_pi_upscale(h))));
    fill(a.net[1], vf, m_vf, v_vf, pw[1], f32 ? 1.0f : 1.0f / learner_value_downscale(h), (f32 ? 1.0f : learner_value_downscale(h)) / mbf);
    a.idx = b->idx_dev; a.spe = steps_per_epoch; a.epoch_stride = epoch_stride;
    a.rows = batch_rows; a.idx_rows = idx_rows;
    a.obs = b->obs_dev; a.old_logits = b->old_logits_dev; a.old_stride = b->old_stride;
    a.wide_old = (b->old_stride % 2 == 0 && ((uintptr_t)b->old_logits_dev & 7u) == 0) ? 1 : 0;
    a.keys = b->keys_dev; a.mouse_u = mouse_u; a.logp_old = b->logp_old_dev; a.adv = b->adv_dev; a.value_old = b->value_old_dev; a.vtarg = b->vtarg_dev;
    a.clip = b->clip_param; a.vf_clip = b->vf_clip_param; a.vf_coeff = b->vf_loss_coeff; a.ent_coeff = b->entropy_coeff;
    a.klc_dev = b->kl_coeff_dev;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.steps = steps;
    a.step_count = (long long*)st;
    a.step0_snap = (const long long*)((const char*)status + STATUS_SNAP_OFF);
    a.stats_acc = (float*)(st + 16);
    a.saturation = b->saturation_dev;
    a.status = status;
    a.allow_local = h->plearner_mode == 1 ? 0 : 1;
    a.census_skew = h->plearner_mode == 2 ? 1 : 0;
    a.prof = h->plearner_prof >= 0 ? reinterpret_cast<unsigned long long*>(status + 4) + 1 : nullptr;      // (bytes 24 .. 183 of the status line)
    a.prof_g = h->plearner_prof >= 0 ? h->plearner_prof : 0;
    a.timeout_ticks = (uint64_t)((timeout_s > 0 ? timeout_s : 5.0) * (h->wall_clock_hz > 0 ? h->wall_clock_hz : 1e8));
    // first on the stream: the mouse pre-images of all rows + the snapshot of the step count both groups start from
    hipLaunchKernelGGL(q1pl::mouse_u_kernel, dim3((unsigned)((batch_rows + 255) / 256)), dim3(256), 0, h->stream, batch_rows, b->mouse_dev, -h->p.action_range_f32,
                       h->p.action_range_f32, mouse_u, (const long long*)st, (long long*)((char*)status + STATUS_SNAP_OFF));
    if (f32 && a.prof) hipLaunchKernelGGL(q1pl32::persistent_learner_f32_kernel<true>, dim3(8 * q1pl::G), dim3(256), q1pl32::LDS_BYTES, h->stream, a);
    else if (f32) hipLaunchKernelGGL(q1pl32::persistent_learner_f32_kernel<false>, dim3(8 * q1pl::G), dim3(256), q1pl32::LDS_BYTES, h->stream, a);
    else if (a.prof) hipLaunchKernelGGL(q1pl::persistent_learner_kernel<true>, dim3(8 * q1pl::G), dim3(256), q1pl::LDS_BYTES, h->stream, a);
    else hipLaunchKernelGGL(q1pl::persistent_learner_kernel<false>, dim3(8 * q1pl::G), dim3(256), q1pl::LDS_BYTES, h->stream, a);
    HIP_TRY(hipGetLastError());
    return Q1ENV_OK;
}
}  // namespace

extern "C" {

int q1env_learner_sgd_epochs(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* pws_dev, const q1env_learner_batch* b,
                             int64_t batch_rows, int64_t idx_rows, int64_t steps, int64_t steps_per_epoch, int64_t epoch_stride, float lr, float beta1,
                             float beta2, float eps, void* adam_state_dev, double timeout_s) {
    return sgd_epochs_impl(h, pi, vf, pws_dev, b, batch_rows, idx_rows, steps, steps_per_epoch, epoch_stride, lr, beta1, beta2, eps, adam_state_dev, timeout_s, false);
}

int q1env_learner_sgd_epochs_f32(q1env_t* h, const q1env_learner_net* pi, const q1env_learner_net* vf, void* pws_dev, const q1env_learner_batch* b,
                                 int64_t batch_rows, int64_t idx_rows, int64_t steps, int64_t steps_per_epoch, int64_t epoch_stride, float lr, float beta1,
                                 float beta2, float eps, void* adam_state_dev, double timeout_s) {
    return sgd_epochs_impl(h, pi, vf, pws_dev, b, batch_rows, idx_rows, steps, steps_per_epoch, epoch_stride, lr, beta1, beta2, eps, adam_state_dev, timeout_s, true);
}

int q1env_learner_persistent_status(q1env_t* h, const void* pws_dev, uint32_t* status4) {
    if (!h || !pws_dev || !status4) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_persistent_status: null argument");
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(status4, pws_dev, 16, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

// out4 = {1 if this library was built with -DQ1_CHECK else 0 ... }: the persistent learner's assertion counters (q1learner_persist.hpp): exchange
// accesses / row indices / barrier readings checked, assertions failed - since the library was loaded.  Synchronises the stream.
int q1env_learner_debug_counters(q1env_t* h, uint64_t* out5) {
    if (!h || !out5) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_learner_debug_counters: null argument");
    for (int k = 0; k < 5; ++k) out5[k] = 0;
#ifdef Q1_CHECK
    DeviceGuard guard(h->device);
    HIP_TRY(hipStreamSynchronize(h->stream));
    unsigned long long c[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyFromSymbol(c, HIP_SYMBOL(q1pl::g_chk_counts), sizeof(c), 0, hipMemcpyDeviceToHost));
    out5[0] = 1;
    for (int k = 0; k < 4; ++k) out5[1 + k] = c[k];
#endif
    return Q1ENV_OK;
}

}  // extern "C"
