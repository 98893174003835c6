// q1env_resident.hip - the resident sampler of libq1env.so (q1env_sample_resident; device code in q1resident.hpp).
#include "q1env_host.hpp"
#include "q1policy.hpp"
#include "q1policy_glue.hpp"
#include "q1resident.hpp"

using namespace q1;

extern "C" {

// ---- resident sampler ----------------------------------------------------------------------------------------------------
int q1env_sample_resident(q1env_t* h, const q1env_resident_args* a) {
    if (!h || !a || !a->pi) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: null argument");
    const q1env_mlp* m = a->pi;
    if (!m->w1 || !m->b1 || !m->w23_image || !m->b2 || !m->b3) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: null pointer in q1env_mlp");
    if (!a->keys_dev || !a->logp_dev || !a->obs_dev || !a->reward_dev || !a->done_dev || !a->ep_return_dev || !a->partials_dev || !a->status_dev)
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: null trajectory / status pointer");
    if (a->ticks <= 0 || !(a->timeout_s > 0.0) || a->timeout_s > 30.0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: bad ticks / timeout_s");
    const int width = policy_row_width(h->p);
    if (m->out_dim != width) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: pi->out_dim must be " + std::to_string(width));
    if (width > 24) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: more than 24 policy outputs do not fit the resident workgroup's LDS (use q1env_sample_step)");
    if (h->p.yaw_mode == 2 && !m->out) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: a discrete-mouse policy needs the logits trajectory (pi->out)");
    if (h->p.yaw_mode != 0 && !a->mouse_dev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_sample_resident: mouse trajectory required");
    DeviceGuard guard(h->device);
    if (!h->resident_attr_set) {
#define Q1_ATTR(SP, TP, R3) HIP_TRY(hipFuncSetAttribute((const void*)sampler_resident_kernel<SP, TP, R3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q1res::Map<R3>::bytes(TP)))
        Q1_ATTR(true, 1, 16); Q1_ATTR(true, 2, 16); Q1_ATTR(true, 1, 24);
        Q1_ATTR(false, 1, 16); Q1_ATTR(false, 2, 16); Q1_ATTR(false, 1, 24);
#undef Q1_ATTR
        h->resident_attr_set = true;
    }
    // a workgroup fills a CU (its LDS holds the network) and serves 128 envs at one tile per policy wave, 256 at two (heads of up to 10
    // outputs only: the 24-row W3 tile of a discrete-mouse head leaves no room for the second tile's hand-off area).  Workgroups are
    // self-contained - every wait is on a tag in the workgroup's own LDS - so a grid larger than the device is legal: the dispatcher
    // starts the next workgroup (its whole horizon) on a CU when one retires.  One tile per wave while that grid is co-resident
    // (lowest latency per tick), two above (half as many workgroups, half as many weight stagings).
    const unsigned n = (unsigned)h->p.n, cus = (unsigned)h->num_cus;
    const bool wide = h->p.yaw_mode == 2 || width > 10;   // a discrete-mouse head: the 24-row variant (gathers all of an env's logits, writes the row before it samples)
    int tp = (wide || (n + 127u) / 128u <= cus) ? 1 : 2;
    if (const char* f = getenv("Q1ENV_RESIDENT_TP")) { if (f[0] == '2' && !wide) tp = 2; else if (f[0] == '1') tp = 1; }      // measurement knob
    ResidentArgs k{};
    k.ticks = a->ticks;
    k.pi = q1pol::Net{m->w1, m->b1, m->w23_image, m->b2, m->b3, m->out, m->out_dim};
    k.seed = a->seed; k.counter_offset = a->counter_offset + (a->counter_dev ? 0 : h->tick_count); k.counter_dev = a->counter_dev;
    k.deterministic = a->deterministic;
    k.keys = a->keys_dev; k.mouse = a->mouse_dev; k.logp = a->logp_dev; k.obs = a->obs_dev; k.reward = a->reward_dev; k.done = a->done_dev;
    k.zero_start = a->zero_start_dev; k.ep_return = a->ep_return_dev; k.partials = a->partials_dev;
    k.status = a->status_dev;
    k.timeout_ticks = (uint64_t)(a->timeout_s * 1.0e8);
    const unsigned per_block = 128u * (unsigned)tp;
    const dim3 g((n + per_block - 1u) / per_block), b(512);
#define Q1_LAUNCH_RS(SP, TP, R3) hipLaunchKernelGGL((sampler_resident_kernel<SP, TP, R3>), g, b, q1res::Map<R3>::bytes(TP), h->stream, h->p, h->st, k)
    if (is_spec(h->p)) { if (wide) Q1_LAUNCH_RS(true, 1, 24); else if (tp == 1) Q1_LAUNCH_RS(true, 1, 16); else Q1_LAUNCH_RS(true, 2, 16); }
    else { if (wide) Q1_LAUNCH_RS(false, 1, 24); else if (tp == 1) Q1_LAUNCH_RS(false, 1, 16); else Q1_LAUNCH_RS(false, 2, 16); }
#undef Q1_LAUNCH_RS
    HIP_TRY(hipGetLastError());
    h->tick_count += (uint64_t)a->ticks;
    return Q1ENV_OK;
}

}  // extern "C"
