// q1env_diag.hip - measurement and self-test entry points of libq1env.so: the handle's timer events, the PMC traffic calibration
// kernel (known bytes in step_kernel's own access pattern) and the on-device check of the exact-division shortcuts.
#include "q1env_host.hpp"

using namespace q1;

// Traffic calibration for the PMC counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE must be calibrated on
// a known byte count in the kernel's own access pattern): reads every SoA state array with exactly the loads
// step_kernel uses and writes the same bytes to a scratch arena: 85 B read + 85 B written per env, no arithmetic.
__global__ void __launch_bounds__(256) calib_copy_kernel(Params p, StatePtrs src, StatePtrs dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)p.n) return;
    Env e;
    load_env(src, (uint32_t)p.n, i, e);
    store_env(dst, (uint32_t)p.n, i, e);
}

// Self-test of the exact-division helpers against the hardware IEEE division on random operands drawn over the
// ranges the hot path produces (and well beyond).  counts[0..3] = mismatches of: div_const<double>, div_shared,
// the float32 vel-obs column, the float32 z-obs column.
__global__ void __launch_bounds__(256)
selftest_division_kernel(uint64_t n, uint64_t seed, double c_extra0, double c_extra1, unsigned long long* counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r[4], r2[4];
    philox_draw(seed, i, 0, 7, 0, r);
    philox_draw(seed, i, 1, 7, 0, r2);
    const double u = u53(r[0], r[1]), w = u53(r[2], r[3]);
    // magnitude sweep 1e-12 .. 1e7, both signs
    const double mag = exp(-27.6 + 43.7 * w);
    const double x = (2.0 * u - 1.0) * mag;
    const double cs[6] = {180.0, 90.0, 100.0, 200.0, c_extra0, c_extra1};
    unsigned bad0 = 0, bad1 = 0, bad2 = 0, bad3 = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double c = cs[k];
        const double a = div_const<double>(x, c, 1.0 / c), b = x / c;
        bad0 += (__double_as_longlong(a) != __double_as_longlong(b));
    }
    const double den = 0.5 + 4000.0 * u53(r2[0], r2[1]);
    const double num = (2.0 * u53(r2[2], r2[3]) - 1.0) * den;
    {
        const double y = rcp_refined(den);
        const double a = div_shared(num, den, y), b = num / den;
        bad1 += (__double_as_longlong(a) != __double_as_longlong(b));
        const double a2 = div_shared(x, den, y), b2 = x / den;
        bad1 += (__double_as_longlong(a2) != __double_as_longlong(b2));
        // friction quotient (phys.py:88-90): new_speed / speed with a float32 speed and 0 <= new_speed <= speed
        const float spf = (float)(0.001 + 3000.0 * u);
        const double ns = fmax(0.0, (double)spf - w * 60.0);
        const double a3 = div_shared(ns, (double)spf, rcp_refined((double)spf)), b3 = ns / (double)spf;
        bad1 += (__double_as_longlong(a3) != __double_as_longlong(b3));
    }
    {   // vel column: v float32 -> trunc(v/16)*16 / 200, float64 reference vs float32 shortcut
        const float v = (float)((2.0 * u - 1.0) * 40000.0);
        const double ref = (trunc((double)(v / 16.0f)) * 16.0 + 0.0) / 200.0;
        const float fast = div_const<float>(truncf(v * 0.0625f) * 16.0f + 0.0f, 200.0f, 1.0f / 200.0f);
        bad2 += (__float_as_uint((float)ref) != __float_as_uint(fast));
        const double z = 24.03125 + 3000.0 * w;
        const double refz = (rint(z * 8.0) / 8.0) / 100.0;
        const float fastz = div_const<float>((float)(rint(z * 8.0) * 0.125), 100.0f, 1.0f / 100.0f);
        bad3 += (__float_as_uint((float)refz) != __float_as_uint(fastz));
    }
    if (bad0) atomicAdd(&counts[0], (unsigned long long)bad0);
    if (bad1) atomicAdd(&counts[1], (unsigned long long)bad1);
    if (bad2) atomicAdd(&counts[2], (unsigned long long)bad2);
    if (bad3) atomicAdd(&counts[3], (unsigned long long)bad3);
}

extern "C" {

int q1env_selftest_division(int device, uint64_t n, uint64_t seed, double c0, double c1, uint64_t* mismatches4) {
    if (!mismatches4 || n == 0 || !(c0 > 0) || !(c1 > 0)) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_selftest_division: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(Q1ENV_ERR_NO_DEVICE, "q1env_selftest_division: no HIP device visible");
    if (device < 0 || device >= ndev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_selftest_division: bad device index");
    DeviceGuard guard(device);
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d, 0, 4 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(selftest_division_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, seed, c0, c1, d);
    unsigned long long hcounts[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpy(hcounts, d, sizeof(hcounts), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(Q1ENV_ERR_HIP, std::string("selftest: ") + hipGetErrorString(e));
    for (int k = 0; k < 4; ++k) mismatches4[k] = hcounts[k];
    return Q1ENV_OK;
}

int q1env_calibrate_traffic(q1env_t* h, int launches) {
    if (!h || launches <= 0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_calibrate_traffic: bad argument");
    DeviceGuard guard(h->device);
    if (int r = ensure_stage(h, arena_bytes((size_t)h->p.n))) return r;
    StatePtrs dst{};
    carve_into(h->stage, (size_t)h->p.n, dst);
    const int blk = block_for(h->p.n);
    for (int l = 0; l < launches; ++l)
        hipLaunchKernelGGL(calib_copy_kernel, grid_for(h->p.n, blk), dim3(blk), 0, h->stream, h->p, h->st, dst);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return Q1ENV_OK;
}

int q1env_timer_start(q1env_t* h) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_timer_start: null handle");
    DeviceGuard guard(h->device);
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return Q1ENV_OK;
}

int q1env_timer_mark(q1env_t* h) {
    if (!h) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_timer_mark: null handle");
    DeviceGuard guard(h->device);
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    return Q1ENV_OK;
}

int q1env_timer_elapsed(q1env_t* h, float* ms) {
    if (!h || !ms) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_timer_elapsed: null argument");
    DeviceGuard guard(h->device);
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return Q1ENV_OK;
}

int q1env_timer_stop(q1env_t* h, float* ms) {
    if (!h || !ms) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_timer_stop: null argument");
    DeviceGuard guard(h->device);
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return Q1ENV_OK;
}

}  // extern "C"
