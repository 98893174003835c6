// q1env_diag.hip - measurement and self-test entry points of libq1env.so: the handle's timer events, the PMC traffic calibration
// kernel (known bytes in step_kernel's own access pattern) and the on-device check of the exact-division shortcuts.
#include "q1env_host.hpp"

#include <time.h>

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
        // the ONE-step form, wherever the host-side bound allows it (the same test q1env_host.hpp div_one_step_ok applies)
        if (fabs(fma(c, 1.0 / c, -1.0)) <= 0x1p-54) {
            const double a1 = div_const1<double>(x, c, 1.0 / c);
            bad0 += (__double_as_longlong(a1) != __double_as_longlong(b));
        }
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
        const float fast = div_const1<float>(truncf(v * 0.0625f), 12.5f, 0.08f);                  // observe<float>'s vel columns
        bad2 += (__float_as_uint((float)ref) != __float_as_uint(fast));
        const double z = 24.03125 + 3000.0 * w;
        const double refz = (rint(z * 8.0) / 8.0) / 100.0;
        const float fastz = (float)(rint(z * 8.0) * (1.0 / 800.0));                                // observe<float>'s z column
        bad3 += (__float_as_uint((float)refz) != __float_as_uint(fastz));
        // EXHAUSTIVE part: thread i < 2^25 checks the integer m = i - 2^24 as trunc(v / 16) and |m| as rint(8 z) - every numerator the
        // two shortcuts are claimed for (|m|, j < 2^24), against the reference's float64 expressions rounded to float32
        if (i < (1ull << 25)) {
            const double md = (double)((long long)i - (1ll << 24));
            const float mf = (float)md;
            const double refm = (md * 16.0 + 0.0) / 200.0;
            bad2 += (__float_as_uint((float)refm) != __float_as_uint(div_const1<float>(mf, 12.5f, 0.08f)));
            const double jd = fabs(md);
            const double refj = (jd * 0.125) / 100.0;
            bad3 += (__float_as_uint((float)refj) != __float_as_uint((float)(jd * (1.0 / 800.0))));
        }
    }
    if (bad0) atomicAdd(&counts[0], (unsigned long long)bad0);
    if (bad1) atomicAdd(&counts[1], (unsigned long long)bad1);
    if (bad2) atomicAdd(&counts[2], (unsigned long long)bad2);
    if (bad3) atomicAdd(&counts[3], (unsigned long long)bad3);
}

// The reader of tests/test_hip_signal.py's visibility check: launched on ANOTHER stream BEFORE a signalled q1env_rollout, it polls the
// launch's sequence number (its device-resident copy, written just before the word the host polls); the moment it is there it reads the launch's tick-major outputs with
// system-scope loads (what any other agent effectively does: nothing served from this XCD's own L2) and counts the 8-byte words that
// differ from `expect` - NEWEST TICK FIRST (done, reward, obs of tick T-1, then T-2, ...): the last ticks' results are the ones that
// could still sit dirty in another XCD's L2 if the signal were published without visibility, and this reader - on the same device,
// a microsecond or two behind the signal, its workgroups spread over all eight XCDs - is the consumer most likely to catch it.
// Arena layout (both out and expect): obs f32 [T][n][6] at 0, reward f32 [T][n] at off_rew, done u8 [T][n] at off_done; n % 8 == 0.
// result[0] = differing words, result[1] = 1 if the sequence number did not arrive within ~timeout_ticks of the wall clock,
// result[2] = wall clock when workgroup 0 saw the signal, result[3] = wall clock when workgroup 0 finished the newest tick.
__global__ void __launch_bounds__(256)
signal_reader_kernel(uint64_t* sig, const uint64_t* seq_dev, uint64_t want_seq, const char* out, const char* expect, uint64_t off_rew, uint64_t off_done,
                     uint32_t n, uint32_t ticks, uint64_t timeout_ticks, unsigned long long* result) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const uint64_t t0 = wall_clock64();
        int seen = 0;
        // "armed": the host side of q1env_diag_signal_reader returns only when workgroup 0 is polling (sig[3] is a spare word of the block)
        if (blockIdx.x == 0) __hip_atomic_store(sig + 3, want_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (;;) {
            // the DEVICE-resident copy of the sequence number, which the signalling wave writes (write-through) just before the host's
            if (__hip_atomic_load(seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= want_seq) { seen = 1; break; }
            if (wall_clock64() - t0 > timeout_ticks) break;
            __builtin_amdgcn_s_sleep(1);
        }
        ok = seen;
        if (blockIdx.x == 0) result[2] = seen ? (unsigned long long)wall_clock64() : 0ull;
    }
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) atomicMax(result + 1, 1ull); return; }
    unsigned long long bad = 0;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (int t = (int)ticks - 1; t >= 0; --t) {
        const uint64_t base[3] = {off_done + (uint64_t)t * n, off_rew + (uint64_t)t * n * 4u, (uint64_t)t * n * 24u};
        const uint64_t words[3] = {n / 8u, n / 2u, n * 3ull};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint64_t* o = reinterpret_cast<const uint64_t*>(out + base[a]);
            const uint64_t* e = reinterpret_cast<const uint64_t*>(expect + base[a]);
            for (uint64_t j = tid; j < words[a]; j += stride)
                bad += __hip_atomic_load(o + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != e[j] ? 1ull : 0ull;
        }
        if (t == (int)ticks - 1 && tid == 0) result[3] = (unsigned long long)wall_clock64();
    }
    if (bad) atomicAdd(result, bad);
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

__global__ void __launch_bounds__(256)
selftest_trig_kernel(uint64_t n, const double* yaw_deg, double* sin_out, double* cos_out, uint64_t seed, unsigned long long* counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double rad = div_const1<double>(yaw_deg[i] * 3.141592653589793, 180.0, 1.0 / 180.0);   // physics_yaw_only's radians
    double sn, cs, sl, cl;
    sincos_yaw(tick_consts(), rad, sn, cs);
    sincos(rad, &sl, &cl);
    sin_out[i] = sn;
    cos_out[i] = cs;
    const long long ds = llabs(__double_as_longlong(sn) - __double_as_longlong(sl));             // (same sign and binade up to an ulp)
    const long long dc = llabs(__double_as_longlong(cs) - __double_as_longlong(cl));
    if (ds) { atomicAdd(&counts[0], 1ull); atomicMax(&counts[1], (unsigned long long)ds); }
    if (dc) { atomicAdd(&counts[0], 1ull); atomicMax(&counts[1], (unsigned long long)dc); }
    uint32_t r[4];
    philox_draw(seed, i, 0, 9, 0, r);
    const double a = ldexp(1.0 + u53(r[0], r[1]), (int)(r[2] % 1400u) - 700);                  // [2^-700, 2^700)
    if (__double_as_longlong(sqrt_normal(a)) != __double_as_longlong(sqrt(a))) atomicAdd(&counts[2], 1ull);
}

int q1env_selftest_trig(int device, uint64_t n, const double* yaw_deg, double* sin_out, double* cos_out, uint64_t seed, uint64_t* counts4) {
    if (!yaw_deg || !sin_out || !cos_out || !counts4 || n == 0) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_selftest_trig: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(Q1ENV_ERR_NO_DEVICE, "q1env_selftest_trig: no HIP device visible");
    if (device < 0 || device >= ndev) return fail(Q1ENV_ERR_INVALID_ARG, "q1env_selftest_trig: bad device index");
    DeviceGuard guard(device);
    double* d = nullptr;
    unsigned long long* c = nullptr;
    const size_t bytes = (size_t)n * sizeof(double);
    HIP_TRY(hipMalloc((void**)&d, 3 * bytes));
    if (hipMalloc((void**)&c, 4 * sizeof(unsigned long long)) != hipSuccess) { (void)hipFree(d); return fail(Q1ENV_ERR_HIP, "selftest_trig: hipMalloc"); }
    hipError_t e = hipMemcpy(d, yaw_deg, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(c, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(selftest_trig_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, d, d + n, d + 2 * n, seed, c);
        e = hipGetLastError();
    }
    unsigned long long hc[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(sin_out, d + n, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(cos_out, d + 2 * n, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    (void)hipFree(c);
    if (e != hipSuccess) return fail(Q1ENV_ERR_HIP, std::string("selftest_trig: ") + hipGetErrorString(e));
    for (int k = 0; k < 4; ++k) counts4[k] = hc[k];
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

// Diagnostics for the completion signal's visibility guarantee (include/q1env.h): enqueue signal_reader_kernel on `reader_stream` for the
// NEXT signalled launch of this handle.  out / expect: device arenas laid out obs | reward | done (see the kernel); result_dev: four
// 64-bit words, zeroed by the caller.  The caller then makes the signalled launch and synchronises reader_stream.
int q1env_diag_signal_reader(q1env_t* h, void* reader_stream, const void* out_dev, const void* expect_dev, uint64_t off_reward,
                             uint64_t off_done, int ticks, int workgroups, double timeout_s, uint64_t* result_dev) {
    if (!h || !out_dev || !expect_dev || !result_dev || ticks <= 0 || workgroups <= 0 || (h->p.n & 7) || (off_reward & 7u) || (off_done & 7u))
        return fail(Q1ENV_ERR_INVALID_ARG, "q1env_diag_signal_reader: bad argument (num_envs and the offsets must be multiples of 8)");
    DeviceGuard guard(h->device);
    if (int r = ensure_signal(h)) return r;
    const uint64_t tmo = (uint64_t)(timeout_s * (h->wall_clock_hz > 0 ? h->wall_clock_hz : 1e8));
    hipLaunchKernelGGL(signal_reader_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)reader_stream, h->sig_dev,
                       reinterpret_cast<const uint64_t*>(h->ticket_dev + SIGNAL_LEAVES * SIGNAL_LEAF_STRIDE + SIGNAL_SEQ_DEV_OFFSET), h->sig_seq + 1,
                       (const char*)out_dev, (const char*)expect_dev, off_reward, off_done, (uint32_t)h->p.n, (uint32_t)ticks, tmo,
                       (unsigned long long*)result_dev);
    HIP_TRY(hipGetLastError());
    // return when the reader is resident and polling (a kernel launched on a fresh stream can take > 100 us to start: the rollout that
    // follows would otherwise be over before its reader runs)
    const uint64_t want = h->sig_seq + 1;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (__atomic_load_n(const_cast<const uint64_t*>(h->sig_host + 3), __ATOMIC_ACQUIRE) != want) {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s)
            return fail(Q1ENV_ERR_HIP, "q1env_diag_signal_reader: the reader kernel did not start within the timeout");
    }
    return Q1ENV_OK;
}

}  // extern "C"
