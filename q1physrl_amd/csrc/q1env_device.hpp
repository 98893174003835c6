// q1env_device.hpp - per-env arithmetic of the q1physrl hot path, written for gfx950 (CDNA4).
//
// One env = one lane.  All per-tick constants live in a `Params` struct passed BY VALUE as a kernel
// argument, so they arrive through the scalar data path (s_load -> SGPRs) and cost no VGPRs, no LDS
// and no per-lane loads.  Every branch on `Params` is wave-uniform.
//
// Two instantiations of everything below:
//   SPEC = false  any Config (runtime num_keys / yaw mode / jump mode / hover / speed_reward), wave-uniform branches
//   SPEC = true   the default action/episode structure baked in at compile time (4 keys, continuous mouse,
//                 jump key, no hover, y-velocity reward: Config.get_default() and data/params.yml; move maxima of
//                 ordinary magnitude) -> no per-tick branches on the Config.  What a tick costs a lone wave on its
//                 SIMD is its instruction COUNT (one issue slot of ~5.2 cycles each, vector, scalar or memory;
//                 a dependent float64 chain issues as fast as independent ones; a branch ~ four slots):
//                 tools/ubench_f64.hip, tools/ubench_select.hip, DESIGN.md section 6.2.
//
// Numerics contract (restated from the reference as executed by NumPy 2.2.6):
//   * yaw, time_remaining, last_key_press_time, z_pos and all horizontal intermediates: float64
//   * vel: float32 storage, round-to-nearest-even on store (phys.py:190)
//   * float32 islands: friction speed/control (phys.py:85-86), the +270 jump add (phys.py:119),
//     the reward multiply (env.py:500-503)
//   * no FMA contraction (this file MUST be compiled with -ffp-contract=off), true division,
//     correctly rounded sqrt; einsum sums start from +0.0 (sign of zero results)
//   * sin/cos: float64, <= 1 ulp from the host libm's (sincos_yaw below; the device library's sincos beyond 2^20 radians and in the
//     reset / phys.apply paths); the float32 rounding of vel absorbs the difference, see DESIGN.md
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace q1 {

constexpr uint32_t FLAG_ON_GROUND = 1u << 0;
constexpr uint32_t FLAG_JUMP_RELEASED = 1u << 1;
constexpr uint32_t FLAG_ZERO_START = 1u << 2;
constexpr int FLAG_KEYS_SHIFT = 3;

constexpr uint32_t STREAM_ACTION = 1;   // Philox stream tags
constexpr uint32_t STREAM_RESET = 2;

// action layouts (include/q1env.h Q1ENV_ACT_*); FMT_RUNTIME = decided by a kernel argument
constexpr int FMT_RUNTIME = -1, FMT_F64_ROWS = 0, FMT_F32_ROWS = 1, FMT_PACKED = 2, FMT_RANDOM = 3;

struct Params {
    int32_t n;
    int32_t num_keys;       // 4, or 3 when auto_jump || !allow_jump (env.py:206-207)
    int32_t act_width;      // num_keys + (yaw_mode != 0)
    int32_t yaw_mode;       // 0 none, 1 continuous, 2 discrete (env.py:233-238)
    int32_t jump_mode;      // 0 never, 1 jump key, 2 auto-jump (env.py:262-267)
    int32_t smooth_keys;
    int32_t hover;
    int32_t speed_reward;
    double dt;              // time_delta
    double time_limit;
    double key_press_delay;
    double yaw_num;         // double(float32(720) * float32(dt))        env.py:230 under NEP 50
    double yaw_den;         // double(action_range) or discrete_yaw_steps env.py:236/238
    double yaw_steps;
    double yaw_den_rcp;     // RN(1 / yaw_den), for the exact constant division below
    double time_limit_rcp;  // RN(1 / time_limit)
    double fmove_max;       // double(float32(fmove_max))                 env.py:261
    double smove_max;       // double(float32(smove_max))                 env.py:260
    double accel_dt;        // double(float32(10)) * dt                   phys.py:78
    double grav_dt;         // double(float32(800)) * dt                  phys.py:122
    double smooth_prev;     // 1.0 if smooth_keys else 0.0   } level = (key + smooth_prev*prev) * smooth_scale
    double smooth_scale;    // 0.5 if smooth_keys else 1.0   } (exact: operands are 0/1)   env.py:251-254
    double zero_start_prob;
    double yaw_lo, yaw_hi;
    double max_initial_speed;
    double action_range;
    float dt_f32;           // float32(dt)                                env.py:501/503
    float action_range_f32;
    float log_range_f32;    // logf(2 * action_range_f32): log(high - low) of the mouse Box (policy-side kernels)
    int64_t env_index_base;
};

template <bool SPEC> __device__ __forceinline__ int cfg_num_keys(const Params& p) { if constexpr (SPEC) return 4; else return p.num_keys; }
template <bool SPEC> __device__ __forceinline__ int cfg_act_width(const Params& p) { if constexpr (SPEC) return 5; else return p.act_width; }
template <bool SPEC> __device__ __forceinline__ int cfg_yaw_mode(const Params& p) { if constexpr (SPEC) return 1; else return p.yaw_mode; }
template <bool SPEC> __device__ __forceinline__ int cfg_jump_mode(const Params& p) { if constexpr (SPEC) return 1; else return p.jump_mode; }
template <bool SPEC> __device__ __forceinline__ bool cfg_hover(const Params& p) { if constexpr (SPEC) return false; else return p.hover != 0; }
template <bool SPEC> __device__ __forceinline__ bool cfg_speed_reward(const Params& p) { if constexpr (SPEC) return false; else return p.speed_reward != 0; }

// SoA state in HBM.  Every array is num_envs long and 256-B aligned; lane i of a wave touches element
// base+i of each array, so every load/store instruction of a wave is one contiguous, aligned segment.
struct StatePtrs {
    float* vx; float* vy; float* vz;
    double* px; double* py; double* z;
    double* yaw; double* trem;
    double* lk;            // [4][n] key-major
    uint8_t* flags;
};

struct Env {               // one env's state in registers
    float vx, vy, vz;
    double px, py, z, yaw, trem;
    double lk[4];
    uint32_t flags;
};

struct Cmd {               // ActionDecoder.map's outputs for one env (env.py:269)
    double fmove, smove;
    uint32_t jump;         // 0 / 1
};

template <typename OBS_T>
struct TickOut {
    OBS_T obs[6];
    float reward;
    bool done;
};

// ---------------------------------------------------------------------------------------- float64 selects
// `c ? a : b` on float64 compiles to v_cmp -> VCC and two VOP2 v_cndmask_b32 ..., vcc.  On gfx950 the second of two back-to-back
// VOP2 v_cndmask_b32 on a freshly written VCC stalls for ~13 ns when all four SIMDs of the CU are busy: 20 ns per select against
// 8-10 ns with the mask in an SGPR pair and the VOP3 encoding (tools/ubench_select.hip, profiles/r3_ubench_select.txt) - and the tick
// had twelve.  Most became v_max / v_min (see physics_core); the rest go through these helpers.  mask = __ballot(condition), which
// folds into the v_cmp that computes the condition (combine conditions as masks, with &, not as booleans).  s_nop 1: the two wait states between a VALU write of an SGPR and a VALU read of
// it, which the compiler cannot see into the assembly to provide.
// In-place forms (the value that is kept where the mask is clear is the in/out operand: no copies around the assembly; early-clobber,
// because an input that happens to hold the same value would otherwise be given the same register and be overwritten mid-sequence).
__device__ __forceinline__ void select2_into_f64(uint64_t mask, double a0, double& r0, double a1, double& r1) {
    uint32_t lo0 = (uint32_t)__double2loint(r0), hi0 = (uint32_t)__double2hiint(r0);
    uint32_t lo1 = (uint32_t)__double2loint(r1), hi1 = (uint32_t)__double2hiint(r1);
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %0, %4, %8\n\tv_cndmask_b32_e64 %1, %1, %5, %8\n\t"
        "v_cndmask_b32_e64 %2, %2, %6, %8\n\tv_cndmask_b32_e64 %3, %3, %7, %8"
        : "+&v"(lo0), "+&v"(hi0), "+&v"(lo1), "+&v"(hi1)
        : "v"(__double2loint(a0)), "v"(__double2hiint(a0)), "v"(__double2loint(a1)), "v"(__double2hiint(a1)), "s"(mask));
    r0 = __hiloint2double((int)hi0, (int)lo0);
    r1 = __hiloint2double((int)hi1, (int)lo1);
}
// out-of-place pair: r0 = mask ? a0 : b0, r1 = mask ? a1 : b1 (no copies of the kept operands first; early-clobber outputs, because the
// sequence writes its first result while later instructions still read inputs)
__device__ __forceinline__ void select2_f64(uint64_t mask, double a0, double b0, double& r0, double a1, double b1, double& r1) {
    uint32_t lo0, hi0, lo1, hi1;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %4, %8, %12\n\tv_cndmask_b32_e64 %1, %5, %9, %12\n\t"
        "v_cndmask_b32_e64 %2, %6, %10, %12\n\tv_cndmask_b32_e64 %3, %7, %11, %12"
        : "=&v"(lo0), "=&v"(hi0), "=&v"(lo1), "=&v"(hi1)
        : "v"(__double2loint(b0)), "v"(__double2hiint(b0)), "v"(__double2loint(b1)), "v"(__double2hiint(b1)),
          "v"(__double2loint(a0)), "v"(__double2hiint(a0)), "v"(__double2loint(a1)), "v"(__double2hiint(a1)), "s"(mask));
    r0 = __hiloint2double((int)hi0, (int)lo0);
    r1 = __hiloint2double((int)hi1, (int)lo1);
}
// v_max_f64 / v_min_f64 on operands the compiler cannot see through (the pinned constants of TickConsts are opaque register values, so
// its own fmax / fmin first canonicalises them - a v_max_f64 x, x per use): the instruction itself, which under the kernels' IEEE mode
// already returns the non-NaN operand and quiets a signalling one.
__device__ __forceinline__ double max_f64_raw(double a, double b) {
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double min_f64_raw(double a, double b) {
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// (m & a) | (~m & b) and the sign-extended one-bit field, as the instructions themselves: the compiler turns the C forms back into
// v_cmp + two VOP2 v_cndmask on VCC - the pair that stalls (tools/ubench_select.hip)
__device__ __forceinline__ uint32_t bfi_b32(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b));
    return d;
}
template <int BIT>
__device__ __forceinline__ uint32_t bit_mask_i32(uint32_t x) {   // all ones if bit BIT of x is set, else zero (the offset as an inline constant: a register operand cost a v_mov per use)
    uint32_t d;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(d) : "v"(x), "n"(BIT));
    return d;
}
// four values, four masks, one replacement (the decoder's key-press timestamps): r[k] = m[k] ? a : r[k]
__device__ __forceinline__ void select4_into_f64(uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3, double a, double r[4]) {
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lo[k] = (uint32_t)__double2loint(r[k]); hi[k] = (uint32_t)__double2hiint(r[k]); }
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %0, %8, %10\n\tv_cndmask_b32_e64 %1, %1, %9, %10\n\t"
        "v_cndmask_b32_e64 %2, %2, %8, %11\n\tv_cndmask_b32_e64 %3, %3, %9, %11\n\t"
        "v_cndmask_b32_e64 %4, %4, %8, %12\n\tv_cndmask_b32_e64 %5, %5, %9, %12\n\t"
        "v_cndmask_b32_e64 %6, %6, %8, %13\n\tv_cndmask_b32_e64 %7, %7, %9, %13"
        : "+&v"(lo[0]), "+&v"(hi[0]), "+&v"(lo[1]), "+&v"(hi[1]), "+&v"(lo[2]), "+&v"(hi[2]), "+&v"(lo[3]), "+&v"(hi[3])
        : "v"(__double2loint(a)), "v"(__double2hiint(a)), "s"(m0), "s"(m1), "s"(m2), "s"(m3));
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __hiloint2double((int)hi[k], (int)lo[k]);
}

// ---------------------------------------------------------------------------------------- exact division
// x / c for a run-time CONSTANT c > 0 whose correctly rounded reciprocal y = RN(1/c) was computed once on the
// host.  q0 = RN(x*y) is within 1.5 ulp of x/c; one FMA correction step (r = x - q*c exactly, q' = RN(q + r*y))
// makes it faithful, and by Markstein's theorem a second identical step yields RN(x/c) - bit-identical to the
// IEEE division the reference performs, for 2^-960 < |x| < 2^960 (every quantity on this path), at 5 FMA-rate
// instructions instead of the ~11-instruction v_div_scale/v_rcp/v_div_fmas/v_div_fixup sequence.
// x = +0 gives +0 and NaN propagates; x = -0 would give +0 instead of -0, which cannot occur here: every dividend on
// this path is a sum that starts from +0.0, a difference of distinct times, or an angle that is never -0 (initial yaw
// is 90 or a uniform draw).  Checked on-device against `/` by q1env_selftest_division.
template <typename T> __device__ __forceinline__ T fma_t(T a, T b, T c);
template <> __device__ __forceinline__ double fma_t<double>(double a, double b, double c) { return fma(a, b, c); }
template <> __device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return fmaf(a, b, c); }

template <typename T>
__device__ __forceinline__ T div_const(T x, T c, T y) {
    T q = x * y;
    T r = fma_t<T>(-q, c, x);
    q = fma_t<T>(r, y, q);
    r = fma_t<T>(-q, c, x);
    q = fma_t<T>(r, y, q);
    return q;
}

// ONE correction step: q0 = RN(x y) is already FAITHFUL (one of the two neighbours of x / c) when the reciprocal's relative error
// e = c y - 1 satisfies |e| <= 2^-54: |x y - x/c| = |x/c| |e| < ulp(x/c) / 2 (a significand is < 2), plus half an ulp of rounding, is
// < 1 ulp.  Markstein's theorem (y = RN(1/c), q faithful, r = x - q c exact => RN(q + r y) = RN(x / c)) then needs no second step:
// 3 instructions.  |e| 2^54 is 0.6875 for 180 and 90, 1.0 for 10, 0.681 for float32(10.08), 0.375 for 100 and 200
// (tests/test_division_shortcuts.py: exact rational arithmetic; q1env_selftest_division on the device); for RUN-TIME constants the
// host checks the bound (q1env_host.hpp div_one_step_ok) and the SPEC kernels require it.  float32: the same with 2^-25.
template <typename T>
__device__ __forceinline__ T div_const1(T x, T c, T y) {
    const T q = x * y;
    const T r = fma_t<T>(-q, c, x);
    return fma_t<T>(r, y, q);
}
template <bool ONE_STEP, typename T>
__device__ __forceinline__ T div_const_sel(T x, T c, T y) {
    if constexpr (ONE_STEP) return div_const1<T>(x, c, y);
    else return div_const<T>(x, c, y);
}

// a / b for two numerators sharing one denominator: the refined reciprocal (v_rcp_f64 + two Newton steps, the
// same recurrence the compiler's own f64 division expands to) is computed once; each quotient is then
// q = a*y, r = a - b*q, q + r*y.  With operands in the normal range (|wish_vel| <= ~2000, speeds <= ~1e4)
// v_div_scale / v_div_fixup are identities, so the result equals the compiler's `/` bit for bit.
__device__ __forceinline__ double rcp_refined(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    return y;
}
__device__ __forceinline__ double div_shared(double a, double b, double y) {
    const double q = a * y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);          // a = +0 -> +0; a is never -0 here (einsum sums start from +0.0)
}

// sqrt(a) for 2^-767 <= a < inf: the compiler's correctly rounded float64 expansion (v_rsq_f64, one coupled Goldschmidt step,
// two residual corrections) without the ldexp scaling of tiny inputs and the zero / infinity pass-through around it.
__device__ __forceinline__ double sqrt_normal(double a) {
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y;
    double h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, a);
    g = fma(d, h, g);
    d = fma(-g, g, a);
    return fma(d, h, g);
}

// ---------------------------------------------------------------------------------------- sin / cos of the yaw
// The tick needs sin and cos of rad = yaw*pi/180, |rad| <~ 130 in an episode.  The library's sincos costs ~75 VALU instructions
// per tick in the fused kernels: a 17-instruction three-constant reduction written without FMA exactness in mind, ten v_mov_b64
// that re-load polynomial coefficients into the accumulator of a two-address v_fmac_f64, NaN / infinity selects, and a Payne-Hanek
// branch.  This restatement keeps the library's accuracy (fdlibm's __kernel_sin / __kernel_cos on a hi + lo reduced argument: against
// glibc - what NumPy calls - 3.1 % of results differ, by 1 ulp, over yaw in +-7500 degrees; tools/sincos_accuracy.c) at 43:
//   * n = rint(x 2/pi) and its integer come from ONE fma with the 1.5 2^52 bias (low word = n, two's complement)
//   * r0 = fma(-n, P1, x) is EXACT (P1 = RN(pi/2); n P1 is a multiple of 2^-52 and |r0| < 1), so hi = RN(r0 - n P2) and
//     lo = ((r0 - hi) - n P2) - n P3 need four more instructions (fdlibm spends three Cody-Waite rounds to get the same)
//   * the polynomial steps are three-address v_fma_f64 with the coefficient in a register pair (inline assembly: the compiler's own
//     selection is the two-address v_fmac_f64 + v_mov_b64)
// Lanes with |x| >= 2^20 (or NaN) send the whole wave to the library's sincos (never in an episode: yaw is unwrapped degrees).
__device__ __forceinline__ double fma3(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// The tick's float64 constants as opaque 64-bit register values: the eleven polynomial coefficients and the six thresholds whose
// low word is zero.  A kernel with a tick loop makes them ONCE, before the loop: inline assembly is not speculated, so the compiler
// cannot hoist it itself; without the opaque copy it assembles each coefficient pair from separately kept halves with a v_mov_b32
// per use, and re-builds each threshold in an SGPR pair with two s_mov_b32 per tick - and on a lone wave a scalar instruction costs
// the same 2.2 ns issue slot as a vector one (tools/ubench_select.hip).
struct TickConsts {
    double s2, s3, s4, s5, s6, c1, c2, c3, c4, c5, c6;
    double neg_bias, two20, tiny_wish, tiny_speed, max_wish;
    bool hoisted;                                                       // compile-time after inlining: which tick_consts*() made it
    // LDS table of the decoder's move commands (fill_move_table), or NULL: (fmove, smove) for every combination of this tick's and the
    // previous tick's forward / strafe key bits - 64 entries made once per launch with the reference's own arithmetic
    const double* move_tab;
    bool has_move_tab;                                                  // compile-time after inlining (a null test of an LDS pointer is not)
    // one Horner step z * acc + coefficient: three-address v_fma_f64 on the register-resident coefficient in a loop kernel, the
    // compiler's own choice in a single-tick kernel (where pinning 32 registers would cost the HBM-bound kernels their occupancy)
    __device__ __forceinline__ double horner(double z, double acc, double coeff) const {
        return hoisted ? fma3(z, acc, coeff) : fma(z, acc, coeff);
    }
};
__device__ __forceinline__ double opaque_vgpr(double c) {
    asm("" : "+v"(c));
    return c;
}
template <bool HOISTED>
__device__ __forceinline__ double tick_const(double c) { return HOISTED ? opaque_vgpr(c) : c; }
template <bool HOISTED>
__device__ __forceinline__ TickConsts make_tick_consts() {
    TickConsts t;                                                       // fdlibm k_sin.c S2..S6, k_cos.c C1..C6
    t.s2 = tick_const<HOISTED>(8.33333333332248946124e-03);  t.s3 = tick_const<HOISTED>(-1.98412698298579493134e-04);
    t.s4 = tick_const<HOISTED>(2.75573137070700676789e-06);  t.s5 = tick_const<HOISTED>(-2.50507602534068634195e-08);
    t.s6 = tick_const<HOISTED>(1.58969099521155010221e-10);
    t.c1 = tick_const<HOISTED>(4.16666666666666019037e-02);  t.c2 = tick_const<HOISTED>(-1.38888888888741095749e-03);
    t.c3 = tick_const<HOISTED>(2.48015872894767294178e-05);  t.c4 = tick_const<HOISTED>(-2.75573143513906633035e-07);
    t.c5 = tick_const<HOISTED>(2.08757232129817482790e-09);  t.c6 = tick_const<HOISTED>(-1.13596475577881948265e-11);
    t.neg_bias = tick_const<HOISTED>(-6755399441055744.0);              // -1.5 * 2^52
    t.two20 = tick_const<HOISTED>(1048576.0);
    t.tiny_wish = tick_const<HOISTED>(0x1p-600);
    t.tiny_speed = tick_const<HOISTED>(0x1p-100);
    t.max_wish = tick_const<HOISTED>(320.0);
    t.hoisted = HOISTED;
    t.move_tab = nullptr;
    t.has_move_tab = false;
    return t;
}
// before a tick loop: the constants pinned in registers
__device__ __forceinline__ TickConsts tick_consts() { return make_tick_consts<true>(); }

__device__ __forceinline__ void sincos_yaw(const TickConsts& k, double x, double& sn_out, double& cs_out) {
    const double nb = fma(x, 0.6366197723675814, -k.neg_bias);          // + 1.5 * 2^52: the low word is n (two's complement)
    const uint32_t q = (uint32_t)__double2loint(nb);                    // n mod 2^32 (only bits 0 and 1 are used)
    const double n = nb + k.neg_bias;
    const double r0 = fma(-n, 1.5707963267948966, x);                   // exact
    const double hi = fma(-n, 6.123233995736766e-17, r0);
    double lo = fma(-n, 6.123233995736766e-17, r0 - hi);
    lo = fma(-n, -1.4973849048591698e-33, lo);
    const double z = hi * hi;
    // sin: hi - ((z (lo/2 - v r) - lo) - v S1), r = S2 + z (S3 + z (S4 + z (S5 + z S6)))       cos: w + (((1 - w) - z/2) + (z^2 c - hi lo)),
    // w = 1 - z/2, c = C1 + z (C2 + z (C3 + z (C4 + z (C5 + z C6))))                           (fdlibm k_sin.c / k_cos.c)
    double rs = k.horner(z, k.s6, k.s5);
    double rc = k.horner(z, k.c6, k.c5);
    const double v = z * hi;
    const double hz = 0.5 * z;
    rs = k.horner(z, rs, k.s4);
    rc = k.horner(z, rc, k.c4);
    const double w = 1.0 - hz;
    const double zz = z * z;
    rs = k.horner(z, rs, k.s3);
    rc = k.horner(z, rc, k.c3);
    double e = (1.0 - w) - hz;
    rs = k.horner(z, rs, k.s2);
    rc = k.horner(z, rc, k.c2);
    e = fma(hi, -lo, e);
    double t = fma(-v, rs, 0.5 * lo);
    rc = k.horner(z, rc, k.c1);
    t = fma(z, t, -lo);
    e = fma(zz, rc, e);
    t = fma(-v, -1.66666666666666324348e-01, t);
    const double cs = w + e;
    const double sn = hi - t;
    // quadrant: sin(x) = {sn, cs, -sn, -cs}[n & 3], cos(x) = {cs, -sn, -cs, sn}[n & 3]: odd n swaps the two, bit 1 of n negates the
    // sine and bit 1 of n + 1 the cosine - so the swap selects plain values and each sign is one three-input bit operation
    double s_sel, c_sel;
    select2_f64(__ballot(q & 1u), cs, sn, s_sel, sn, cs, c_sel);
    const uint32_t flip_s = q << 30, flip_c = (q << 30) + 0x40000000u;   // bit 31 = bit 1 of n / of n + 1
    sn_out = __hiloint2double(__double2hiint(s_sel) ^ (int)(flip_s & 0x80000000u), __double2loint(s_sel));
    cs_out = __hiloint2double(__double2hiint(c_sel) ^ (int)(flip_c & 0x80000000u), __double2loint(c_sel));
    // |x| >= 2^20 or NaN anywhere in the wave: the library's path for the whole wave.  (The results pass through an empty asm first so
    // that the compiler cannot sink their last instructions into an else-side of this branch: one untaken s_cbranch per tick, not two.)
    if (k.hoisted) asm("" : "+v"(sn_out), "+v"(cs_out));
    if (__builtin_expect(__ballot(!(fabs(x) < k.two20)) != 0ull, 0)) sincos(x, &sn_out, &cs_out);
}

// ---------------------------------------------------------------------------------------- state I/O
// 32-bit unsigned lane index + SGPR base pointers: the loads/stores use the saddr addressing form instead of
// per-array 64-bit VGPR address arithmetic.
__device__ __forceinline__ void load_env(const StatePtrs& s, uint32_t n, uint32_t i, Env& e) {
    e.vx = s.vx[i]; e.vy = s.vy[i]; e.vz = s.vz[i];
    e.px = s.px[i]; e.py = s.py[i]; e.z = s.z[i];
    e.yaw = s.yaw[i]; e.trem = s.trem[i];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) e.lk[k] = (s.lk + (size_t)k * n)[i];
    e.flags = s.flags[i];
}

// ---- stores that are VISIBLE when they are acknowledged
// An MI355X has eight XCDs, each with its own write-back L2 that is not coherent with the others: an ordinary (or non-temporal) store is
// acknowledged by the issuing XCD's L2 and may sit there dirty until the dispatch's end-of-kernel release writes the cache back - after
// the last wave has gone, behind the runtime's completion path.  A kernel that announces its own completion (signal_done below) would
// then announce results a DMA engine, the host or another XCD cannot read yet.  A SYSTEM-scope store (sc0 sc1 on gfx942 / gfx950: a
// relaxed system-scope atomic store is exactly the plain store instruction with those two bits) is written THROUGH the L2 to the
// memory side (Infinity Cache / HBM, which every agent reads coherently) and its acknowledgement - what s_waitcnt vmcnt(0) waits for -
// comes from there.  "All my stores are acknowledged" then means "all my results are readable by anyone", with no cache write-back.
// The outputs and the final state of a multi-tick launch are written once and not read again by the launch: write-through costs nothing
// a write-back would not have to pay later.  tests/test_hip_signal.py reads them back through a DMA copy on a second stream the
// instant the signal arrives, 1 000 times, to hold the kernels to this.
__device__ __forceinline__ void store_sys(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void store_sys(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), (uint64_t)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void store_sys(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_sys(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__device__ __forceinline__ void store_env_sys(const StatePtrs& s, uint32_t n, uint32_t i, const Env& e) {
    store_sys(s.vx + i, e.vx); store_sys(s.vy + i, e.vy); store_sys(s.vz + i, e.vz);
    store_sys(s.px + i, e.px); store_sys(s.py + i, e.py); store_sys(s.z + i, e.z);
    store_sys(s.yaw + i, e.yaw); store_sys(s.trem + i, e.trem);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) store_sys(s.lk + (size_t)k * n + i, e.lk[k]);
    store_sys(s.flags + i, (uint8_t)e.flags);
}

// the same, non-temporal (the round-4 form, kept as the A/B: Q1_ROLLOUT_STATE_NT = 1): lines left dirty in L2 are written back by
// the dispatch's end-of-kernel release - after the last wave has gone
__device__ __forceinline__ void store_env_nt(const StatePtrs& s, uint32_t n, uint32_t i, const Env& e) {
    __builtin_nontemporal_store(e.vx, s.vx + i); __builtin_nontemporal_store(e.vy, s.vy + i); __builtin_nontemporal_store(e.vz, s.vz + i);
    __builtin_nontemporal_store(e.px, s.px + i); __builtin_nontemporal_store(e.py, s.py + i); __builtin_nontemporal_store(e.z, s.z + i);
    __builtin_nontemporal_store(e.yaw, s.yaw + i); __builtin_nontemporal_store(e.trem, s.trem + i);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) __builtin_nontemporal_store(e.lk[k], s.lk + (size_t)k * n + i);
    __builtin_nontemporal_store((uint8_t)e.flags, s.flags + i);
}

__device__ __forceinline__ void store_env(const StatePtrs& s, uint32_t n, uint32_t i, const Env& e) {
    s.vx[i] = e.vx; s.vy[i] = e.vy; s.vz[i] = e.vz;
    s.px[i] = e.px; s.py[i] = e.py; s.z[i] = e.z;
    s.yaw[i] = e.yaw; s.trem[i] = e.trem;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) (s.lk + (size_t)k * n)[i] = e.lk[k];
    s.flags[i] = (uint8_t)e.flags;
}

// The tick's write-back: position, horizontal velocity, yaw and time_remaining change on every tick; the key timestamps (a rising
// edge at most once per key_press_delay), the flag byte, z and vel_z (constant while the player stands on the floor) often do not.
// Product form (Q1_DELTA_PER_LANE = 1): those are written only by the lanes whose bit pattern changed (`old` = the state as loaded), up to
// 45 B of the 85 B.  The alternative VERDICT r3 item 7 asked for - written by the WHOLE WAVE or not at all, `__ballot(changed)` being
// wave-uniform, so that every store instruction covers full aligned lines (round 3 counted 1.37 write requests per env where 1.0 would
// carry the bytes) - was built and measured in round 4 (Q1_DELTA_PER_LANE = 0, tools/exp_step_large.py, profiles/r4_step_large.txt):
// 7.63 vs 7.37 us per tick at 262 144 envs, 38.4 vs 36.2 at 1 M, 157.0 vs 157.3 at 4 M - no gain, a loss at 1 M: the partial lines are
// not what holds the kernel below the copy kernel, and the per-lane form writes fewer bytes when keys are held (a trained policy).
#ifndef Q1_DELTA_PER_LANE            // 1 (product) = per-lane conditional stores; 0 = the whole-wave form above (tools/exp_step_large.py A/B)
#define Q1_DELTA_PER_LANE 1
#endif
__device__ __forceinline__ void store_env_delta(const StatePtrs& s, uint32_t n, uint32_t i, const Env& e, const Env& old) {
    if constexpr (Q1_DELTA_PER_LANE == 1) {
        s.vx[i] = e.vx; s.vy[i] = e.vy;
        if (__float_as_uint(e.vz) != __float_as_uint(old.vz)) s.vz[i] = e.vz;
        s.px[i] = e.px; s.py[i] = e.py;
        if (__double_as_longlong(e.z) != __double_as_longlong(old.z)) s.z[i] = e.z;
        s.yaw[i] = e.yaw; s.trem[i] = e.trem;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (__double_as_longlong(e.lk[k]) != __double_as_longlong(old.lk[k])) (s.lk + (size_t)k * n)[i] = e.lk[k];
        if ((e.flags & 0xFFu) != (old.flags & 0xFFu)) s.flags[i] = (uint8_t)e.flags;
        return;
    }
    s.vx[i] = e.vx; s.vy[i] = e.vy;
    if (__ballot(__float_as_uint(e.vz) != __float_as_uint(old.vz)) != 0ull) s.vz[i] = e.vz;
    s.px[i] = e.px; s.py[i] = e.py;
    if (__ballot(__double_as_longlong(e.z) != __double_as_longlong(old.z)) != 0ull) s.z[i] = e.z;
    s.yaw[i] = e.yaw; s.trem[i] = e.trem;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (__ballot(__double_as_longlong(e.lk[k]) != __double_as_longlong(old.lk[k])) != 0ull) (s.lk + (size_t)k * n)[i] = e.lk[k];
    if (__ballot((e.flags & 0xFFu) != (old.flags & 0xFFu)) != 0ull) s.flags[i] = (uint8_t)e.flags;
}

// ---------------------------------------------------------------------------------------- completion signal
// include/q1env.h "completion signal": sig = three 64-bit words of host-coherent pinned memory (start stamp, end stamp, sequence).
// Tickets are a two-level tree of RELAXED agent-scope counters (64 leaves 256 B apart + a root): 1 024 waves that finish within a
// microsecond of each other would otherwise queue on one address at the memory side (agent-scope atomics execute there on a
// multi-XCD part), and a release / acquire on every ticket would be an L2 write-back per wave - measured on MI355X: 30 us added to a
// 22 us kernel.  A wave's ticket only has to say "my stores have been acknowledged": s_waitcnt vmcnt(0) before it is enough; the ONE
// wave that draws the last root ticket performs the system-scope release and publishes.
struct Signal {
    uint64_t* sig;          // NULL: no stamps, no signal
    uint32_t* ticket;       // device counters: leaf g at ticket[64 * g] (g < 64), root at ticket[64 * 64]; all return to 0
    uint64_t seq;           // the sequence number this launch publishes
    uint32_t waves;         // waves of this launch that own at least one env
    uint32_t flags;         // bit 0: stamp the start; bit 1: stamp the end + publish seq
};
constexpr uint32_t SIGNAL_LEAVES = 64, SIGNAL_LEAF_STRIDE = 64;            // (uint32 units: 256 B)
constexpr uint32_t SIGNAL_SEQ_DEV_OFFSET = 16;                             // (uint32 units behind the root ticket: the device-resident sequence copy)
__device__ __forceinline__ void signal_start(const Signal& g) {
    if (g.sig && (g.flags & 1u) && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(g.sig, (uint64_t)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Called by every wave that owns an env (wave = its index among them), after its last store.
__device__ __forceinline__ void signal_done(const Signal& g, uint32_t wave) {
    if (!g.sig || !(g.flags & 2u)) return;
    __builtin_amdgcn_s_waitcnt(0);                                         // every store of this wave has been acknowledged
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t leaf = wave % SIGNAL_LEAVES;
        const uint32_t leaf_size = (g.waves + SIGNAL_LEAVES - 1u - leaf) / SIGNAL_LEAVES;
        uint32_t* lc = g.ticket + leaf * SIGNAL_LEAF_STRIDE;
        if (__hip_atomic_fetch_add(lc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == leaf_size) {
            __hip_atomic_store(lc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t leaves = g.waves < SIGNAL_LEAVES ? g.waves : SIGNAL_LEAVES;
            uint32_t* root = g.ticket + SIGNAL_LEAVES * SIGNAL_LEAF_STRIDE;
            if (__hip_atomic_fetch_add(root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == leaves) {
                __hip_atomic_store(root, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g.sig + 1, (uint64_t)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                // a device-resident copy of the sequence number (64 B behind the root ticket), for on-device consumers that poll for
                // this launch (the visibility reader of q1env_diag_signal_reader: polling host memory from the device proved slow)
                __hip_atomic_store(reinterpret_cast<uint64_t*>(root + SIGNAL_SEQ_DEV_OFFSET), g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(g.sig + 2, g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// The same for a launch whose RESULTS live in host-coherent memory (the host-direct small-batch path of q1env_step_host /
// q1env_reset_draws_host): the host reads them the moment it sees the sequence number, so every wave releases its stores at SYSTEM
// scope before its ticket and the ticket is acquire-release - the last wave's system-scope release of the sequence number then
// carries all of them.  One counter: these launches are a few waves (<= HOST_DIRECT_MAX_ENVS / 64).
__device__ __forceinline__ void signal_done_strict(const Signal& g) {
    if (!g.sig || !(g.flags & 2u)) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if ((threadIdx.x & 63u) == 0u) {
        uint32_t* root = g.ticket + SIGNAL_LEAVES * SIGNAL_LEAF_STRIDE;
        if (__hip_atomic_fetch_add(root, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == g.waves) {
            __hip_atomic_store(root, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.sig + 1, (uint64_t)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(g.sig + 2, g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------------------- Philox
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__host__ __device__ __forceinline__ void philox_draw(uint64_t seed, uint64_t genv, uint64_t counter,
                                                     uint32_t stream, uint32_t sub, uint32_t out[4]) {
    out[0] = (uint32_t)genv; out[1] = (uint32_t)(genv >> 32);
    out[2] = (uint32_t)counter;
    out[3] = (stream << 28) | ((sub & 0xFu) << 24) | ((uint32_t)(counter >> 32) & 0xFFFFFFu);
    philox4x32_10(out, (uint32_t)seed, (uint32_t)(seed >> 32));
}

__host__ __device__ __forceinline__ double u53(uint32_t a, uint32_t b) {   // [0,1) with 53 random bits
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// iid random action for (seed, global env, tick): Bernoulli(1/2) keys, mouse ~ U(-range, range) in float32
// (or a uniform discrete step).  Returns key bits; *yaw_act is what a policy would have emitted.
template <bool SPEC>
__device__ __forceinline__ uint32_t random_action(const Params& p, uint64_t seed, uint64_t genv, uint64_t tick,
                                                  double* yaw_act) {
    uint32_t r[4];
    philox_draw(seed, genv, tick, STREAM_ACTION, 0, r);
    const uint32_t keys = r[0] & ((1u << cfg_num_keys<SPEC>(p)) - 1u);
    if (cfg_yaw_mode<SPEC>(p) == 1) {
        const float u = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
        const float a = (u * 2.0f - 1.0f) * p.action_range_f32;
        *yaw_act = (double)a;
    } else if (cfg_yaw_mode<SPEC>(p) == 2) {
        const uint32_t m = 2u * (uint32_t)p.yaw_steps + 1u;
        *yaw_act = (double)(r[1] % m);
    } else {
        *yaw_act = 0.0;
    }
    return keys;
}

// ---------------------------------------------------------------------------------------- actions
// Fetch env idx's action from a device layout: key bits (bit k = trunc(a_k) & 1, the reference's
// `key_actions & (...)` on astype(int), env.py:228,243) and the mouse value.
template <bool SPEC, int FMT>
__device__ __forceinline__ uint32_t fetch_action(const Params& p, int fmt_rt, const void* a, const void* b,
                                                 size_t idx, double* yaw_act) {
    const int fmt = FMT >= 0 ? FMT : fmt_rt;
    const int nk = cfg_num_keys<SPEC>(p);
    uint32_t keys = 0;
    *yaw_act = 0.0;
    if (fmt == FMT_PACKED) {                          // 1 B keys + 4 B mouse
        keys = ((const uint8_t*)a)[idx] & ((1u << nk) - 1u);
        if (cfg_yaw_mode<SPEC>(p)) *yaw_act = (double)((const float*)b)[idx];
    } else if (fmt == FMT_F64_ROWS) {
        const double* row = (const double*)a + idx * (size_t)cfg_act_width<SPEC>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nk) keys |= (uint32_t)((long long)row[k] & 1) << k;
        if (cfg_yaw_mode<SPEC>(p)) *yaw_act = row[nk];
    } else {                                          // float32 rows
        const float* row = (const float*)a + idx * (size_t)cfg_act_width<SPEC>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nk) keys |= (uint32_t)((long long)row[k] & 1) << k;
        if (cfg_yaw_mode<SPEC>(p)) *yaw_act = (double)row[nk];
    }
    return keys;
}

// ---------------------------------------------------------------------------------------- decode
// The move commands of env.py:251-261,269 for given key bits: level_k = (key_k + prev_k) * 0.5 with smoothing, key_k without.  Every
// level is one of {0, 0.5, 1}, so the difference right - left and the forward level are formed EXACTLY in integers (units of
// smooth_scale) and converted once: the two products see the same float64 operands as the reference's (strafe_right - strafe_left)
// and forward levels.  keys / prev: bit 0 = left, 1 = right, 2 = forward.
__device__ __forceinline__ void move_commands(const Params& p, uint32_t keys, uint32_t prev, double& fmove, double& smove) {
    const int sp = p.smooth_keys ? 1 : 0;
    const int lr2 = ((int)((keys >> 1) & 1u) - (int)(keys & 1u)) + sp * ((int)((prev >> 1) & 1u) - (int)(prev & 1u));     // (lvl[1] - lvl[0]) / smooth_scale
    const int fw2 = (int)((keys >> 2) & 1u) + sp * (int)((prev >> 2) & 1u);                                                  // lvl[2] / smooth_scale
    smove = trunc(p.smove_max * ((double)lr2 * p.smooth_scale)) + 0.0;     // astype(int): toward zero, +0 (env.py:259-260,269)
    fmove = trunc(p.fmove_max * ((double)fw2 * p.smooth_scale)) + 0.0;     // env.py:261,269
}
// The 64 possible (fmove, smove) pairs of a launch, made once with move_commands itself: entry (keys & 7) | (prev & 7) << 3, which is
// (keys & 7) | (flags & 0x38) because the previous keys sit at bit 3 of the flag byte.  A multi-tick kernel then spends 4 instructions
// per tick on the commands (index, address, one 16-byte LDS read) instead of ~22 of bit arithmetic, conversions, products and
// truncations whose inputs can only take these 64 values.  Call from EVERY thread of the workgroup before any early exit.
__device__ __forceinline__ const double* fill_move_table(const Params& p, double* tab /* LDS, 128 doubles */) {
    for (uint32_t idx = threadIdx.x; idx < 64u; idx += blockDim.x) {
        double f, sm;
        move_commands(p, idx & 7u, idx >> 3, f, sm);
        tab[2u * idx] = f;
        tab[2u * idx + 1u] = sm;
    }
    __syncthreads();
    return tab;
}

// ActionDecoder.map for one env (env.py:225-269).  z_vel / trem are passed separately because the
// stand-alone decoder takes them from the caller (mkdemo.py:47-55).
template <bool SPEC>
__device__ __forceinline__ Cmd decode(const Params& p, const TickConsts& k_, Env& e, uint32_t keybits, double yaw_act,
                                      float z_vel, double trem) {
    const double now = p.time_limit - trem;                             // env.py:241,246
    const uint32_t prev = (e.flags >> FLAG_KEYS_SHIFT) & 0xFu;
    const int nk = cfg_num_keys<SPEC>(p);
    // env.py:241-248 for the four keys at once, as bit masks: may_press_k = now >= last_press_k + delay, keys = key_actions &
    // (may_press | last_keys), rising edge = keys & ~last_keys.  The float64 comparison is the SIGN of the float64 difference
    // now - (last_press_k + delay) - a correctly rounded difference is negative exactly when now < the sum, and +0 when they are equal
    // (the times are finite) - so the four results are gathered with one v_alignbit each (shift the accumulator left, bring the sign
    // bit in) instead of a compare, a select and a hazard no-op each.
    uint32_t neg = 0;
#pragma unroll
    for (int k = 3; k >= 0; --k) {
        const double d = now - (e.lk[k] + p.key_press_delay);
        neg = __builtin_amdgcn_alignbit(neg, (uint32_t)__double2hiint(d), 31);      // (neg << 1) | sign(d): bit k = now < last_press_k + delay
    }
    const uint32_t keys = keybits & ((1u << nk) - 1u) & (~neg | prev);  // env.py:243
    const uint32_t rise = keys & ~prev;
    // env.py:244-248: last_press_k = now where the key went down.  Per key one sign-extended bit field (all ones / zero) and one bit
    // select per 32-bit half: no SGPR masks, no hazard
    const uint32_t now_lo = (uint32_t)__double2loint(now), now_hi = (uint32_t)__double2hiint(now);
    {
        const uint32_t m[4] = {bit_mask_i32<0>(rise), bit_mask_i32<1>(rise), bit_mask_i32<2>(rise), bit_mask_i32<3>(rise)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = (uint32_t)__double2loint(e.lk[k]), hi = (uint32_t)__double2hiint(e.lk[k]);
            e.lk[k] = __hiloint2double((int)bfi_b32(m[k], now_hi, hi), (int)bfi_b32(m[k], now_lo, lo));
        }
    }
    Cmd c;
    if (k_.has_move_tab) {                                              // (compile-time after inlining)
        const uint32_t idx = (keys & 7u) | (e.flags & 0x38u);           // this tick's keys | the previous tick's, already at bit 3
        const double2 mv = *reinterpret_cast<const double2*>(k_.move_tab + 2u * idx);
        c.fmove = mv.x;
        c.smove = mv.y;
    } else {
        move_commands(p, keys, prev, c.fmove, c.smove);
    }
    e.flags = (e.flags & 0x7u) | (keys << FLAG_KEYS_SHIFT);             // env.py:256

    double dyaw = 0.0;
    if (cfg_yaw_mode<SPEC>(p) == 1) dyaw = div_const_sel<SPEC, double>(yaw_act * p.yaw_num, p.yaw_den, p.yaw_den_rcp);   // env.py:236
    else if (cfg_yaw_mode<SPEC>(p) == 2) dyaw = div_const<double>((yaw_act - p.yaw_steps) * p.yaw_num, p.yaw_den, p.yaw_den_rcp);   // env.py:238
    e.yaw = e.yaw + dyaw;                                               // env.py:258

    if (cfg_jump_mode<SPEC>(p) == 2) c.jump = z_vel <= 16.0f ? 1u : 0u;   // env.py:263
    else if (cfg_jump_mode<SPEC>(p) == 1) c.jump = (keys >> 3) & 1u;    // env.py:265
    else c.jump = 0u;                                                   // env.py:267
    return c;
}

// ---------------------------------------------------------------------------------------- physics
// phys.apply for one env (phys.py:184-197).  The 2x2 basis (forward | right, phys.py:65-66) is passed in:
// the env always has pitch = roll = 0, i.e. m = [[cos, sin], [sin, -cos]].
// Per-lane conditions are selects (or v_max / v_min), not branches; the only branch is the wave-level ballot of the PREVIOUS
// tick's on_ground: a wave with nobody on the ground skips the friction block (float32 sqrt + float64 divide) through a
// wave-uniform s_cbranch (an untaken branch costs a lone wave ~10 ns, four instructions' worth: tools/ubench_select.hip).
// VT = the dtype PlayerState.vel arrives in: float for the env (float32 storage: friction speed / control and the +270 add are
// float32 islands, the result is rounded to float32 on store, phys.py:190), double for DataFrame-driven callers
// (PlayerState.from_df yields a float64 vel, phys.py:168-170: then NOTHING on the path is float32).
// NORMAL = the caller guarantees |wish_vel|^2 is 0 or within [2^-700, 2^700] (the SPEC kernels: fmove_max / smove_max are checked on
// the host, q1env_host.hpp is_spec): the square root is then the compiler's own correctly rounded expansion minus its input scaling
// (an identity for 2^-767 <= x).
template <typename VT, bool NORMAL>
__device__ __forceinline__ void physics_core(const TickConsts& k_, VT& vx, VT& vy, VT& vz, double& zpos, uint32_t& flags, const Cmd& c,
                                             double m00, double m01, double m10, double m11,
                                             double dt, double accel_dt, double grav_dt) {
    const bool og = flags & FLAG_ON_GROUND;
    const uint64_t og_mask = __ballot(og);
    // einsum('ijk,ik->ij') accumulates from +0.0 (phys.py:97)
    const double wx = (0.0 + m00 * c.fmove) + m01 * c.smove;
    const double wy = (0.0 + m10 * c.fmove) + m11 * c.smove;
    // |wish_vel| (phys.py:98), with 2^-300 standing in where there is none.  The stand-in changes no result: without a wish
    // wx = wy = +0 (the einsum sums start from +0.0), so wish_dir = +0 / 2^-300 = +0 = the reference's pass-through (phys.py:99-101),
    // and the acceleration - some tiny positive number instead of 0 - multiplies +0: vel + (+0) either way.  One v_max_f64 instead of
    // the selects around the square root, the reciprocal and the two quotients.
    const double wsq = k_.hoisted ? max_f64_raw(wx * wx + wy * wy, k_.tiny_wish) : fmax(wx * wx + wy * wy, k_.tiny_wish);
    double wlen;
    if constexpr (NORMAL) wlen = sqrt_normal(wsq);
    else wlen = sqrt(wsq);
    const double yw = rcp_refined(wlen);
    const double dx = div_shared(wx, wlen, yw);                         // phys.py:99-101
    const double dy = div_shared(wy, wlen, yw);
    const double wish_speed = k_.hoisted ? min_f64_raw(k_.max_wish, wlen) : fmin(k_.max_wish, wlen);   // phys.py:103 (320)

    double hx = (double)vx, hy = (double)vy;
    if (__builtin_expect(og_mask != 0ull, 1)) {                         // wave-uniform skip (somebody stands on the floor in most waves: fall through)
        VT speed, control;                                              // norm / control in the velocity's own dtype (phys.py:85-86)
        if constexpr (sizeof(VT) == 4) {
            // float32 norm: sum of squares in float32, then the correctly rounded float32 square root.  Through float64: sqrt is one of
            // the operations whose double rounding is innocuous when the wide format has >= 2 p + 2 bits (53 >= 50), so
            // RN32(RN64(sqrt(x))) = RN32(sqrt(x)); and sqrt_normal's 10 instructions replace the compiler's float32 expansion (input
            // scaling, v_sqrt_f32, two residual tests, un-scaling, class test: 17 + 6 hazard no-ops).  x = 0 gives NaN here and is
            // discarded by the `speed > 0` mask below, like the reference's 0 / 0 (phys.py:90).
            const float x = vx * vx + vy * vy;
            if constexpr (NORMAL) speed = (float)sqrt_normal((double)x);
            else speed = sqrtf(x);
            control = fmaxf(speed, 100.0f);
        } else { speed = sqrt(vx * vx + vy * vy); control = fmax(speed, 100.0); }
        const double drop = (dt * (double)control) * 4.0;               // phys.py:87
        const double ns = fmax(0.0, (double)speed - drop);              // phys.py:88
        const double sd = k_.hoisted ? max_f64_raw((double)speed, k_.tiny_speed) : fmax((double)speed, k_.tiny_speed);   // (speed == 0: any divisor, the quotient is not used)
        const double k = div_shared(ns, sd, rcp_refined(sd));           // phys.py:90 (exact: operands in the normal range)
        select2_into_f64(og_mask & __ballot(speed > (VT)0), (double)vx * k, hx, (double)vy * k, hy);
    }
    const double cur = (0.0 + hx * dx) + hy * dy;                       // phys.py:71
    // phys.py:73-75: in the air a wish_speed above 30 is clipped to 30.  wish_speed <= 320, so min(wish_speed, og ? 320 : 30) is
    // the same number, and the two bounds differ in the high word only: one 32-bit select
    const double cap = __hiloint2double(og ? 0x40740000 : 0x403e0000, 0);                   // 320.0 : 30.0
    const double capped = k_.hoisted ? min_f64_raw(wish_speed, cap) : fmin(wish_speed, cap);
    const double add = fmax(0.0, capped - cur);                         // phys.py:77
    const double acc = fmin(accel_dt * wish_speed, add);                // phys.py:78 (unclipped wish_speed)
    vx = (VT)(hx + acc * dx);                                           // phys.py:80, RNE to float32 at phys.py:190
    vy = (VT)(hy + acc * dy);

    // z (phys.py:112-132), the flag logic as bit arithmetic on bit 0 = on_ground, bit 1 = jump_released, jump = 0 / 1:
    // jump_released |= ~jump (117); do_jump = on_ground & jump & jump_released (118) = on_ground & jump & the OLD jump_released
    const uint32_t dj = flags & (flags >> 1) & c.jump;                  // 0 / 1 (c.jump has no other bits)
    VT z_vel;
    if constexpr (sizeof(VT) == 4) z_vel = fmaf((float)dj, 270.0f, vz); // vz + (do_jump ? 270 : 0) in float32 (119): the product is exact
    else z_vel = vz + (dj ? (VT)270 : (VT)0);
    z_vel = (VT)((double)z_vel - grav_dt);                              // float64 subtract, RNE (phys.py:122)
    const double z = zpos + dt * (double)z_vel;                         // phys.py:127
    const bool landed = z < 24.03125;                                   // phys.py:128
    zpos = fmax(z, 24.03125);                                           // phys.py:129 (landed ? floor : z)
    vz = landed ? (VT)0 : z_vel;                                        // phys.py:130
    flags = ((flags | ((c.jump ^ 1u) << 1)) & ~FLAG_ON_GROUND) | (landed ? FLAG_ON_GROUND : 0u);
}

template <bool NORMAL>
__device__ __forceinline__ void physics(const TickConsts& k_, Env& e, const Cmd& c, double m00, double m01, double m10, double m11,
                                        double dt, double accel_dt, double grav_dt) {
    physics_core<float, NORMAL>(k_, e.vx, e.vy, e.vz, e.z, e.flags, c, m00, m01, m10, m11, dt, accel_dt, grav_dt);
}

// yaw -> basis with pitch = roll = 0 (phys.py:56-66): radians = yaw*pi/180 (mul THEN div), float64 sincos
template <bool SPEC>
__device__ __forceinline__ void physics_yaw_only(const Params& p, const TickConsts& tc, Env& e, const Cmd& c) {
    const double rad = div_const1<double>(e.yaw * 3.141592653589793, 180.0, 1.0 / 180.0);   // |180 RN(1/180) - 1| = 0.6875 x 2^-54: one step
    double sn, cs;
    sincos_yaw(tc, rad, sn, cs);
    physics<SPEC>(tc, e, c, cs, sn, sn, -cs, p.dt, p.accel_dt, p.grav_dt);
}

// env.py:392-400 with _round_origin (385-390), _round_vel (381-383), get_obs_scale (294-296).
// OBS_T = double reproduces the reference's float64 row.  OBS_T = float is DEFINED as that row rounded to float32, computed without
// the float64 divisions for the four columns whose numerators are small integers:
//   z:   rint(8 z) = j, the reference's value is RN64(RN64(j / 8) / 100), i.e. j / 800 to within 2^-53; j / 800 = j / (25 * 32) is never
//        within 2^-29.6 (relative) of a float32 rounding boundary for 0 < |j| < 2^24, and never ON one (that needs j >= 25 * 2^24), so
//        ANY float64 value within 2^-52 of j / 800 rounds to the same float32: RN32(j * RN64(1 / 800)), one product and one conversion
//   vel: trunc(v / 16) = m, the reference's value is RN64(16 m / 200) = 2 m / 25 to within 2^-53 and the same argument makes its
//        float32 rounding equal to the correctly rounded float32 quotient m / 12.5, which one Markstein step on RN32(0.08) delivers
//        (|12.5 RN32(0.08) - 1| = 0.75 x 2^-25; m = -0 yields +0 like the reference's integer cast)
// Both are checked EXHAUSTIVELY - every j and every m below 2^24 in magnitude - by tests/test_division_shortcuts.py (NumPy float64
// emulation, exact for these operand widths) and on the device by q1env_selftest_division.
// ONE_STEP: time_limit's reciprocal passes the one-step bound (SPEC kernels); 90 always does.
template <typename OBS_T, bool ONE_STEP = false>
__device__ __forceinline__ void observe(const Params& p, const Env& e, OBS_T o[6]) {
    o[0] = (OBS_T)div_const_sel<ONE_STEP, double>(e.trem, p.time_limit, p.time_limit_rcp);
    o[1] = (OBS_T)div_const1<double>(e.yaw, 90.0, 1.0 / 90.0);
    if constexpr (sizeof(OBS_T) == 8) {
        o[2] = div_const<double>(rint(e.z * 8.0) * 0.125, 100.0, 1.0 / 100.0);
        o[3] = div_const<double>(trunc((double)(e.vx * 0.0625f)) * 16.0 + 0.0, 200.0, 1.0 / 200.0);
        o[4] = div_const<double>(trunc((double)(e.vy * 0.0625f)) * 16.0 + 0.0, 200.0, 1.0 / 200.0);
        o[5] = div_const<double>(trunc((double)(e.vz * 0.0625f)) * 16.0 + 0.0, 200.0, 1.0 / 200.0);
    } else {
        o[2] = (float)(rint(e.z * 8.0) * (1.0 / 800.0));
        o[3] = div_const1<float>(truncf(e.vx * 0.0625f), 12.5f, 0.08f);
        // (vel_y, vel_z) as ONE two-wide float32 chain (v_pk_mul / v_pk_fma): columns 4 and 5 are neighbours in the row, so the result
        // pair is the register pair the row's third 8-byte LDS write takes
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 m = {e.vy, e.vz};
        m = __builtin_elementwise_trunc(m * 0.0625f);
        const f2 q = m * 0.08f;
        const f2 r = __builtin_elementwise_fma(-q, (f2)12.5f, m);
        const f2 o45 = __builtin_elementwise_fma(r, (f2)0.08f, q);
        o[4] = o45.x;
        o[5] = o45.y;
    }
}

// VectorPhysEnv.vector_step for one env (env.py:482-510)
// tc: tick_consts(), made before the tick loop by the kernels that have one
template <typename OBS_T, bool SPEC>
__device__ __forceinline__ void tick(const Params& p, const TickConsts& tc, Env& e, uint32_t keybits, double yaw_act, TickOut<OBS_T>& out) {
    if (cfg_hover<SPEC>(p)) { e.vz = 0.0f; e.z = 100.0; }               // env.py:483-485
    const Cmd c = decode<SPEC>(p, tc, e, keybits, yaw_act, e.vz, e.trem);
    physics_yaw_only<SPEC>(p, tc, e, c);
    if (cfg_speed_reward<SPEC>(p)) out.reward = p.dt_f32 * sqrtf(e.vx * e.vx + e.vy * e.vy);   // env.py:501 (float32)
    else out.reward = p.dt_f32 * e.vy;                                  // env.py:503 (float32)
    e.px = e.px + p.dt * (double)e.vx;                                  // extension: distance integrals
    e.py = e.py + p.dt * (double)e.vy;
    e.trem = e.trem - p.dt;                                             // env.py:505
    out.done = e.trem < 0.0;                                            // env.py:506
    observe<OBS_T, SPEC>(p, e, out.obs);
}

template <typename OBS_T, bool SPEC>
__device__ __forceinline__ void tick(const Params& p, Env& e, uint32_t keybits, double yaw_act, TickOut<OBS_T>& out) {
    tick<OBS_T, SPEC>(p, make_tick_consts<false>(), e, keybits, yaw_act, out);
}

// ---------------------------------------------------------------------------------------- resets
// Initial player state on the 100 m map (env.py:54-58) + decoder reset (env.py:277-281 / 289-291)
// from raw draws (env.py:432-451 / 461-476).
__device__ __forceinline__ void reset_from_draws(const Params& p, Env& e, bool zero_start, double yaw_draw,
                                                 double time_draw, double speed_draw, double angle_draw) {
    double speed = zero_start ? 0.0 : speed_draw;
    double angle = angle_draw;
    if (p.hover) { speed = 320.0; angle = 1.5707963267948966; }         // env.py:443-448
    double sn, cs;
    sincos(angle, &sn, &cs);
    e.vx = (float)(speed * cs);                                         // env.py:450
    e.vy = (float)(speed * sn);                                         // env.py:451
    e.vz = -12.0f;
    e.z = (double)32.843201f;
    e.px = 0.0; e.py = 0.0;
    e.yaw = zero_start ? 90.0 : yaw_draw;                               // env.py:434-436
    e.trem = zero_start ? p.time_limit : time_draw;                     // env.py:437-439
#pragma unroll
    for (int k = 0; k < 4; ++k) e.lk[k] = -p.key_press_delay;           // env.py:277-278
    e.flags = FLAG_JUMP_RELEASED | (zero_start ? FLAG_ZERO_START : 0u); // on_ground False, last_keys False
}

// Device RNG reset: the reference's distributions, including uniform(x) == uniform(low=x, high=1.0)
// (env.py:439,442,446): value = low + (high - low) * u.
__device__ __forceinline__ void reset_philox(const Params& p, Env& e, uint64_t seed, uint64_t genv, uint64_t counter) {
    uint32_t r0[4], r1[4], r2[4];
    philox_draw(seed, genv, counter, STREAM_RESET, 0, r0);
    philox_draw(seed, genv, counter, STREAM_RESET, 1, r1);
    philox_draw(seed, genv, counter, STREAM_RESET, 2, r2);
    const bool zs = u53(r0[0], r0[1]) < p.zero_start_prob;              // env.py:432
    const double yaw = p.yaw_lo + (p.yaw_hi - p.yaw_lo) * u53(r0[2], r0[3]);               // env.py:436
    const double tm = p.time_limit + (1.0 - p.time_limit) * u53(r1[0], r1[1]);             // env.py:439
    const double sp = p.max_initial_speed + (1.0 - p.max_initial_speed) * u53(r1[2], r1[3]);   // env.py:442
    const double an = 6.283185307179586 + (1.0 - 6.283185307179586) * u53(r2[0], r2[1]);   // env.py:446
    reset_from_draws(p, e, zs, yaw, tm, sp, an);
}

// ---------------------------------------------------------------------------------------- observation rows
template <typename T>
__device__ __forceinline__ void write_obs(T* obs, size_t i, const T o[6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j) obs[i * 6 + j] = o[j];
}

// Row-major (N,6) float32 observations are what a policy network consumes, but a lane owning one row means six
// 4-byte stores at a 24-byte lane stride per wave.  Stage the wave's 64 rows (1536 contiguous bytes) through a
// wave-private LDS slab instead and write them out as three all-lane 8-byte-per-lane stores (512 contiguous bytes
// each): 3 fully coalesced VMEM instructions instead of 6 strided ones, and - because every lane takes part in each -
// a STATIC number of stores per tick, which lets the compiler wait for a prefetched load with vmcnt(#stores) instead
// of draining the stores (gfx9 counts loads and stores in one in-order vmcnt).  LDS ops of one wave execute in order,
// so only a wave-scope fence + compiler-level wave barrier is needed between the write and the transposed read.
// `slab` = this wave's 384 floats.  Only for waves whose 64 lanes are all active (the transposed store needs every
// lane); the ragged tail wave writes its rows directly with write_obs.
__device__ __forceinline__ void write_obs_wave_f32(float* obs, size_t wave_first, uint32_t lane, const float o[6],
                                                   float* slab) {
    float2* w = reinterpret_cast<float2*>(slab + lane * 6);
    w[0] = make_float2(o[0], o[1]);
    w[1] = make_float2(o[2], o[3]);
    w[2] = make_float2(o[4], o[5]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float2* dst = reinterpret_cast<float2*>(obs + wave_first * 6);
    const float2* src = reinterpret_cast<const float2*>(slab);
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k) dst[k * 64u + lane] = src[k * 64u + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Same, with system-scope (write-through) stores: see store_sys.
__device__ __forceinline__ void write_obs_wave_f32_sys(float* obs, size_t wave_first, uint32_t lane, const float o[6],
                                                       float* slab) {
    float2* w = reinterpret_cast<float2*>(slab + lane * 6);
    w[0] = make_float2(o[0], o[1]);
    w[1] = make_float2(o[2], o[3]);
    w[2] = make_float2(o[4], o[5]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint64_t* dst = reinterpret_cast<uint64_t*>(obs + wave_first * 6);
    const uint64_t* src = reinterpret_cast<const uint64_t*>(slab);
    // all three LDS reads first, then the stores: the compiler keeps other memory operations in program order around an atomic store,
    // and would otherwise issue read / wait / store three times (one LDS latency per store instead of one per row)
    uint64_t v0 = src[lane], v1 = src[64u + lane], v2 = src[128u + lane];
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2));
    store_sys(dst + lane, v0);
    store_sys(dst + 64u + lane, v1);
    store_sys(dst + 128u + lane, v2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Same, with non-temporal stores (outputs are written once and read by a later kernel / the host: no reason to keep them in L2).
__device__ __forceinline__ void write_obs_wave_f32_nt(float* obs, size_t wave_first, uint32_t lane, const float o[6],
                                                      float* slab) {
    float2* w = reinterpret_cast<float2*>(slab + lane * 6);
    w[0] = make_float2(o[0], o[1]);
    w[1] = make_float2(o[2], o[3]);
    w[2] = make_float2(o[4], o[5]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2* dst = reinterpret_cast<f2*>(obs + wave_first * 6);
    const f2* src = reinterpret_cast<const f2*>(slab);
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k) __builtin_nontemporal_store(src[k * 64u + lane], dst + k * 64u + lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace q1
