/* A plain-C client of the drop-in boundary: include/q1env.h + libq1env.so, nothing else of this repo's Python.
 * It is what a non-Python host (or a maintainer's cffi/cgo stub) would write, and it is a parity test: the library's
 * q1env_step_host / q1env_get_state_host against the C oracle (oracle/q1_oracle.c, TEST INFRASTRUCTURE - linked here as the
 * checker only) on a seeded 300-tick rollout of 1 000 envs from a zero start, default Config (reference env.py:150-170).
 *
 *   gcc -O2 -ffp-contract=off -I include tests/c_abi_client.c oracle/q1_oracle.c -o /tmp/c_abi_client \
 *       -L q1physrl_amd -lq1env -Wl,-rpath,$PWD/q1physrl_amd -lm -fopenmp
 * Exit status 0 = every reward / done / observation of every tick and the final state agree (integers exactly, floats <= 1e-5
 * relative with >= 99.9 % of the velocity words bit-identical); 2 = no GPU (the library has no CPU fallback and says so).
 * Run by tests/test_c_abi_client.py. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "q1env.h"

/* the oracle's interface (oracle/q1_oracle.c) */
typedef struct { int32_t num_keys, yaw_mode, jump_mode, smooth_keys, hover, speed_reward; double dt, time_limit, key_press_delay, yaw_num, yaw_den, yaw_steps, fmove_max, smove_max; } q1o_params;
typedef struct { float *vel; double *z_pos, *yaw, *t_rem, *last_press; uint8_t *on_ground, *jump_released, *last_keys; } q1o_state;
void q1o_make_params(q1o_params *p, int allow_yaw, int discrete_yaw_steps, int auto_jump, int allow_jump, int smooth_keys, int hover,
                     int speed_reward, double dt, double time_limit, double key_press_delay, double action_range, double fmove_max, double smove_max);
void q1o_step(const q1o_params *p, const q1o_state *s, int64_t n, const double *actions, double *obs, float *reward, uint8_t *done, int threads);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd32(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, q1env_last_error()); return rc_ == Q1ENV_ERR_NO_DEVICE ? 2 : 1; } } while (0)

int main(void) {
    const int n = 1000, ticks = 300;
    const double dt = 1.0 / 72, action_range = (double)(720.0f * 0.014f);
    if (q1env_abi_version() != Q1ENV_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    q1env_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.num_envs = n; cfg.allow_yaw = 1; cfg.discrete_yaw_steps = -1; cfg.smooth_keys = 1; cfg.allow_jump = 1;
    cfg.zero_start_prob = 1.0; cfg.initial_yaw_lo = 0; cfg.initial_yaw_hi = 360; cfg.max_initial_speed = 700;
    cfg.time_delta = dt; cfg.time_limit = 10.0; cfg.action_range = action_range; cfg.fmove_max = 800; cfg.smove_max = 1060;
    cfg.key_press_delay = 0.3;
    q1env_t *env = NULL;
    CHECK(q1env_create(&cfg, 0, NULL, &env));                 /* state after create = zero-start reset of every env (env.py:54-58) */
    if (q1env_num_keys(env) != 4 || q1env_action_width(env) != 5) { fprintf(stderr, "action layout\n"); return 1; }

    /* oracle state: the same zero start */
    q1o_params op;
    q1o_make_params(&op, 1, -1, 0, 1, 1, 0, 0, dt, 10.0, 0.3, action_range, 800, 1060);
    q1o_state os;
    os.vel = calloc((size_t)n * 3, sizeof(float)); os.z_pos = malloc(n * sizeof(double)); os.yaw = malloc(n * sizeof(double));
    os.t_rem = malloc(n * sizeof(double)); os.last_press = malloc((size_t)n * 4 * sizeof(double));
    os.on_ground = calloc(n, 1); os.jump_released = malloc(n); os.last_keys = calloc((size_t)n * 4, 1);
    for (int i = 0; i < n; ++i) {
        os.vel[3 * i + 2] = -12.0f; os.z_pos[i] = (double)32.843201f; os.yaw[i] = 90.0; os.t_rem[i] = 10.0; os.jump_released[i] = 1;
        for (int k = 0; k < 4; ++k) os.last_press[4 * i + k] = -0.3;
    }

    double *act = malloc((size_t)n * 5 * sizeof(double));
    double *obs_g = malloc((size_t)n * 6 * sizeof(double)), *obs_o = malloc((size_t)n * 6 * sizeof(double));
    float *rew_g = malloc(n * sizeof(float)), *rew_o = malloc(n * sizeof(float));
    uint8_t *done_g = malloc(n), *done_o = malloc(n), *zs = malloc(n), *keys = calloc(n, 1);
    double worst = 0.0;
    for (int t = 0; t < ticks; ++t) {
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < 4; ++k) if (rnd32() % 20 == 0) keys[i] ^= (uint8_t)(1u << k);      /* keys flip with p = 0.05 */
            for (int k = 0; k < 4; ++k) act[5 * i + k] = (keys[i] >> k) & 1;
            act[5 * i + 4] = (double)(float)(((double)rnd32() / 4294967296.0 * 2.0 - 1.0) * action_range);   /* a float32 Box sample */
        }
        CHECK(q1env_step_host(env, Q1ENV_ACT_F64_ROWS, act, NULL, Q1ENV_OBS_F64, obs_g, rew_g, done_g, zs));
        q1o_step(&op, &os, n, act, obs_o, rew_o, done_o, 1);
        for (int i = 0; i < n; ++i) {
            if (done_g[i] != done_o[i] || zs[i] != 1) { fprintf(stderr, "tick %d env %d: done/zero_start mismatch\n", t, i); return 1; }
            const double er = fabs((double)rew_g[i] - (double)rew_o[i]) / fmax(fabs((double)rew_o[i]), 1.0);
            if (er > worst) worst = er;
            for (int j = 0; j < 6; ++j) {
                const double e = fabs(obs_g[6 * i + j] - obs_o[6 * i + j]) / fmax(fabs(obs_o[6 * i + j]), 1.0);
                if (e > worst) worst = e;
            }
        }
    }
    /* final state through the state-exchange entry point */
    float *vx = malloc(n * sizeof(float)), *vy = malloc(n * sizeof(float)), *vz = malloc(n * sizeof(float));
    double *z = malloc(n * sizeof(double)), *yaw = malloc(n * sizeof(double)), *tr = malloc(n * sizeof(double));
    uint8_t *flags = malloc(n);
    q1env_state st;
    memset(&st, 0, sizeof st);
    st.vel_x = vx; st.vel_y = vy; st.vel_z = vz; st.z_pos = z; st.yaw = yaw; st.time_remaining = tr; st.flags = flags;
    CHECK(q1env_get_state_host(env, &st));
    long same = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t a[3], b[3];
        memcpy(&a[0], &vx[i], 4); memcpy(&a[1], &vy[i], 4); memcpy(&a[2], &vz[i], 4);
        memcpy(b, os.vel + 3 * i, 12);
        same += (a[0] == b[0]) + (a[1] == b[1]) + (a[2] == b[2]);
        if (((flags[i] & Q1ENV_FLAG_ON_GROUND) != 0) != (os.on_ground[i] != 0)) { fprintf(stderr, "env %d: on_ground mismatch\n", i); return 1; }
        if (tr[i] != os.t_rem[i] || z[i] != os.z_pos[i]) { fprintf(stderr, "env %d: time_remaining / z mismatch\n", i); return 1; }
        const double ey = fabs(yaw[i] - os.yaw[i]) / fmax(fabs(os.yaw[i]), 1.0);
        if (ey > worst) worst = ey;
    }
    /* ABI v4: the kernel-written completion signal and the build id, from plain C */
    {
        uint64_t t_start = 0, t_end = 0;
        double hz = 0.0;
        CHECK(q1env_signal_mark(env));
        CHECK(q1env_signal_wait(env, 5.0));
        CHECK(q1env_signal_read(env, &t_start, &t_end, &hz));
        if (!(hz > 0.0) || t_end == 0) { fprintf(stderr, "completion signal: no stamp (hz %g, end %llu)\n", hz, (unsigned long long)t_end); return 1; }
        const char *bid = q1env_build_id();
        if (!bid || strlen(bid) != 16) { fprintf(stderr, "q1env_build_id: %s\n", bid ? bid : "(null)"); return 1; }
        printf("c_abi_client: completion signal ok (device clock %.0f Hz), library build id %s\n", hz, bid);
    }
    CHECK(q1env_destroy(env));
    const double frac = (double)same / (3.0 * n);
    printf("c_abi_client: %d envs x %d ticks, max rel err %.3e, vel bit-identical %.5f\n", n, ticks, worst, frac);
    if (worst > 1e-5 || frac < 0.999) return 1;
    printf("C_ABI_CLIENT_OK\n");
    return 0;
}
