"""NumPy restatement of the library's counter RNG (Philox4x32-10 and the draw layouts of
q1physrl_amd/csrc/q1env_device.hpp: philox_draw, random_action, reset_philox).  TEST INFRASTRUCTURE: lets the
tests predict device-generated random actions / resets and feed them to the oracle."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
STREAM_ACTION, STREAM_RESET = 1, 2
U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & U32
        n1 = p1 & U32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & U32
        n3 = p0 & U32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def draw(seed, genv, counter, stream, sub):
    genv = np.asarray(genv, dtype=np.uint64)
    counter = np.broadcast_to(np.asarray(counter, dtype=np.uint64), genv.shape)
    c3 = np.uint64((stream << 28) | ((sub & 0xF) << 24)) | ((counter >> np.uint64(32)) & np.uint64(0xFFFFFF))
    return philox4x32_10(genv & U32, genv >> np.uint64(32), counter & U32, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def u53(a, b):
    return ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


def random_actions(cfg, seed, genv, tick):
    """(N, A) float64 action rows the device generates for Q1ENV_ACT_RANDOM at `tick`."""
    r0, r1, _, _ = draw(seed, genv, tick, STREAM_ACTION, 0)
    k = cfg.num_keys
    keys = ((r0[:, None] >> np.arange(k, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float64)
    if not cfg.allow_yaw:
        return keys
    if cfg.discrete_yaw_steps == -1:
        u = (r1 >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        a = (u * np.float32(2) - np.float32(1)) * np.float32(cfg.action_range)
        yaw = a.astype(np.float64)
    else:
        yaw = (r1 % np.uint64(2 * cfg.discrete_yaw_steps + 1)).astype(np.float64)
    return np.concatenate([keys, yaw[:, None]], axis=1)


def reset_draws(cfg, seed, genv, counter):
    """(zero_start, yaw, time_remaining, speed, angle) exactly as reset_philox draws them on the device."""
    a = draw(seed, genv, counter, STREAM_RESET, 0)
    b = draw(seed, genv, counter, STREAM_RESET, 1)
    c = draw(seed, genv, counter, STREAM_RESET, 2)
    zs = u53(a[0], a[1]) < cfg.zero_start_prob
    lo, hi = cfg.initial_yaw_range
    yaw = lo + (hi - lo) * u53(a[2], a[3])
    tm = cfg.time_limit + (1.0 - cfg.time_limit) * u53(b[0], b[1])
    sp = cfg.max_initial_speed + (1.0 - cfg.max_initial_speed) * u53(b[2], b[3])
    an = 6.283185307179586 + (1.0 - 6.283185307179586) * u53(c[0], c[1])
    return zs, yaw, tm, sp, an
