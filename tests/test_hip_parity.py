"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes -> libq1env.so), against
  (1) golden vectors produced by the reference itself (tests/golden, made by oracle/gen_golden.py), and
  (2) the oracle (oracle/np_oracle.py) on fresh seeded inputs.

Bar: integer / boolean / index outputs (done, on_ground, zero_start, last_keys, smove, fmove, jump)
BIT-EXACT; floating point within REL_TOL = 1e-5 relative to max(|ref|, 1) - the tolerance BASELINE.json's
north_star states - over the whole 720-tick (10 s) rollout.  The float64 sincos of the device library may
differ from NumPy's in the last ulp, which is why floats are not required to be bit-identical; the
fraction of bit-identical elements is asserted separately (>= 99.9 %) so that a systematic deviation
(wrong dtype, fused multiply-add, reordered arithmetic) cannot hide inside the tolerance.
"""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests import _replay as R

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5
INT_FIELDS = ("done", "zero_start", "on_ground", "jump_released", "last_keys")
FLOAT_FIELDS = ("obs", "reward", "vel", "z_pos", "yaw", "time_remaining", "last_key_press_time")


def hip_env(kw):
    from q1physrl_amd import env as E
    return E.VectorPhysEnv(kw)


HIP_GETTERS = {
    "vel": lambda e: e.player_state.vel, "z_pos": lambda e: e.player_state.z_pos,
    "on_ground": lambda e: e.player_state.on_ground, "jump_released": lambda e: e.player_state.jump_released,
    "yaw": lambda e: e._yaw, "time_remaining": lambda e: e._time_remaining,
    "last_key_press_time": lambda e: e._action_decoder._last_key_press_time,
    "last_keys": lambda e: e._action_decoder._last_keys.astype(np.uint8),
}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0)


def bit_identical_fraction(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, b.dtype, a.shape, b.shape)
    u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return float(np.mean(a.view(u) == b.view(u)))


def compare(res, ref, name, min_bit_frac=0.999):
    for k in INT_FIELDS:
        assert np.array_equal(np.asarray(res[k]).astype(np.int64), np.asarray(ref[k]).astype(np.int64)), f"{name}: {k} mismatch"
    worst = {}
    for k in FLOAT_FIELDS:
        assert res[k].dtype == ref[k].dtype, (k, res[k].dtype, ref[k].dtype)
        e = float(np.max(rel_err(res[k], ref[k]))) if res[k].size else 0.0
        f = bit_identical_fraction(res[k], ref[k]) if res[k].size else 1.0
        worst[k] = (e, f)
        assert e <= REL_TOL, f"{name}: {k} max rel err {e:.3e} > {REL_TOL}"
        assert f >= min_bit_frac, f"{name}: {k} only {f:.5f} of elements bit-identical"
    return worst


@pytest.mark.parametrize("name", R.TRACE_FIXTURES)
def test_golden_trace(name):
    """Every reference-generated trace: zero-start 720-tick rollouts (G2), full Config with RLlib-style
    reset_at on done (G3), all Config variants (G4), the reference's own test scenario (S1), S2."""
    fx = R.load(name)
    res, env = R.replay(hip_env, fx, name, HIP_GETTERS)
    assert res["obs0"].dtype == np.float64 and res["obs0"].shape == fx["obs0"].shape
    assert np.max(rel_err(res["obs0"], fx["obs0"])) <= REL_TOL
    worst = compare(res, fx, name)
    assert res["reset_obs"].shape == fx["reset_obs"].shape
    if fx["reset_obs"].size:
        assert np.max(rel_err(res["reset_obs"], fx["reset_obs"])) <= REL_TOL
    assert np.max(rel_err(env.player_state.vel, fx["final_vel"])) <= REL_TOL
    print(name, {k: f"{e:.1e}/{f:.5f}" for k, (e, f) in worst.items()})
    env.close()


def test_position_parity_720_ticks():
    """BASELINE metric: max |pos - ref| over the 10 s rollout.  pos_y of the reference = float64 running sum
    of dt*vel_y over its per-tick float32 velocities (the reference never integrates x/y itself)."""
    fx = R.load("g2_zero_start_720")
    res, env = R.replay(hip_env, fx, "g2", {"dist": lambda e: e.distance, "z": lambda e: e.player_state.z_pos})
    dt = 1.0 / 72
    ref_xy = np.cumsum(dt * fx["vel"][:, :, :2].astype(np.float64), axis=0)
    err_xy = np.abs(res["dist"] - ref_xy)
    err_z = np.abs(res["z"] - fx["z_pos"])
    scale = np.maximum(np.abs(ref_xy), 1.0)
    print("max |pos_xy - ref| =", err_xy.max(), " max |z - ref| =", err_z.max(), " final |y| max =", np.abs(ref_xy[-1, :, 1]).max())
    assert float(np.max(err_xy / scale)) < 1e-5 and float(err_z.max()) < 1e-5
    env.close()


def test_s1_known_answers_on_gpu():
    fx = R.load("s1_reference_test_scenario")
    res, env = R.replay(hip_env, fx, "s1", {"vel": HIP_GETTERS["vel"], "yaw": HIP_GETTERS["yaw"]})
    n = int(np.argmax(res["done"][:, 0])) + 1
    assert n == 358 and res["yaw"][n - 1, 0] == -426.0
    assert abs(float(np.sum(res["reward"][:n, 0].astype(np.float64))) - 226.97054830007255) < 1e-5 * 226.97
    env.close()


def test_g1_physenv_gym_loop():
    """BASELINE config 1 plumbing: PhysEnv(get_default) step/reset loop, RLlib-shaped single actions."""
    from q1physrl_amd import env as E
    import q1physrl_amd
    fx = R.load("g1_physenv_default")
    np.random.seed(int(fx["seed"]))
    pe = q1physrl_amd.make('Q1PhysEnv-v0')
    assert isinstance(pe, E.PhysEnv)
    reset_obs = [pe.reset()]
    for t in range(fx["actions"].shape[0]):
        a = fx["actions"][t]
        o, r, d, info = pe.step([int(a[0]), int(a[1]), int(a[2]), int(a[3]), np.array([a[4]], dtype=np.float32)])
        assert o.shape == (6,) and o.dtype == np.float64 and isinstance(r, np.float32)
        assert np.max(rel_err(o, fx["obs"][t])) <= REL_TOL and abs(float(r) - float(fx["reward"][t])) <= REL_TOL * max(1, abs(float(r)))
        assert bool(d) == bool(fx["done"][t]) and bool(info["zero_start"]) == bool(fx["zero_start"][t])
        if d:
            reset_obs.append(pe.reset())
    assert np.max(rel_err(np.stack(reset_obs), fx["reset_obs"])) <= REL_TOL
    pe.close()


def test_standalone_decoder_and_phys_apply_micro_vectors():
    """G5: ActionDecoder.map on its own (mkdemo-style, incl. time exactly at the key_press_delay boundary)
    and stateless phys.apply on random states."""
    from q1physrl_amd import env as E, phys as P
    fx = R.load("g5_micro")
    cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": 6})
    dec = E.ActionDecoder(cfg)
    dec.vector_reset(fx["dec_yaw0"])
    for t in range(fx["dec_actions"].shape[0]):
        y, sm, fm, j = dec.map(fx["dec_actions"][t], np.zeros(6, np.float32), np.full(6, fx["dec_time_remaining"][t]))
        assert np.array_equal(sm, fx["dec_out_smove"][t]) and np.array_equal(fm, fx["dec_out_fmove"][t])
        assert np.array_equal(j, fx["dec_out_jump"][t]) and sm.dtype == np.int64 and j.dtype == np.bool_
        assert np.array_equal(y, fx["dec_out_yaw"][t])                     # pure float64 mul/div/add: exact
        assert np.array_equal(dec._last_key_press_time, fx["dec_out_lkpt"][t])
        assert np.array_equal(dec._last_keys.astype(np.uint8), fx["dec_out_lk"][t])
    ins = P.Inputs(**{k: fx["ap_in_" + k] for k in ("yaw", "pitch", "roll", "fmove", "smove", "button2", "time_delta")})
    ps = P.PlayerState(**{k: fx["ap_ps_" + k] for k in ("z_pos", "vel", "on_ground", "jump_released")})
    out = P.apply(ins, ps)
    assert out.vel.dtype == np.float32 and np.max(rel_err(out.vel, fx["ap_out_vel"])) <= REL_TOL
    assert bit_identical_fraction(out.vel, fx["ap_out_vel"]) >= 0.99
    assert np.array_equal(out.z_pos, fx["ap_out_z_pos"])                   # no transcendental on the z path: exact
    assert np.array_equal(out.on_ground, fx["ap_out_on_ground"]) and np.array_equal(out.jump_released, fx["ap_out_jump_released"])
    # z physics edge cases (exactly on the floor, landing, jump while released/not released)
    n = fx["z_in_pos"].shape[0]
    ins = P.Inputs(yaw=np.zeros(n), pitch=np.zeros(n), roll=np.zeros(n), fmove=np.zeros(n), smove=np.zeros(n),
                   button2=fx["z_in_jump"], time_delta=fx["z_dt"])
    vel = np.zeros((n, 3), np.float32)
    vel[:, 2] = fx["z_in_vel"]
    out = P.apply(ins, P.PlayerState(fx["z_in_pos"], vel, fx["z_in_on_ground"], fx["z_in_jump_released"]))
    assert np.array_equal(out.z_pos, fx["z_out_pos"]) and np.array_equal(out.vel[:, 2], fx["z_out_vel"])
    assert np.array_equal(out.on_ground, fx["z_out_on_ground"]) and np.array_equal(out.jump_released, fx["z_out_jump_released"])


@pytest.mark.parametrize("n,ticks,seed", [(1, 50, 1), (63, 100, 2), (257, 200, 3), (4096, 720, 4)])
def test_against_oracle_random(n, ticks, seed):
    """Fresh seeded inputs (not in the fixtures), ragged sizes (1, 63, 257: partial waves / partial blocks),
    full Config with random starts; the oracle runs beside the GPU on identical NumPy RNG state."""
    from q1physrl_amd import env as E
    kw = dict(O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.25, time_limit=3.0).__dict__)
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    ora = O.OracleVectorEnv(dict(kw))
    np.random.seed(seed)
    hip = E.VectorPhysEnv(dict(kw))
    keys = rng.random((n, 4)) < 0.5
    worst = 0.0
    for t in range(ticks):
        keys ^= rng.random((n, 4)) < 0.1
        yaw = rng.uniform(-10.08, 10.08, n).astype(np.float32)
        a = np.concatenate([keys.astype(np.float64), yaw[:, None].astype(np.float64)], axis=1)
        o1, r1, d1, z1 = ora.vector_step(a)
        o2, r2, d2, i2 = hip.vector_step(a)
        assert np.array_equal(d1, d2) and np.array_equal(z1, np.array([i["zero_start"] for i in i2]) if n <= 64 else i2._zs)
        worst = max(worst, float(np.max(rel_err(o2, o1))), float(np.max(rel_err(r2, r1))))
        assert worst <= REL_TOL, (t, worst)
        for i in np.nonzero(d1)[0]:
            st = np.random.get_state()
            a1 = ora.reset_at(int(i))
            np.random.set_state(st)
            a2 = hip.reset_at(int(i))
            assert np.max(rel_err(a2, a1)) <= REL_TOL
    ps = hip.player_state
    assert np.array_equal(ps.on_ground, ora.st["on_ground"])
    assert bit_identical_fraction(ps.vel, ora.st["vel"]) >= 0.999
    hip.close()


def _random_config(rng, n):
    kw = dict(O.OracleConfig.get_default(num_envs=n).__dict__)
    kw.update(time_delta=float(rng.choice([1.0 / 72, 0.013888888888888, 0.02, 1.0 / 125, 0.014])),
              time_limit=float(rng.choice([1.0, 2.5, 10.0])), key_press_delay=float(rng.choice([0.0, 0.05, 0.3])),
              smooth_keys=bool(rng.integers(2)), fmove_max=float(rng.choice([0.0, 0.4, 127.0, 400.0, 800.0, 1200.0])),
              smove_max=float(rng.choice([0.0, 350.0, 700.0, 1060.0])), zero_start_prob=float(rng.choice([0.0, 0.3, 1.0])),
              max_initial_speed=float(rng.choice([0.0, 320.0, 700.0])), hover=bool(rng.random() < 0.15),
              speed_reward=bool(rng.random() < 0.3), action_range=float(rng.choice([10.0, 10.079999923706055, 3.0])))
    mode = rng.integers(4)
    if mode == 1:
        kw.update(auto_jump=True)
    elif mode == 2:
        kw.update(allow_jump=False)
    ymode = rng.integers(4)
    if ymode == 1:
        kw.update(discrete_yaw_steps=int(rng.choice([1, 4, 10])))
    elif ymode == 2:
        kw.update(allow_yaw=False)
    return kw


@pytest.mark.parametrize("seed", range(24))
def test_random_configs_against_oracle(seed):
    """Config fuzz (round 3: the tick's selects, square root and sin / cos were re-written): random Config - time step, key delay,
    smoothing, move maxima incl. 0 and sub-unit ones (the wish-less stand-in), jump / yaw modes, hover, both rewards - 192 envs (three
    waves), 150 ticks of sticky random keys with stretches of NO key pressed and of all keys pressed, reset_at on done.  Ints exact,
    floats within REL_TOL with >= 99.9 % bit-identical (in practice all)."""
    from q1physrl_amd import env as E
    rng = np.random.default_rng(1000 + seed)
    n = 192
    kw = _random_config(rng, n)
    np.random.seed(seed)
    ora = O.OracleVectorEnv(dict(kw))
    np.random.seed(seed)
    hip = E.VectorPhysEnv(dict(kw))
    cfg = O.OracleConfig(**kw)
    nk = cfg.num_keys
    keys = rng.random((n, nk)) < 0.5
    for t in range(150):
        keys ^= rng.random((n, nk)) < 0.15
        if 40 <= t < 55:
            keys[: n // 2] = False                                  # nobody in the first half presses anything: no wish at all
        if 90 <= t < 100:
            keys[n // 2:] = True
        cols = [keys.astype(np.float64)]
        if cfg.allow_yaw:
            if cfg.discrete_yaw_steps >= 0:
                cols.append(rng.integers(0, 2 * cfg.discrete_yaw_steps + 1, (n, 1)).astype(np.float64))
            else:
                cols.append(rng.uniform(-cfg.action_range, cfg.action_range, (n, 1)).astype(np.float32).astype(np.float64))
        a = np.concatenate(cols, axis=1)
        o1, r1, d1, z1 = ora.vector_step(a)
        o2, r2, d2, i2 = hip.vector_step(a)
        assert np.array_equal(d1, d2), (kw, t)
        assert float(np.max(rel_err(o2, o1))) <= REL_TOL and float(np.max(rel_err(r2, r1))) <= REL_TOL, (kw, t)
        assert bit_identical_fraction(np.asarray(o2, np.float64), np.asarray(o1, np.float64)) >= 0.999, (kw, t)
        for i in np.nonzero(d1)[0]:
            st = np.random.get_state()
            a1 = ora.reset_at(int(i))
            np.random.set_state(st)
            a2 = hip.reset_at(int(i))
            assert np.max(rel_err(a2, a1)) <= REL_TOL
    ps = hip.player_state
    assert np.array_equal(ps.on_ground, ora.st["on_ground"]) and np.array_equal(ps.jump_released, ora.st["jump_released"])
    assert bit_identical_fraction(ps.vel, ora.st["vel"]) >= 0.999 and bit_identical_fraction(ps.z_pos, ora.st["z_pos"].astype(ps.z_pos.dtype)) >= 0.999
    hip.close()


def test_state_roundtrip_and_error_behaviour():
    from q1physrl_amd import env as E, _lib
    cfg = dict(E.Config.get_default().__dict__, num_envs=8)
    e = E.VectorPhysEnv(cfg)
    st = e.get_state()
    st["vel_x"][:] = np.arange(8, dtype=np.float32)
    e.set_state(**st)
    assert np.array_equal(e.get_state()["vel_x"], np.arange(8, dtype=np.float32))
    assert np.array_equal(e._get_obs(), e._dev.observe_host()) and e._get_obs_at(3).shape == (6,)
    e.reset_at(-1)                                      # negative index wraps like the reference's NumPy indexing
    with pytest.raises(ValueError):
        e.vector_step(np.zeros((7, 5)))                 # wrong batch
    with pytest.raises(TypeError):
        E.VectorPhysEnv(dict(cfg, bogus=1))             # unknown Config key -> TypeError like the reference
    with pytest.raises(AssertionError, match="num_envs must be None"):
        E.PhysEnv(E.Config(**cfg))
    with pytest.raises(_lib.Q1EnvError):
        E.VectorPhysEnv(dict(cfg, num_envs=0))
    e.close()


def test_exact_division_shortcuts_selftest():
    """The kernels replace IEEE divisions by exact FMA sequences (q1env_device.hpp div_const / div_shared and the
    float32 obs columns).  On-device comparison with the hardware division on 2^24 random operands per seed and the
    constants 180, 90, 100, 200, time_limit, action_range: zero mismatches allowed."""
    import ctypes as C
    from q1physrl_amd import _lib
    lib = _lib.load()
    for seed, c0, c1 in ((1, 10.0, 10.079999923706055), (2, 5.0, 10.0), (3, 0.5, 7.0), (4, 3.0, 5.0)):
        out = (C.c_uint64 * 4)()
        # 2^25 threads: the first 2^25 also run the EXHAUSTIVE check of the two float32 observation-column shortcuts (every
        # trunc(v / 16) = m and rint(8 z) = j below 2^24 in magnitude); constants that pass the one-step bound are tested in both forms
        _lib.check(lib.q1env_selftest_division(0, 1 << 25, seed, c0, c1, out))
        assert list(out) == [0, 0, 0, 0], (seed, list(out))


def test_inline_sincos_and_sqrt_selftest():
    """The tick's in-line sin / cos of the yaw (q1env_device.hpp sincos_yaw: exact FMA reduction + fdlibm kernels on hi + lo) and its
    scaling-free square root.  2^22 yaw values over +-7500 degrees (an episode stays within ~+-7600) plus a sweep up to the 2^20-radian
    hand-over to the library: against THIS host's libm (what NumPy calls in the reference) at most 1 ulp and < 5 % of values differ -
    the same class as the device library's own sincos, which it is also compared with; sqrt_normal == sqrt on 2^22 operands."""
    import ctypes as C
    from q1physrl_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    n = 1 << 22
    for span, seed in ((7500.0, 11), (5.9e7, 12)):
        yaw = rng.uniform(-span, span, n)
        yaw[:8] = [0.0, 90.0, -90.0, 180.0, 270.0, 360.0, 45.0, 1e-300]
        sn, cs = np.empty(n), np.empty(n)
        out = (C.c_uint64 * 4)()
        _lib.check(lib.q1env_selftest_trig(0, n, yaw.ctypes.data, sn.ctypes.data, cs.ctypes.data, seed, out))
        rad = (yaw * np.pi) / 180.0
        ds = np.abs(sn.view(np.int64) - np.sin(rad).view(np.int64))
        dc = np.abs(cs.view(np.int64) - np.cos(rad).view(np.int64))
        assert ds.max() <= 1 and dc.max() <= 1, (span, int(ds.max()), int(dc.max()))
        frac = ((ds != 0).sum() + (dc != 0).sum()) / (2.0 * n)
        assert frac < 0.05, (span, frac)
        assert out[1] <= 2 and out[0] < 0.1 * 2 * n, list(out)           # vs the device library: <= 2 ulp apart, < 10 % of values
        assert out[2] == 0, list(out)                                     # the square root is bit-identical to the compiler's
    # beyond the hand-over the library's path runs: identical to it by construction
    yaw = rng.uniform(6.1e7, 1e12, 4096) * rng.choice([-1.0, 1.0], 4096)
    sn, cs = np.empty(4096), np.empty(4096)
    out = (C.c_uint64 * 4)()
    _lib.check(lib.q1env_selftest_trig(0, 4096, yaw.ctypes.data, sn.ctypes.data, cs.ctypes.data, 13, out))
    assert out[0] == 0 and out[2] == 0, list(out)


def test_extreme_yaw_and_speed_states_match_oracle():
    """States far outside what an episode reaches (yaw up to 1e9 degrees: the large-argument path of the device sincos;
    speeds of 1e4; tiny and negative-zero-free velocities) injected through set_state: 50 ticks against the oracle."""
    from q1physrl_amd import env as E
    n = 512
    kw = dict(O.OracleConfig.get_default(num_envs=n, zero_start_prob=1.0).__dict__)
    np.random.seed(0)
    ora = O.OracleVectorEnv(dict(kw))
    np.random.seed(0)
    hip = E.VectorPhysEnv(dict(kw))
    rng = np.random.default_rng(9)
    yaw = rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(0, 9, n)
    vel = (rng.uniform(-1, 1, (n, 3)) * 10.0 ** rng.uniform(-6, 4, (n, 1))).astype(np.float32)
    vel[:, 2] = rng.uniform(-300, 300, n).astype(np.float32)
    ora.yaw = yaw.copy(); ora.dec["yaw"] = yaw.copy(); ora.st["vel"] = vel.copy()
    hip.set_state(yaw=yaw, vel_x=vel[:, 0], vel_y=vel[:, 1], vel_z=vel[:, 2])
    worst, same = 0.0, []
    for t in range(50):
        a = np.concatenate([(rng.random((n, 4)) < 0.5).astype(np.float64), rng.uniform(-10, 10, (n, 1)).astype(np.float32).astype(np.float64)], axis=1)
        o1, r1, d1, _ = ora.vector_step(a)
        o2, r2, d2, _ = hip.vector_step(a)
        worst = max(worst, float(np.max(rel_err(o2, o1))), float(np.max(rel_err(r2, r1))))
        same.append(bit_identical_fraction(np.ascontiguousarray(o2), np.ascontiguousarray(o1)))
    print("extreme states: worst rel err", worst, "min bit-identical fraction", min(same))
    assert worst <= REL_TOL and min(same) >= 0.999
    hip.close()


def test_reset_many_equals_sequential_reset_at():
    """reset_many(indices) == [reset_at(i) for i in indices]: same NumPy-stream consumption, same observations, same state
    (including the repeated-index and negative-index cases), checked against the oracle's reset_at as well."""
    from q1physrl_amd import env as E
    n = 300
    kw = dict(O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.4, time_limit=3.0).__dict__)
    envs = []
    for make in (E.VectorPhysEnv, E.VectorPhysEnv, O.OracleVectorEnv):
        np.random.seed(21)                              # identical initial episodes
        envs.append(make(dict(kw)))
    a, b, ora = envs
    for idx in ([5, 299, 0, 17, 64, 63, 128], list(range(0, 300, 3)), [7, 7, 9], [-1, 3], []):
        np.random.seed(33)
        oa = np.stack([a.reset_at(i) for i in idx]) if idx else np.empty((0, 6))
        end_a = np.random.get_state()[1].copy(), np.random.get_state()[2]
        np.random.seed(33)
        ob = b.reset_many(idx)
        end_b = np.random.get_state()[1].copy(), np.random.get_state()[2]
        np.random.seed(33)
        oo = np.stack([ora.reset_at(i) for i in idx]) if idx else np.empty((0, 6))
        assert ob.shape == (len(idx), 6) and np.array_equal(oa, ob)
        assert np.array_equal(end_a[0], end_b[0]) and end_a[1] == end_b[1]          # the global stream advanced identically
        assert np.max(rel_err(ob, oo), initial=0.0) <= REL_TOL
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    a.close(); b.close()


@pytest.mark.parametrize("behaviour", ["rllib", "skip_last", "descending", "peek", "foreign_index"])
def test_speculative_resets_equal_the_plain_protocol(behaviour):
    """VectorPhysEnv(speculative_resets=True) resets the whole done set at the first reset_at of a tick and answers the later
    calls from that batch.  Whatever the caller does - RLlib's ascending sweep, leaving a finished env un-reset, descending
    order, reading state between the calls, resetting an env that was not done - observations, rewards, dones, the final state
    and the position of the global NumPy stream must equal the plain one-call-per-env protocol."""
    from q1physrl_amd import env as E
    n, ticks = 257, 160
    kw = dict(O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.3, time_limit=0.25).__dict__)
    envs = []
    for spec in (False, True):
        np.random.seed(5)
        envs.append(E.VectorPhysEnv(dict(kw), speculative_resets=spec))
    plain, fast = envs
    rng = np.random.default_rng(1)
    states = []
    for e in envs:
        np.random.seed(6)
        rng = np.random.default_rng(1)
        log = []
        for t in range(ticks):
            a = np.concatenate([(rng.random((n, 4)) < 0.5).astype(np.float64), rng.uniform(-10, 10, (n, 1))], axis=1)
            obs, rew, done, _ = e.vector_step(a)
            idx = np.flatnonzero(done)
            if behaviour == "skip_last":
                idx = idx[:-1]
            elif behaviour == "descending":
                idx = idx[::-1]
            elif behaviour == "foreign_index" and idx.size > 2:
                idx = np.concatenate([idx[:1], [(int(idx[0]) + 1) % n if (int(idx[0]) + 1) % n not in idx else int(idx[0])], idx[1:]])
            fresh = []
            for j, i in enumerate(idx):
                fresh.append(e.reset_at(int(i)))
                if behaviour == "peek" and j == 0:
                    log.append(e._time_remaining.copy())
            log.append((obs.copy(), rew.copy(), done.copy(), np.array(fresh)))
        states.append((log, e.get_state(), np.random.get_state()[1].copy(), np.random.get_state()[2], dict(e.speculation_stats)))
        e.close()
    (la, sa, ra, pa, st_a), (lb, sb, rb, pb, st_b) = states
    assert st_a["batches"] == 0
    if behaviour == "descending":                 # the first call is not the smallest finished index: speculation never starts
        assert st_b["batches"] == 0
    else:                                         # the fast path really ran (and rolled back whenever the caller deviated)
        assert st_b["batches"] > 20 and st_b["claimed"] > 100
        assert (st_b["rolled_back"] > 0) == (behaviour != "rllib")
    assert len(la) == len(lb)
    for x, y in zip(la, lb):
        if isinstance(x, tuple):
            for u, v in zip(x, y):
                assert np.array_equal(u, v)
        else:
            assert np.array_equal(x, y)
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(ra, rb) and pa == pb


# ---- round-2 vectors (oracle/gen_golden_r2.py) -------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g3_legacy_promotion_params_yml_400", "g4_legacy_promotion_dt014_400"])
def test_legacy_numpy_promotion_traces(name):
    """numpy_promotion="legacy": env.py:230 as the NumPy 1.18.2 of the reference's requirements.txt evaluates it (float64
    product) - traces generated from the reference with that promotion; and the default ("nep50" under this NumPy) must NOT
    reproduce the dt = 0.014 trace bit for bit (the flag is live)."""
    from q1physrl_amd import env as E
    fx = R.load(name)
    tag = "g3_x" if name.startswith("g3") else "g4_x"
    res, env = R.replay(lambda kw: E.VectorPhysEnv(kw, numpy_promotion="legacy"), fx, tag, HIP_GETTERS)
    compare(res, fx, name)
    assert np.max(rel_err(res["reset_obs"], fx["reset_obs"])) <= REL_TOL
    env.close()
    if "dt014" in name:
        res2, env2 = R.replay(lambda kw: E.VectorPhysEnv(kw, numpy_promotion="nep50"), fx, tag, {"yaw": HIP_GETTERS["yaw"]})
        assert bit_identical_fraction(res2["yaw"], fx["yaw"]) < 0.5
        env2.close()


def test_phys_apply_float64_velocity_and_general_angles():
    """phys.apply with a float64 PlayerState.vel (what PlayerState.from_df yields, phys.py:168-170) and non-zero pitch / roll:
    the dtype of vel selects the arithmetic, as in the reference; output dtype follows the input."""
    from q1physrl_amd import phys as P
    fx = R.load("g5b_apply_f64vel")
    ins = P.Inputs(**{k: fx["in_" + k] for k in ("yaw", "pitch", "roll", "fmove", "smove", "button2", "time_delta")})
    ps = P.PlayerState(**{k: fx["ps_" + k] for k in ("z_pos", "vel", "on_ground", "jump_released")})
    out = P.apply(ins, ps)
    assert out.vel.dtype == np.float64 and out.z_pos.dtype == np.float64
    assert np.max(rel_err(out.vel, fx["out_vel"])) <= 1e-12                 # float64 end to end: only sincos' last ulp can differ
    assert bit_identical_fraction(out.vel, fx["out_vel"]) >= 0.9
    assert np.max(rel_err(out.z_pos, fx["out_z_pos"])) <= 1e-15 and bit_identical_fraction(out.z_pos, fx["out_z_pos"]) == 1.0
    assert np.array_equal(out.on_ground, fx["out_on_ground"]) and np.array_equal(out.jump_released, fx["out_jump_released"])
    # many calls in a row reuse the cached per-device scratch context (analyse.py's hypothetical_delta_speeds pattern)
    for _ in range(50):
        again = P.apply(ins, ps)
    assert np.array_equal(again.vel, out.vel) and np.array_equal(again.z_pos, out.z_pos)
    # a float32 call after a float64 one (same scratch context, different layout)
    g5 = R.load("g5_micro")
    out32 = P.apply(P.Inputs(**{k: g5["ap_in_" + k] for k in ("yaw", "pitch", "roll", "fmove", "smove", "button2", "time_delta")}),
                    P.PlayerState(**{k: g5["ap_ps_" + k] for k in ("z_pos", "vel", "on_ground", "jump_released")}))
    assert out32.vel.dtype == np.float32 and np.max(rel_err(out32.vel, g5["ap_out_vel"])) <= REL_TOL
    assert np.array_equal(out32.z_pos, g5["ap_out_z_pos"])


def test_g6_device_rng_reset_against_reference_draws():
    """two-sample Kolmogorov-Smirnov of the DEVICE reset (q1env_reset_philox) against 100 000 resets drawn by
    the reference itself (tests/golden/g6_reset_draws.npz: yaw, time_remaining, initial speed, move angle, zero-start rate)."""
    import json
    from scipy import stats
    from q1physrl_amd.env import Config
    from q1physrl_amd.tensor_env import TensorVectorEnv
    fx = R.load("g6_reset_draws")
    kw = json.loads(str(fx["config_json"]))
    kw["initial_yaw_range"] = tuple(kw["initial_yaw_range"])
    n = 100_000
    env = TensorVectorEnv(Config(num_envs=n, **kw), device=0, seed=2024)
    env.reset()
    st = env.get_state()
    env.close()
    zs = (st["flags"] & 4) != 0
    ref_zs = fx["zero_start"]
    # zero-start rate: binomial(1e5, 0.01) has sigma 31.5; both samples within 5 sigma of 1000 and of each other
    assert abs(int(zs.sum()) - 1000) < 160 and abs(int(zs.sum()) - int(ref_zs.sum())) < 230
    speed = np.hypot(st["vel_x"].astype(np.float64), st["vel_y"].astype(np.float64))
    angle = np.mod(np.arctan2(st["vel_y"].astype(np.float64), st["vel_x"].astype(np.float64)), 2 * np.pi)
    for name, mine, ref in (("yaw", st["yaw"][~zs], fx["yaw"][~ref_zs]), ("time_remaining", st["time_remaining"][~zs], fx["time_remaining"][~ref_zs]),
                            ("speed", speed[~zs], fx["speed"][~ref_zs]), ("angle", angle[~zs], fx["angle"][~ref_zs])):
        d, p = stats.ks_2samp(mine.astype(np.float64), ref.astype(np.float64))
        assert p > 1e-3 and d < 0.01, (name, d, p)
        lo, hi = {"yaw": (0.0, 360.0), "time_remaining": (1.0, 10.0), "speed": (0.999, 700.0), "angle": (0.999, 2 * np.pi + 1e-6)}[name]
        assert lo <= mine.min() and mine.max() <= hi and lo <= ref.min() and ref.max() <= hi, name      # the supports, quirk included
    # zero starts are exact states, not distributions
    assert np.all(st["yaw"][zs] == 90.0) and np.all(st["time_remaining"][zs] == 10.0) and np.all(speed[zs] == 0.0)
    assert np.all(st["vel_z"] == -12.0) and np.all(st["z_pos"] == np.float64(np.float32(32.843201)))


def test_speculative_rollback_leaves_a_foreign_rng_user_alone():
    """the speculative batch's rollback rewinds the GLOBAL NumPy stream.  If other code drew from np.random between the
    reset_at calls of a tick, rewinding would replay its draws - so the rollback then keeps the stream where it is (RuntimeWarning,
    counted in speculation_stats) while still restoring the device state and re-applying the claimed resets."""
    import warnings
    from q1physrl_amd import env as E
    n = 257
    kw = dict(O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.3, time_limit=0.25).__dict__)
    np.random.seed(5)
    e = E.VectorPhysEnv(dict(kw), speculative_resets=True)
    rng = np.random.default_rng(1)
    hit = 0
    for t in range(60):
        a = np.concatenate([(rng.random((n, 4)) < 0.5).astype(np.float64), rng.uniform(-10, 10, (n, 1))], axis=1)
        obs, rew, done, _ = e.vector_step(a)
        idx = np.flatnonzero(done)
        if idx.size < 3:
            for i in idx:
                e.reset_at(int(i))
            continue
        first = e.reset_at(int(idx[0]))                 # starts the speculative batch (draws for ALL finished envs)
        foreign = np.random.random()                    # somebody else uses the global stream
        after_foreign = np.random.get_state()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            st = e.get_state()                          # a state read settles the batch: idx[1:] were not claimed -> rollback
        assert any(issubclass(x.category, RuntimeWarning) and "global NumPy RNG" in str(x.message) for x in w)
        now = np.random.get_state()
        assert now[2] == after_foreign[2] and np.array_equal(now[1], after_foreign[1])      # the stream was NOT rewound
        assert 0.0 <= foreign < 1.0
        # device state: env idx[0] is reset (its obs is what reset_at returned), the unclaimed ones are still finished
        assert st["time_remaining"][idx[0]] > 0 and np.all(st["time_remaining"][idx[1:]] < 0)
        assert np.allclose(e._get_obs_at(int(idx[0])), first, rtol=0, atol=0)
        for i in idx[1:]:
            e.reset_at(int(i))
        hit += 1
    assert hit > 5 and e.speculation_stats.get("foreign_rng_use", 0) == hit
    e.close()


def test_large_batch_host_arrays_are_page_locked_and_owned_by_the_caller():
    """Above the packed-staging threshold (16 384 envs) the NumPy arrays vector_step / vector_reset / get_state return live in
    page-locked blocks from a pool (direct DMA instead of staged pageable copies).  They must still behave like the reference's
    fresh arrays: later calls never overwrite them, a block returns to the pool only when the last array viewing it is collected,
    and the values equal the oracle's."""
    import gc
    from q1physrl_amd import _lib, env as E
    n = 20_000
    kw = dict(O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.5).__dict__)
    np.random.seed(3)
    ora = O.OracleVectorEnv(dict(kw))
    np.random.seed(3)
    env = E.VectorPhysEnv(dict(kw))
    rng = np.random.default_rng(0)
    kept, want = [], []
    for t in range(6):
        a = np.concatenate([(rng.random((n, 4)) < 0.5).astype(np.float64), rng.uniform(-10, 10, (n, 1)).astype(np.float32).astype(np.float64)], axis=1)
        o1, r1, d1, _ = ora.vector_step(a)
        o2, r2, d2, _ = env.vector_step(a)
        kept.append((o2, r2, d2))
        want.append((o1.copy(), r1.copy(), d1.copy()))
    for (o2, r2, d2), (o1, r1, d1) in zip(kept, want):              # every tick's arrays are still what they were when returned
        assert np.array_equal(o2, o1) and np.array_equal(r2, r1) and np.array_equal(d2, d1)
        assert o2.dtype == np.float64 and r2.dtype == np.float32 and d2.dtype == np.bool_ and o2.flags["C_CONTIGUOUS"]
    base = kept[0][0]
    while isinstance(base, np.ndarray) and base.base is not None:
        base = base.base
    assert isinstance(base, _lib._PinnedBlock)                         # page-locked, pool-owned
    view = kept[0][0][5:10]                                           # a view keeps its block alive after the array itself is gone
    pool = _lib.pinned_pool()
    cached_before = pool._cached
    del kept, o2, r2, d2, base
    gc.collect()
    assert pool._cached > cached_before                               # the other blocks went back to the pool ...
    assert np.array_equal(view, want[0][0][5:10])                     # ... the viewed one is intact
    o3, _, _, _ = env.vector_step(a)                                  # and recycled blocks serve the next call
    assert np.array_equal(view, want[0][0][5:10]) and o3.shape == (n, 6)
    env.close()


def test_library_and_torch_share_one_hip_runtime_whatever_the_import_order():
    """`import q1physrl_amd.env`, step an env, THEN `import torch`: PyTorch-ROCm bundles its own libamdhip64.so; two runtimes in one
    process leave torch.cuda without devices.  The binding loads torch's copy first when there is one, so both orders work."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys
import numpy as np
from q1physrl_amd import env as E
assert "torch" not in sys.modules
e = E.VectorPhysEnv(dict(E.Config.get_default().__dict__, num_envs=256))
o, r, d, _ = e.vector_step(np.zeros((256, 5)))
import torch
assert torch.cuda.is_available() and torch.cuda.device_count() >= 1
x = torch.arange(8, device="cuda").float().sum().item()
assert x == 28.0
from q1physrl_amd.tensor_env import TensorVectorEnv
t = TensorVectorEnv(dict(E.Config.get_default().__dict__, num_envs=256))
t.reset()
o2, r2, d2 = t.step_tensor(torch.zeros((256, 5), device="cuda"))
torch.cuda.synchronize()
maps = open("/proc/self/maps").read()
libs = sorted({ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln})
assert len(libs) == 1, libs
print("OK", libs[0])
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
