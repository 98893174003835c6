"""CPU: pin the oracle (oracle/np_oracle.py) BIT-FOR-BIT against vectors produced by the reference
itself (tests/golden/*.npz, made by oracle/gen_golden.py).  This is what makes the oracle a valid
checker for the HIP path.  Bit-exact comparisons: same NumPy build generated the vectors."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests import _replay as R

GETTERS = {
    "vel": lambda e: e.st["vel"], "z_pos": lambda e: e.st["z_pos"], "on_ground": lambda e: e.st["on_ground"],
    "jump_released": lambda e: e.st["jump_released"], "yaw": lambda e: e.yaw,
    "time_remaining": lambda e: e.t_rem, "last_key_press_time": lambda e: e.dec["last_press"],
    "last_keys": lambda e: e.dec["last_keys"].astype(np.uint8),
    "smove": lambda e: e.last_cmd[0], "fmove": lambda e: e.last_cmd[1], "jump": lambda e: e.last_cmd[2],
}


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind == "f" or b.dtype.kind == "f":
        assert a.dtype == b.dtype, (a.dtype, b.dtype)
        return np.array_equal(a.view(np.uint8), b.view(np.uint8))
    return np.array_equal(a, b)


@pytest.mark.parametrize("name", R.TRACE_FIXTURES)
def test_trace_bit_exact(name):
    fx = R.load(name)
    res, env = R.replay(lambda kw: O.OracleVectorEnv(kw), fx, name, GETTERS)
    assert bits_equal(res["obs0"], fx["obs0"])
    for k in ("obs", "reward", "done", "zero_start", "vel", "z_pos", "on_ground", "jump_released", "yaw",
              "time_remaining", "last_key_press_time", "last_keys", "smove", "fmove", "jump"):
        assert bits_equal(res[k], fx[k]), f"{name}: field {k} differs from the reference"
    assert bits_equal(res["reset_obs"], fx["reset_obs"])
    assert bits_equal(env.st["vel"], fx["final_vel"])
    assert bits_equal(env.yaw, fx["final_yaw"])


def test_s1_known_answers():
    """Numbers recorded in SURVEY.md 8c for the reference's own test scenario (tests/test_integration.py:50-84)."""
    fx = R.load("s1_reference_test_scenario")
    res, _ = R.replay(lambda kw: O.OracleVectorEnv(kw), fx, "s1", {"vel": GETTERS["vel"], "z_pos": GETTERS["z_pos"],
                                                                 "yaw": GETTERS["yaw"], "time_remaining": GETTERS["time_remaining"]})
    n = int(np.argmax(res["done"][:, 0])) + 1
    assert n == 358
    assert float(np.sum(res["reward"][:n, 0].astype(np.float64))) == 226.97054830007255
    assert res["yaw"][n - 1, 0] == -426.0
    assert res["vel"][n - 1, 0].tolist() == [-87.4294662475586, 270.1275939941406, 135.6000213623047]
    assert res["z_pos"][n - 1, 0] == 57.16085076904297
    assert res["time_remaining"][n - 1, 0] == -0.011999999999988329
    assert res["vel"][0, 0].tolist() == [1.8369700935892946e-15, 30.0, -23.200000762939453]
    assert res["reward"][0, 0] == np.float32(0.42000002)


def test_s2_known_answers():
    fx = R.load("s2_constant_action_720")
    res, _ = R.replay(lambda kw: O.OracleVectorEnv(kw), fx, "s2", {"vel": GETTERS["vel"], "z_pos": GETTERS["z_pos"], "yaw": GETTERS["yaw"]})
    assert res["done"][:, 0].tolist() == [False] * 719 + [True]        # done on the 720th step
    assert float(np.sum(res["reward"][:, 0].astype(np.float64))) == -55.83386661252007
    assert res["yaw"][-1, 0] == 804.2857196920389
    assert res["obs0"][0].tolist() == [1.0, 1.0, 0.32875, 0, 0, 0]
    assert res["obs"][0, 0].tolist() == [0.9986111111111111, 1.0110229277730252, 0.325, 0.08, 0.08, -0.08]


def test_g1_physenv_plumbing():
    """G1: gym-style single env loop with reset() on done (config 1 of BASELINE.json)."""
    fx = R.load("g1_physenv_default")
    np.random.seed(int(fx["seed"]))
    env = O.OracleVectorEnv(O.OracleConfig.get_default(num_envs=1))
    reset_obs = [env.vector_reset()[0]]
    for t in range(fx["actions"].shape[0]):
        obs, rew, done, zs = env.vector_step(fx["actions"][t][None, :])
        assert bits_equal(obs[0], fx["obs"][t]), t
        assert rew[0] == fx["reward"][t] and done[0] == fx["done"][t] and zs[0] == fx["zero_start"][t]
        if done[0]:
            reset_obs.append(env.vector_reset()[0])
    assert bits_equal(np.stack(reset_obs), fx["reset_obs"])


def test_g5_micro_vectors():
    fx = R.load("g5_micro")
    c, s = O.basis_from_yaw(fx["av_yaw"])
    assert bits_equal(c, fx["av_out"][:, 0, 0]) and bits_equal(s, fx["av_out"][:, 1, 0])
    assert np.array_equal(s, fx["av_out"][:, 0, 1]) and np.array_equal(-c, fx["av_out"][:, 1, 1])
    fx_, fy_ = O.friction(fx["fr_h_vel"][:, 0], fx["fr_h_vel"][:, 1], np.float64(fx["fr_dt"][0]))
    assert bits_equal(np.stack([fx_, fy_], 1), fx["fr_out"])
    hx, hy = O.horizontal_move(fx["am_yaw"], fx["am_fmove"], fx["am_smove"], fx["am_on_ground"], np.float64(fx["am_dt"][0]),
                               fx["am_h_vel"][:, 0], fx["am_h_vel"][:, 1])
    assert bits_equal(np.stack([hx, hy], 1), fx["am_out"])
    z, vz, og, jr = O.vertical_move(fx["z_in_jump"], np.float64(fx["z_dt"][0]), fx["z_in_pos"], fx["z_in_vel"],
                                    fx["z_in_on_ground"], fx["z_in_jump_released"])
    assert bits_equal(z, fx["z_out_pos"]) and bits_equal(vz, fx["z_out_vel"])
    assert np.array_equal(og, fx["z_out_on_ground"]) and np.array_equal(jr, fx["z_out_jump_released"])
    st = O.phys_apply(fx["ap_in_yaw"], fx["ap_in_fmove"], fx["ap_in_smove"], fx["ap_in_button2"], float(fx["ap_in_time_delta"][0]),
                      {"z_pos": fx["ap_ps_z_pos"], "vel": fx["ap_ps_vel"], "on_ground": fx["ap_ps_on_ground"],
                       "jump_released": fx["ap_ps_jump_released"]})
    assert bits_equal(st["vel"], fx["ap_out_vel"]) and bits_equal(st["z_pos"], fx["ap_out_z_pos"])
    assert np.array_equal(st["on_ground"], fx["ap_out_on_ground"])
    # stand-alone decoder (mkdemo-style)
    cfg = O.OracleConfig.get_default(num_envs=6)
    dec = {"last_press": np.full((6, 4), -0.3), "last_keys": np.zeros((6, 4), bool), "yaw": fx["dec_yaw0"].copy()}
    for t in range(fx["dec_actions"].shape[0]):
        y, sm, fm, j = O.decode(cfg, dec, fx["dec_actions"][t], np.zeros(6, np.float32), np.full(6, fx["dec_time_remaining"][t]))
        assert bits_equal(y, fx["dec_out_yaw"][t]) and np.array_equal(sm, fx["dec_out_smove"][t])
        assert np.array_equal(fm, fx["dec_out_fmove"][t]) and np.array_equal(j, fx["dec_out_jump"][t])
        assert bits_equal(dec["last_press"], fx["dec_out_lkpt"][t])
        assert np.array_equal(dec["last_keys"].astype(np.uint8), fx["dec_out_lk"][t])


# ---- round-2 vectors (oracle/gen_golden_r2.py) -------------------------------------------------------------------------------
LEGACY_FIXTURES = ["g3_legacy_promotion_params_yml_400", "g4_legacy_promotion_dt014_400"]


@pytest.mark.parametrize("name", LEGACY_FIXTURES)
def test_legacy_promotion_traces_bit_exact(name):
    """env.py:230 as NumPy < 2 evaluates it (float64 product): traces generated from the reference with that promotion."""
    fx = R.load(name)
    res, env = R.replay(lambda kw: O.OracleVectorEnv(kw, legacy_promotion=True), fx, name.replace("g3_legacy", "g3_").replace("g4_legacy", "g4_"), GETTERS)
    for k in ("obs", "reward", "done", "vel", "z_pos", "on_ground", "yaw", "time_remaining", "last_key_press_time", "last_keys",
              "smove", "fmove", "jump"):
        assert bits_equal(res[k], fx[k]), f"{name}: field {k} differs from the reference"
    assert bits_equal(res["reset_obs"], fx["reset_obs"]) and bits_equal(env.yaw, fx["final_yaw"])
    # ... and the fixture really pins the promotion: the NEP 50 product gives a different yaw track
    res2, _ = R.replay(lambda kw: O.OracleVectorEnv(kw), fx, "g3_x", {"yaw": GETTERS["yaw"]})
    assert not bits_equal(res2["yaw"], fx["yaw"])
    rel = np.max(np.abs(res2["yaw"] - fx["yaw"]) / np.maximum(np.abs(fx["yaw"]), 1.0))
    assert rel < (1e-6 if "dt014" in name else 1e-11)          # 7.6e-9 relative per tick for dt = 0.014, 6e-14 for params.yml's dt


def test_g5b_apply_with_float64_velocity():
    """phys.apply on PlayerState.from_df / Inputs.from_df data (float64 vel, per-frame dt, pitch / roll): no float32 anywhere."""
    fx = R.load("g5b_apply_f64vel")
    assert fx["ps_vel"].dtype == np.float64 and fx["out_vel"].dtype == np.float64
    z, vel, og, jr = O.phys_apply_general(fx["in_yaw"], fx["in_pitch"], fx["in_roll"], fx["in_fmove"], fx["in_smove"], fx["in_button2"],
                                          fx["in_time_delta"], fx["ps_z_pos"], fx["ps_vel"], fx["ps_on_ground"], fx["ps_jump_released"])
    assert bits_equal(z, fx["out_z_pos"]) and bits_equal(vel, fx["out_vel"])
    assert np.array_equal(og, fx["out_on_ground"]) and np.array_equal(jr, fx["out_jump_released"])
    # the same function on the float32 micro-vector of G5 reproduces the env-path arithmetic
    g5 = R.load("g5_micro")
    z, vel, og, jr = O.phys_apply_general(g5["ap_in_yaw"], g5["ap_in_pitch"], g5["ap_in_roll"], g5["ap_in_fmove"], g5["ap_in_smove"],
                                          g5["ap_in_button2"], g5["ap_in_time_delta"], g5["ap_ps_z_pos"], g5["ap_ps_vel"],
                                          g5["ap_ps_on_ground"], g5["ap_ps_jump_released"])
    assert bits_equal(vel, g5["ap_out_vel"]) and bits_equal(z, g5["ap_out_z_pos"]) and np.array_equal(og, g5["ap_out_on_ground"])


def test_g6_reset_draws_fixture_matches_the_oracle_draw_order():
    """G6: the reference's vector_reset state for 100 000 envs.  The oracle, seeded the same, reproduces it draw for
    draw (pins env.py:432-451 incl. the one-argument uniform quirk); the fixture's ranges are the quirk's."""
    import json
    fx = R.load("g6_reset_draws")
    kw = json.loads(str(fx["config_json"]))
    kw["initial_yaw_range"] = tuple(kw["initial_yaw_range"])
    np.random.seed(int(fx["seed"]))
    env = O.OracleVectorEnv(O.OracleConfig(num_envs=fx["yaw"].shape[0], **kw))
    assert np.array_equal(env.zero_start, fx["zero_start"])
    assert np.array_equal(env.yaw.astype(np.float32), fx["yaw"]) and np.array_equal(env.t_rem.astype(np.float32), fx["time_remaining"])
    v = env.st["vel"]
    assert np.array_equal(np.hypot(v[:, 0].astype(np.float64), v[:, 1].astype(np.float64)).astype(np.float32), fx["speed"])
    nz = ~fx["zero_start"]
    assert fx["time_remaining"][nz].min() > 1.0 and fx["time_remaining"][nz].max() <= 10.0      # uniform(10) == uniform(10, 1)
    assert fx["speed"][nz].min() > 0.999 and fx["speed"][nz].max() <= 700.0
    assert fx["angle"][nz].min() > 0.999 and fx["angle"][nz].max() <= 2 * np.pi + 1e-6
    assert abs(fx["zero_start"].mean() - 0.01) < 0.002
