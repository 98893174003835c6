#!/usr/bin/env python3
"""Round-2 golden vectors, again produced by RUNNING THE REFERENCE (read-only, /root/reference) in this container - a
separate script so that oracle/gen_golden.py and the fixtures it owns stay byte-for-byte reproducible.

    python oracle/gen_golden_r2.py          (needs /root/reference)

  g5b_apply_f64vel.npz          phys.apply (phys.py:184-197) driven the way analyse.py does: PlayerState.from_df / Inputs.from_df of
                                a DataFrame (float64 velocity, per-frame host_frametime, non-zero pitch and roll) -> outputs.  Pins
                                the float64-velocity arithmetic (no float32 island anywhere) and the general _angle_vectors.
  g6_reset_draws.npz            SURVEY.md 8c G6: 100 000 envs' worth of VectorPhysEnv.vector_reset (env.py:428-455) state for the
                                run's own env_config (data/params.yml) - yaw, time_remaining, initial speed and move angle as
                                the reference left them in its state (float32) - for two-sample KS tests of the device RNG reset.
  g3_legacy_promotion_*.npz     traces with env.py:230 evaluated as NumPy < 2 does (float64 product): the reference is imported
                                and its module constant _MAX_YAW_SPEED replaced by np.float64(720) - under value-based promotion
                                np.float32(720) * python_float WAS the float64 product, and env.py:230 is the only promotion-
                                sensitive expression on the path.  One trace for params.yml's truncated dt, one
                                for the dataclass default dt = 0.014 (where NEP 50 and legacy differ by 7.6e-9 relative).
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (imports the reference through _refshim; running its main() is NOT triggered)

ref_env, ref_phys = G.ref_env, G.ref_phys

PARAMS_YML = dict(action_range=10, allow_jump=True, allow_yaw=True, auto_jump=False, discrete_yaw_steps=-1, fmove_max=800,
                  smove_max=1060, hover=False, initial_yaw_range=(0, 360), key_press_delay=0.3, max_initial_speed=700,
                  smooth_keys=True, speed_reward=False, time_delta=0.013888888888888, time_limit=10, zero_start_prob=0.01)


def apply_f64():
    import pandas as pd
    rng = np.random.default_rng(20260928)
    m = 192
    og = rng.random(m) < 0.5
    z = np.where(og, 24.03125, rng.uniform(24.04, 90, m))
    vel = rng.uniform(-700, 700, (m, 3))
    vel[og, 2] = 0
    vel[:6, :2] = 0                                   # speed 0 on and off the ground
    vel[6:10, :2] = [[30, 40], [1e-3, 0], [99.99, 0], [100.0000001, 0]]
    pitch = np.where(rng.random(m) < 0.5, 0.0, rng.uniform(-70, 70, m))
    roll = np.where(rng.random(m) < 0.7, 0.0, rng.uniform(-20, 20, m))
    df = pd.DataFrame({"z": z, "velx": vel[:, 0], "vely": vel[:, 1], "velz": vel[:, 2], "onground": og.astype(np.float64),
                       "jumpreleased": (rng.random(m) < 0.8).astype(np.float64),
                       "yaw": rng.uniform(-720, 720, m), "pitch": pitch, "roll": roll,
                       "fmove": rng.choice(np.array([0., 200., 400., 800.]), m), "smove": rng.choice(np.array([0., -350., 350., -1060., 1060.]), m),
                       "button2": (rng.random(m) < 0.5).astype(np.float64),
                       "host_frametime": rng.choice(np.array([1. / 72, 0.014, 0.0138, 0.01, 0.025]), m)})
    ins, ps = ref_phys.Inputs.from_df(df), ref_phys.PlayerState.from_df(df)
    assert ps.vel.dtype == np.float64
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = ref_phys.apply(ins, ps)
    assert out.vel.dtype == np.float64
    d = {"in_" + k: np.asarray(getattr(ins, k)) for k in ("yaw", "pitch", "roll", "fmove", "smove", "button2", "time_delta")}
    d.update({"ps_" + k: np.asarray(getattr(ps, k)) for k in ("z_pos", "vel", "on_ground", "jump_released")})
    d.update({"out_" + k: np.asarray(getattr(out, k)) for k in ("z_pos", "vel", "on_ground", "jump_released")})
    G.save("g5b_apply_f64vel", d)


def reset_draws():
    n, seed = 100_000, 606
    np.random.seed(seed)
    e = ref_env.VectorPhysEnv(ref_env.Config(num_envs=n, **PARAMS_YML))      # __init__ calls vector_reset (env.py:426)
    vel = e.player_state.vel
    zs = np.asarray(e._zero_start)
    d = {"seed": np.int64(seed), "config_json": np.array(G.json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in PARAMS_YML.items()})),
         "zero_start": zs, "yaw": np.asarray(e._yaw, dtype=np.float32), "time_remaining": np.asarray(e._time_remaining, dtype=np.float32),
         "speed": np.hypot(vel[:, 0].astype(np.float64), vel[:, 1].astype(np.float64)).astype(np.float32),
         "angle": np.mod(np.arctan2(vel[:, 1].astype(np.float64), vel[:, 0].astype(np.float64)), 2 * np.pi).astype(np.float32),
         "vel_z": vel[:, 2].copy(), "z_pos": np.asarray(e.player_state.z_pos, dtype=np.float32)}
    print("G6: zero starts", int(zs.sum()), "of", n, "; min t_rem", float(d["time_remaining"].min()), "min speed(non-zero-start)",
          float(d["speed"][~zs].min()))
    G.save("g6_reset_draws", d)


def legacy_promotion():
    saved = ref_env._MAX_YAW_SPEED
    ref_env._MAX_YAW_SPEED = np.float64(720.0)          # NumPy < 2: float32(720) * python float == float64(720) * float
    try:
        rng = np.random.default_rng(77)
        kw = dict(PARAMS_YML, num_envs=16)
        acts = G.persistent_actions(rng, 400, 16, 4, "continuous", 10.0, -1)
        t = G.run_trace(kw, acts, seed=41, reset_on_done=True)
        G.save("g3_legacy_promotion_params_yml_400", t)
        kw = dict(num_envs=12, zero_start_prob=0.5, initial_yaw_range=(0, 360), max_initial_speed=700.)      # dataclass defaults: dt 0.014
        acts = G.persistent_actions(rng, 400, 12, 4, "continuous", float(np.float32(720) * np.float32(0.014)), -1)
        t = G.run_trace(kw, acts, seed=42, reset_on_done=True)
        G.save("g4_legacy_promotion_dt014_400", t)
    finally:
        ref_env._MAX_YAW_SPEED = saved


if __name__ == "__main__":
    apply_f64()
    reset_draws()
    legacy_promotion()
