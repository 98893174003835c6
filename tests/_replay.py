"""Shared replay harness: drive an env object (oracle or HIP-backed drop-in) through a golden
fixture exactly the way oracle/gen_golden.py drove the reference, and collect the same arrays."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRACE_FIXTURES = [
    "g2_zero_start_720", "g2b_zero_start_iid_720", "g3_params_yml_1500", "g3_get_default_1500",
    "g3_mixed_zero_start_800", "g3_list_actions_200", "g4_discrete_yaw5", "g4_no_yaw", "g4_auto_jump",
    "g4_no_jump", "g4_hover", "g4_speed_reward", "g4_no_smooth", "g4_delay0", "g4_dataclass_defaults",
    "g4_short_episodes", "g4_fmove_small", "g4_weird_key_values", "s1_reference_test_scenario", "s2_constant_action_720",
]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def config_kwargs(fx):
    kw = json.loads(str(fx["config_json"]))
    kw["initial_yaw_range"] = tuple(kw["initial_yaw_range"])
    if abs(kw["action_range"] - 10.079999923706055) < 1e-12:
        kw["action_range"] = np.float32(720) * np.float32(0.014)     # the dataclass default, as float32
    return kw


def rllib_rows(a):
    """ndarray row block -> RLlib list-of-tuples format (scalars + (1,) float32 arrays)."""
    return [tuple([int(x) for x in row[:-1]] + [np.array([row[-1]], dtype=np.float32)]) for row in a]


def replay(make_env, fx, name, getters):
    """make_env(kwargs) -> env with vector_reset / vector_step / reset_at.
    getters: dict field -> callable(env) returning the array for that field after a tick."""
    kw = config_kwargs(fx)
    np.random.seed(int(fx["seed"]))
    env = make_env(kw)
    obs0 = env.vector_reset() if bool(fx["second_reset"]) else None
    out = {k: [] for k in ("obs", "reward", "done", "zero_start")}
    out.update({k: [] for k in getters})
    reset_obs = []
    actions = fx["actions"]
    reset_on_done = name.startswith(("g3_", "g4_"))
    for t in range(actions.shape[0]):
        a = actions[t]
        if name == "g3_list_actions_200":
            a = rllib_rows(a)
        obs, rew, done, infos = env.vector_step(a)
        out["obs"].append(np.asarray(obs))
        out["reward"].append(np.asarray(rew))
        out["done"].append(np.asarray(done))
        out["zero_start"].append(np.asarray(infos if isinstance(infos, np.ndarray)
                                            else [i["zero_start"] for i in infos]))
        for k, g in getters.items():
            out[k].append(np.array(g(env)))
        if reset_on_done:
            for i in np.nonzero(np.asarray(done))[0]:
                reset_obs.append(np.asarray(env.reset_at(int(i)), dtype=np.float64))
    res = {k: np.stack(v) for k, v in out.items()}
    res["obs0"] = obs0
    res["reset_obs"] = np.stack(reset_obs) if reset_obs else np.zeros((0, 6))
    return res, env
