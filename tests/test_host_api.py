"""CPU: host-side mirror of the reference interface (no compute): Config, enums, spaces, action
normalisation (_fix_actions job), lazy infos, registry, drop-in namespace."""
import dataclasses

import numpy as np
import pytest

from q1physrl_amd import env as E, registry, spaces


def test_config_matches_reference_contract():
    d = E.Config.get_default()
    assert d.num_envs is None and d.time_delta == 1. / 72 and d.time_limit == 10. and d.smove_max == 1060
    assert d.smooth_keys and d.key_press_delay == 0.3 and d.zero_start_prob == 0.01 and d.initial_yaw_range == (0, 360)
    assert isinstance(d.action_range, np.float32) and float(d.action_range) == 10.079999923706055   # f32(720)*f32(0.014)
    assert d.conforms_to_rules() and not dataclasses.replace(d, hover=True).conforms_to_rules()
    assert not dataclasses.replace(d, time_delta=0.014).conforms_to_rules()
    c = E.Config(num_envs=3, zero_start_prob=1., initial_yaw_range=(90, 90), max_initial_speed=0.)
    assert (c.time_delta, c.time_limit, c.smove_max, c.smooth_keys) == (0.014, 5, 700., False)      # field defaults
    with pytest.raises(TypeError):
        E.Config(num_envs=1, zero_start_prob=0, initial_yaw_range=(0, 1), max_initial_speed=0, nope=1)
    with pytest.raises(dataclasses.FrozenInstanceError):
        d.hover = True
    assert [f.name for f in dataclasses.fields(E.Config)] == [
        "num_envs", "zero_start_prob", "initial_yaw_range", "max_initial_speed", "time_delta", "time_limit", "allow_yaw",
        "action_range", "discrete_yaw_steps", "speed_reward", "fmove_max", "smove_max", "hover", "key_press_delay",
        "smooth_keys", "auto_jump", "allow_jump"]


def test_enums_and_obs_scale():
    assert [k.name for k in E.Key] == ["STRAFE_LEFT", "STRAFE_RIGHT", "FORWARD", "JUMP"] and int(E.Key.JUMP) == 3
    assert [o.name for o in E.Obs] == ["TIME_LEFT", "YAW", "Z_POS", "X_VEL", "Y_VEL", "Z_VEL"]
    assert E.get_obs_scale(E.Config.get_default()) == [10., 90., 100, 200, 200, 200]
    assert E.INITIAL_YAW_ZERO == np.float32(90)


@pytest.mark.parametrize("over,nkeys,mouse", [({}, 4, "box"), ({"auto_jump": True}, 3, "box"), ({"allow_jump": False}, 3, "box"),
                                              ({"discrete_yaw_steps": 5}, 4, 11), ({"allow_yaw": False}, 4, None)])
def test_action_space(over, nkeys, mouse):
    cfg = dataclasses.replace(E.Config.get_default(), num_envs=2, **over)
    sp = E.ActionDecoder(cfg).action_space
    parts = list(sp.spaces)
    assert len(parts) == nkeys + (0 if mouse is None else 1)
    assert all(p.n == 2 for p in parts[:nkeys])
    if mouse == "box":
        assert parts[-1].shape == (1,) and parts[-1].dtype == np.float32
        assert float(parts[-1].high[0]) == float(np.float32(cfg.action_range))
    elif mouse is not None:
        assert parts[-1].n == mouse


def test_action_rows_all_formats():
    want = np.array([[0, 1, 0, 1, 0.5], [1, 0, 1, 0, -2.25]])
    rllib = [(0, 1, 0, 1, np.array([0.5], dtype=np.float32)), (1, 0, 1, 0, np.array([-2.25], dtype=np.float32))]
    nested = [[[0.], [1.], [0.], [1.], [0.5]], [[1.], [0.], [1.], [0.], [-2.25]]]          # tests/test_integration.py:47 style
    ragged = [(0, np.array([1, 9]), 0, 1, 0.5), (1, 0, np.array([1]), 0, np.array([-2.25, 7., 8.]))]
    for a in (want, want.astype(np.float32), want.tolist(), rllib, nested, ragged):
        got = E._action_rows(a, 5)
        assert got.dtype == np.float64 and got.flags["C_CONTIGUOUS"] and np.array_equal(got, want)


def test_lazy_infos_behaves_like_the_list_of_dicts():
    zs = np.array([True, False, True])
    infos = E._LazyInfos(zs)
    assert len(infos) == 3 and infos[0] == {"zero_start": True} and infos[-2] == {"zero_start": False}
    assert list(infos) == [{"zero_start": z} for z in zs] and infos == [{"zero_start": z} for z in zs]
    (a, b, c) = infos
    assert c["zero_start"] and infos[1:] == [{"zero_start": False}, {"zero_start": True}]


def test_registry_and_dropin_namespace():
    assert "Q1PhysEnv-v0" in registry.registered()
    entry, kwargs = registry.registered()["Q1PhysEnv-v0"]
    assert entry is E.PhysEnv and kwargs["config"] == E.Config.get_default()
    with pytest.raises(KeyError):
        registry.make("nope-v0")
    import q1physrl_env.env as RE
    import q1physrl_env.phys as RP
    assert RE.VectorPhysEnv is E.VectorPhysEnv and RE.Config is E.Config and RE.Key is E.Key
    assert {"apply", "Inputs", "PlayerState"} <= set(dir(RP))
    for name in ('ActionDecoder', 'Config', 'get_obs_scale', 'INITIAL_YAW_ZERO', 'Key', 'Obs', 'PhysEnv', 'VectorPhysEnv'):
        assert hasattr(RE, name)


def test_physenv_rejects_num_envs():
    with pytest.raises(AssertionError, match="num_envs must be None for PhysEnv"):
        E.PhysEnv(dataclasses.replace(E.Config.get_default(), num_envs=4))


def test_fallback_spaces():
    b = spaces.Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32)
    assert b.contains(b.sample()) if hasattr(b, "contains") else True
    t = spaces.Tuple([spaces.Discrete(2), b])
    s = t.sample()
    assert len(s) == 2


def test_player_state_helpers():
    from q1physrl_amd import phys as P
    ps = P.PlayerState(np.array([1., 2.]), np.arange(6, dtype=np.float32).reshape(2, 3), np.array([True, False]), np.array([True, True]))
    cat = P.PlayerState.concatenate([ps, ps])
    assert cat.vel.shape == (4, 3) and cat.z_pos.tolist() == [1., 2., 1., 2.]
    df = ps.to_df()
    back = P.PlayerState.from_df(df)
    assert np.array_equal(back.vel, ps.vel) and np.array_equal(back.on_ground, ps.on_ground)
