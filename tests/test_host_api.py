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


def test_action_rows_c_helper_equals_the_numpy_path(monkeypatch):
    """csrc/q1rows.c (built by build.py with gcc): RLlib's list of N tuples - Python ints / bools / floats, NumPy scalars, arrays of
    length >= 1 of several dtypes - in one C pass; identical to the NumPy formulations on everything it accepts, and it declines
    (None -> NumPy path, which raises the reference's errors) on rows of the wrong length, nested lists and empty arrays."""
    from q1physrl_amd import build, env as E
    build.build_rows_helper()
    helper = E._load_rows_helper()
    if helper is None:
        pytest.skip("no compiler / Python headers here: the NumPy path is the only one")
    rng = np.random.default_rng(3)
    n = 4000
    rows = []
    for i in range(n):
        kind = i % 5
        keys = [int(rng.integers(2)), bool(rng.integers(2)), np.int64(rng.integers(2)), np.uint8(rng.integers(2))]
        mouse = [np.array([rng.uniform(-10, 10)], dtype=np.float32), float(rng.uniform(-10, 10)), np.float64(rng.uniform(-10, 10)),
                 np.array([rng.uniform(-10, 10), 99.0]), np.array([[rng.integers(-5, 5)]], dtype=np.int32)][kind]
        rows.append(tuple(keys) + (mouse,))
    monkeypatch.setattr(E, "_ROWS_HELPER", None)
    want = E._action_rows(rows, 5)
    monkeypatch.setattr(E, "_ROWS_HELPER", helper)
    got = E._action_rows(rows, 5)
    assert got.dtype == np.float64 and got.flags.c_contiguous and np.array_equal(got, want)
    assert np.array_equal(E._action_rows([list(r) for r in rows], 5), want)           # lists of lists too
    out = np.empty((1, 3))
    assert helper.rows([(1, 2)], 3, out) is None and helper.rows([(1, 2, [3])], 3, out) is None
    assert helper.rows([(1, 2, np.array([], dtype=np.float32))], 3, out) is None and helper.rows("abc", 3, out) is None
    with pytest.raises(ValueError):
        E._checked_rows([(1, 0, 1)], 5, 1)                                              # the wrapper's shape check is unchanged


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


# ---- vectorised reset draws (the RLlib reset_at sweep without 5 scalar np.random calls per env) ------------------------------
class _DrawHost:
    """Just enough of a VectorPhysEnv to call _draw_reset_rows without a device."""
    _draw_reset_rows = E.VectorPhysEnv._draw_reset_rows

    def __init__(self, config):
        self._config = config


def _plain_draws(c, k):
    """The reference's reset_at draw order, one np.random call per draw (env.py:461-471)."""
    zs, yaw, tm, sp, an = np.zeros(k, bool), np.zeros(k), np.zeros(k), np.zeros(k), np.zeros(k)
    for j in range(k):
        z = zs[j] = np.random.random() < c.zero_start_prob
        yaw[j] = 0.0 if z else np.random.uniform(*c.initial_yaw_range)
        tm[j] = 0.0 if z else np.random.uniform(c.time_limit)
        sp[j] = 0.0 if z else np.random.uniform(c.max_initial_speed)
        an[j] = np.random.uniform(2 * np.pi)
    return zs, yaw, tm, sp, an


@pytest.mark.parametrize("p,k", [(0.01, 1), (0.01, 2), (0.01, 3), (0.01, 1035), (0.5, 700), (1.0, 50), (0.0, 64), (0.3, 4097)])
def test_vectorised_reset_draws_equal_the_scalar_protocol(p, k):
    c = dataclasses.replace(E.Config.get_default(), num_envs=8, zero_start_prob=p, initial_yaw_range=(-30, 400), time_limit=7.5)
    np.random.seed(1234 + k)
    want = _plain_draws(c, k)
    end_want = np.random.random(3)                       # where the global stream is afterwards
    np.random.seed(1234 + k)
    consumed = []
    got = _DrawHost(c)._draw_reset_rows(k, consumed)
    end_got = np.random.random(3)
    for a, b in zip(want, got):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    assert np.array_equal(end_want, end_got)
    # cumulative consumption: 2 doubles per zero start, 5 otherwise - what a partial rollback rewinds to
    assert consumed == np.cumsum(np.where(want[0], 2, 5)).tolist()
    # rewinding to a claimed prefix lands exactly where the plain protocol would be after that many resets
    j = k // 2
    if j:
        np.random.seed(1234 + k)
        state = np.random.get_state()
        _DrawHost(c)._draw_reset_rows(k)
        np.random.set_state(state)
        np.random.random(consumed[j - 1])
        mine = np.random.random(2)
        np.random.seed(1234 + k)
        _plain_draws(c, j)
        assert np.array_equal(mine, np.random.random(2))


def test_numpy_promotion_default_follows_the_running_numpy(monkeypatch):
    from q1physrl_amd import device
    monkeypatch.delenv("Q1PHYSRL_NUMPY_PROMOTION", raising=False)
    assert device.legacy_promotion_default() == (int(np.__version__.split(".")[0]) < 2)
    cfg = dataclasses.replace(E.Config.get_default(), num_envs=4)
    assert device.make_c_config(cfg).legacy_promotion == int(device.legacy_promotion_default())
    assert device.make_c_config(cfg, numpy_promotion="legacy").legacy_promotion == 1
    assert device.make_c_config(cfg, numpy_promotion="nep50").legacy_promotion == 0
    monkeypatch.setenv("Q1PHYSRL_NUMPY_PROMOTION", "legacy")
    assert device.make_c_config(cfg).legacy_promotion == 1
    with pytest.raises(ValueError):
        device.make_c_config(cfg, numpy_promotion="numpy1")


# ---- gym integration (the reference is `class PhysEnv(gym.Env)`, registered at import: env.py:299, 516-521) -------------------
_STUB_GYM = '''
import sys, types
gym = types.ModuleType("gym")
class Env:
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    @property
    def unwrapped(self):
        return self
    def seed(self, seed=None):
        return
    def close(self):
        pass
class Error(Exception):
    pass
gym.Env = Env
err = types.ModuleType("gym.error"); err.Error = Error
envs = types.ModuleType("gym.envs")
reg = types.ModuleType("gym.envs.registration")
reg.calls = []
def register(id, **kw):
    if any(c[0] == id for c in reg.calls):
        raise Error("Cannot re-register id: " + id)
    reg.calls.append((id, kw))
reg.register = register
gym.error, gym.envs, envs.registration = err, envs, reg
sys.modules.update({"gym": gym, "gym.error": err, "gym.envs": envs, "gym.envs.registration": reg})
'''


def test_physenv_is_a_gym_env_and_registers_when_gym_is_importable():
    """With a gym on the path PhysEnv must BE a gym.Env (gym 0.17's EnvSpec.make does `env.unwrapped.spec = spec`; RLlib and
    wrappers test isinstance) and importing the module must register Q1PhysEnv-v0 exactly as env.py:516-521 does; importing it
    twice (the drop-in namespace re-exports it) must not fail; any OTHER registration error is genuine breakage and propagates at
    import, unless Q1PHYSRL_LENIENT_GYM_REGISTER=1 asks for a RuntimeWarning instead."""
    import subprocess
    import sys
    code = _STUB_GYM + '''
import importlib, dataclasses
import q1physrl_amd.env as E
import gym
assert issubclass(E.PhysEnv, gym.Env)
(i, kw), = gym.envs.registration.calls
assert i == "Q1PhysEnv-v0" and kw["entry_point"] == "q1physrl_amd.env:PhysEnv" and kw["nondeterministic"] is False
assert kw["kwargs"] == {"config": E.Config.get_default()}
importlib.reload(E)                                   # "Cannot re-register id" is swallowed, nothing else
e = E.PhysEnv.__new__(E.PhysEnv)                      # no device here: what gym.make does after constructing the env
e.unwrapped.spec = "spec"
assert e.spec == "spec" and e.seed(3) is None
import q1physrl_env.env as D
assert D.PhysEnv is E.PhysEnv or issubclass(D.PhysEnv, gym.Env)
def boom(id, **kw):
    raise RuntimeError("registry is broken")
gym.envs.registration.register = boom
try:                                                  # any OTHER registration failure propagates
    importlib.reload(E)
    raise SystemExit("a broken registry must fail the import")
except RuntimeError as ex:
    assert "registry is broken" in str(ex)
import os, warnings
os.environ["Q1PHYSRL_LENIENT_GYM_REGISTER"] = "1"     # ... unless the lenient behaviour is asked for: reported, import works
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    importlib.reload(E)
assert any("registry is broken" in str(x.message) and issubclass(x.category, RuntimeWarning) for x in w), [str(x.message) for x in w]
from q1physrl_amd import registry
assert "Q1PhysEnv-v0" in registry._REGISTRY if hasattr(registry, "_REGISTRY") else True
print("OK")
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_physenv_without_gym_has_the_gym_env_attributes():
    e = E.PhysEnv.__new__(E.PhysEnv)
    assert e.unwrapped is e and e.spec is None and e.seed(1) is None and E.PhysEnv.metadata == {}
    e.unwrapped.spec = 5
    assert e.spec == 5


def test_pinned_pool_recycles_blocks_only_when_every_view_is_gone(monkeypatch):
    """_lib.PinnedPool is np.empty for page-locked memory.  With the allocator stubbed (no GPU here): size classes, a block returns to
    the pool when the LAST array viewing it is collected - not before - and recycled blocks are handed out again."""
    import ctypes
    import gc
    from q1physrl_amd import _lib
    libc = ctypes.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = ctypes.c_void_p, [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    log = {"alloc": [], "free": []}

    class FakeLib:
        @staticmethod
        def q1env_host_alloc(size):
            p = libc.malloc(size)
            log["alloc"].append((p, size))
            return p

        @staticmethod
        def q1env_host_free(p):
            log["free"].append(p.value)
            libc.free(p)
            return 0

        @staticmethod
        def q1env_last_error():
            return b""
    monkeypatch.setattr(_lib, "load", lambda: FakeLib)
    monkeypatch.setattr(_lib, "_lib", FakeLib)
    pool = _lib.PinnedPool(cache_bytes=1 << 20)
    a = pool.empty((1000, 6), np.float64)                # 48 000 B -> 64 KiB class
    assert a.shape == (1000, 6) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and log["alloc"][-1][1] == 65536
    a[:] = 7.0
    view = a[10:20, 2]
    ptr_a = log["alloc"][-1][0]
    del a
    gc.collect()
    assert pool._cached == 0 and not log["free"]          # the view keeps the block alive
    assert float(view.sum()) == 70.0
    del view
    gc.collect()
    assert pool._cached == 65536                          # ... now it is back in the pool
    b = pool.empty((8192,), np.float64)                   # same class: recycled, no new allocation
    assert len(log["alloc"]) == 1 and pool._cached == 0 and b.ctypes.data == ptr_a
    c = pool.empty((3,), np.uint8)                        # smallest class: 4 KiB
    assert log["alloc"][-1][1] == 4096
    big = pool.empty((1 << 21,), np.uint8)                # 2 MiB: above the cache cap of this pool -> freed, not cached, on release
    del big
    gc.collect()
    assert len(log["free"]) == 1 and pool._cached == 0
    del b, c
    gc.collect()
    assert pool._cached == 65536 + 4096
    pool.trim()
    assert pool._cached == 0 and len(log["free"]) == 3
