"""Host-side mirror of the reference env module (q1physrl_env/q1physrl_env/env.py) over libq1env.so.

Same names, argument meaning, return types and error behaviour as the reference:

    Config (+get_default, conforms_to_rules)        env.py:94-180
    Key / Obs                                        env.py:61-86
    ActionDecoder (action_space, map, vector_reset, reset_at)   env.py:183-291
    get_obs_scale, INITIAL_YAW_ZERO                  env.py:294-296, 58
    PhysEnv (gym.Env protocol: step / reset)         env.py:299-358
    VectorPhysEnv (RLlib VectorEnv protocol)         env.py:369-513
    registration of 'Q1PhysEnv-v0'                   env.py:516-521

but every tick is ONE fused HIP kernel launch on an MI355X (include/q1env.h) instead of a few dozen
NumPy temporaries.  The only arithmetic left on the host is what the reference contract pins to the
host: drawing reset randomness from the GLOBAL NumPy MT19937 stream in the reference's order
(env.py:432-446, 461-471), so that `np.random.seed(s)` reproduces the reference's episodes.

There is no CPU fallback.  Without libq1env.so / without a GPU, constructing an env raises.
"""
import dataclasses
import enum
from typing import Optional, Tuple, Union

import numpy as np

from . import _lib, phys, spaces
from .device import DeviceEnv

__all__ = (
    'ActionDecoder',
    'Config',
    'get_obs_scale',
    'INITIAL_YAW_ZERO',
    'Key',
    'Obs',
    'PhysEnv',
    'VectorPhysEnv',
)

INITIAL_YAW_ZERO = np.float32(90)                       # env.py:58

# only used for the default of Config.action_range (env.py:89-91, 139): float32(720) * float32(0.014)
_DEFAULT_TIME_DELTA = np.float32(0.014)
_MAX_YAW_SPEED = np.float32(2 * 360)


class Key(enum.IntEnum):
    """Action-vector index of each key (env.py:61-73).  The mouse dimension follows the keys."""
    STRAFE_LEFT = 0
    STRAFE_RIGHT = 1
    FORWARD = 2
    JUMP = 3            # absent from the action vector when auto_jump or not allow_jump


class Obs(enum.IntEnum):
    """Column order of an observation row (env.py:76-86)."""
    TIME_LEFT = 0
    YAW = 1
    Z_POS = 2
    X_VEL = 3
    Y_VEL = 4
    Z_VEL = 5


@dataclasses.dataclass(frozen=True)
class Config:
    """Configuration of a PhysEnv / VectorPhysEnv; field-for-field the reference's (env.py:94-148).

    num_envs must be None iff used with `PhysEnv`.  The field defaults are the reference's
    "backwards compatibility" defaults; `get_default()` returns what `gym.make` uses.
    Unknown keys raise TypeError (dataclass constructor), as in the reference.
    """
    num_envs: Optional[int]
    zero_start_prob: float
    initial_yaw_range: Tuple[float, float]
    max_initial_speed: float
    time_delta: float = 0.014
    time_limit: float = 5
    allow_yaw: bool = True
    action_range: float = _MAX_YAW_SPEED * _DEFAULT_TIME_DELTA
    discrete_yaw_steps: int = -1
    speed_reward: bool = False
    fmove_max: float = 800.
    smove_max: float = 700.
    hover: bool = False
    key_press_delay: float = 0.3
    smooth_keys: bool = False
    auto_jump: bool = False
    allow_jump: bool = True

    @classmethod
    def get_default(cls):
        """Defaults used when the env is made through `gym.make` (env.py:150-170)."""
        return cls(num_envs=None, zero_start_prob=0.01, initial_yaw_range=(0, 360), max_initial_speed=700,
                   time_delta=1. / 72, time_limit=10., allow_yaw=True, discrete_yaw_steps=-1, speed_reward=False,
                   fmove_max=800, smove_max=1060, hover=False, key_press_delay=0.3, smooth_keys=True,
                   auto_jump=False, allow_jump=True)

    def conforms_to_rules(self):
        """Would these settings be legal under speed-running rules (env.py:172-180)."""
        return self.time_delta == 1. / 72 and not self.hover


def _num_keys(config) -> int:
    return len(Key) if (config.allow_jump and not config.auto_jump) else len(Key) - 1     # env.py:206-207


def get_obs_scale(config):
    """Observations are divided by these before being returned (env.py:294-296)."""
    return [config.time_limit, 90., 100, 200, 200, 200]


def _load_rows_helper():
    """csrc/q1rows.c, built by build.py next to the package (optional: None -> the NumPy formulations below)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_q1rows.so")
    if not os.path.exists(path):
        return None
    try:
        spec = importlib.util.spec_from_file_location("q1physrl_amd._q1rows", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    except Exception:   # noqa: BLE001 - built for another interpreter / NumPy: not an error, the NumPy path is equivalent
        return None


_ROWS_HELPER = _load_rows_helper()


def _action_rows(actions, width):
    """The job of ActionDecoder._fix_actions (env.py:221-223) without its per-element Python loop.

    Accepts (a) an (N, A) array, (b) RLlib's list of N tuples whose members are scalars or arrays of
    length >= 1 (first element taken, like np.ravel(x)[0]).  Returns float64 (N, A), C-contiguous.
    """
    if isinstance(actions, np.ndarray) and actions.ndim == 2 and actions.dtype != object:
        return np.ascontiguousarray(actions, dtype=np.float64)
    if _ROWS_HELPER is not None and type(actions) in (list, tuple) and len(actions):
        out = np.empty((len(actions), width), dtype=np.float64)         # RLlib's list of N tuples: one C pass (csrc/q1rows.c)
        if _ROWS_HELPER.rows(actions, int(width), out):
            return out
    first = actions[0] if len(actions) else ()
    mixed = isinstance(first, (tuple, list)) and any(isinstance(c, np.ndarray) for c in first) and \
        not all(isinstance(c, np.ndarray) for c in first)  # RLlib's rows: Python ints + a (1,) array - np.asarray can only fail on them
    if not mixed:
        try:                                               # homogeneous nested sequence of scalars
            a = np.asarray(actions, dtype=np.float64)
            if a.ndim == 2:
                return np.ascontiguousarray(a)
            if a.ndim == 3:                                # every component a length-m array
                return np.ascontiguousarray(a[:, :, 0])
        except (ValueError, TypeError):
            pass
    try:                                                   # column-wise: A conversions instead of N*A
        cols = list(zip(*actions))
        out = np.empty((len(actions), len(cols)), dtype=np.float64)
        for j, col in enumerate(cols):
            c = np.asarray(col, dtype=np.float64)
            out[:, j] = c.reshape(c.shape[0], -1)[:, 0]
        return out
    except (ValueError, TypeError):                        # ragged inside a column: the reference's way
        return np.array([[np.ravel(x)[0] for x in a] for a in actions], dtype=np.float64)


def _checked_rows(actions, width, n):
    rows = _action_rows(actions, width)
    if rows.shape != (n, width):
        raise ValueError(f"expected {n} actions of {width} components, got an array of shape {rows.shape}")
    return rows


class ActionDecoder:
    """Stateful action -> move-command decoder (env.py:183-291), state and arithmetic on the GPU.

    Tasks (unchanged): scale the mouse action to a yaw delta, rate-limit key presses
    (`key_press_delay`), smooth key transitions (`smooth_keys`), auto-jump.  The decoder must be reset
    with the env it accompanies.  Usable stand-alone (mkdemo.py:47-55, analyse.py:199-207) or as the
    view `VectorPhysEnv._action_decoder` onto the env's own fused decoder state.
    """

    def __init__(self, config: Config, *, device: int = 0, numpy_promotion: Optional[str] = None,
                 _shared: Optional[DeviceEnv] = None):
        self._config = config
        self._num_keys = _num_keys(config)
        self._device = device
        self._numpy_promotion = numpy_promotion
        self._dev: Optional[DeviceEnv] = _shared

    @property
    def action_space(self):
        c = self._config
        if not c.allow_yaw:
            mouse = []
        elif c.discrete_yaw_steps == -1:
            mouse = [spaces.Box(low=-c.action_range, high=c.action_range, shape=(1,), dtype=np.float32)]
        else:
            mouse = [spaces.Discrete(2 * c.discrete_yaw_steps + 1)]
        return spaces.Tuple([*(spaces.Discrete(2) for _ in range(self._num_keys)), *mouse])

    def _fix_actions(self, actions):
        return _action_rows(actions, self._num_keys + (1 if self._config.allow_yaw else 0))

    def _require(self) -> DeviceEnv:
        if self._dev is None:
            raise AttributeError("ActionDecoder used before vector_reset()")      # reference: missing attribute
        return self._dev

    def map(self, actions, z_vel, time_remaining):
        """Action vector -> (yaw, smove, fmove, jump) (env.py:225-269): float64, int64, int64, bool arrays."""
        dev = self._require()
        return dev.decode_host(self._fix_actions(actions), z_vel, time_remaining)

    def vector_reset(self, yaw):
        """Reset every decoder element before new episodes (env.py:271-281)."""
        yaw = np.array(yaw, dtype=np.float64).reshape(-1)
        if self._dev is None:
            n = self._config.num_envs if self._config.num_envs is not None else yaw.shape[0]
            self._dev = DeviceEnv(self._config, num_envs=n, device=self._device, numpy_promotion=self._numpy_promotion)
        self._dev.decoder_reset(yaw)

    def reset_at(self, index, yaw):
        """Reset one decoder element (env.py:283-291)."""
        self._require().decoder_reset(np.float64(yaw), idx=int(index))

    # read-only views of the decoder state (reference attributes, env.py:200-202)
    @property
    def _last_key_press_time(self):
        return self._require().get_state(("last_key_press_time",))["last_key_press_time"][:, :self._num_keys]

    @property
    def _last_keys(self):
        f = self._require().get_state(("flags",))["flags"]
        return ((f[:, None] >> (3 + np.arange(self._num_keys))[None, :]) & 1).astype(np.int64)

    @property
    def _yaw(self):
        return self._require().get_state(("yaw",))["yaw"]


class _LazyInfos:
    """The per-env info dicts of vector_step (env.py:510) built on demand: the reference's list
    comprehension is a third of its vectorised tick at 64 k envs."""
    __slots__ = ("_zs",)

    def __init__(self, zero_start):
        self._zs = zero_start

    def __len__(self):
        return self._zs.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [{'zero_start': z} for z in self._zs[i]]
        return {'zero_start': self._zs[i]}

    def __iter__(self):
        return ({'zero_start': z} for z in self._zs)

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return repr(list(self))


try:                                              # the reference's base class (env.py:299: `class PhysEnv(gym.Env)`) when gym is installed
    import gym as _gym
    _GymEnv = _gym.Env
except ImportError:
    _gym = None

    class _GymEnv:
        """What gym.Env (0.17, the reference's pin) gives a subclass, for images without gym: `gym.make` does
        `env.unwrapped.spec = spec`, wrappers read `metadata` / `reward_range` / `spec`, RLlib calls `seed` / `close`."""
        metadata = {'render.modes': []}
        reward_range = (-float('inf'), float('inf'))
        spec = None
        action_space = None
        observation_space = None

        @property
        def unwrapped(self):
            return self

        def seed(self, seed=None):
            return

        def render(self, mode='human'):
            raise NotImplementedError

        def close(self):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *args):
            self.close()
            return False

        def __str__(self):
            return f'<{type(self).__name__} instance>' if self.spec is None else f'<{type(self).__name__}<{self.spec.id}>>'


class PhysEnv(_GymEnv):
    """Single-env gym facade (env.py:299-358): 4 discrete keys + one continuous mouse dimension in,
    6-d observation (time left, yaw, z, velocity) out, reward = distance travelled along +Y this frame.
    No auto-reset: call reset() after done.  A gym.Env subclass whenever gym is importable (as in the reference)."""
    metadata = {}

    def __init__(self, config: Union[Config, dict], **device_kwargs):
        if isinstance(config, dict):
            config = Config(**config)
        if config.num_envs is not None:
            assert config.num_envs is None, "num_envs must be None for PhysEnv"
        config = dataclasses.replace(config, num_envs=1)
        self._env = VectorPhysEnv(config, **device_kwargs)
        self.observation_space = self._env.observation_space
        self.action_space = self._env.action_space
        self.reward_range = self._env.reward_range

    def step(self, action):
        (obs,), (reward,), (done,), (info,) = self._env.vector_step([action])
        return obs, reward, done, info

    def reset(self):
        (obs,) = self._env.vector_reset()
        return obs

    def seed(self, seed=None):
        """The reference inherits gym.Env.seed (a no-op returning None): its randomness is the GLOBAL NumPy stream."""
        return

    def close(self):
        self._env.close()


try:                                              # RLlib's base class when ray is installed (env.py:363-366)
    from ray.rllib.env import VectorEnv            # pragma: no cover
except ImportError:
    VectorEnv = object


class VectorPhysEnv(VectorEnv):
    """Vectorised Quake-1 movement env (env.py:369-513) whose state lives in HBM as SoA arrays.

    vector_reset() -> obs (N,6) float64; reset_at(i) -> obs (6,); vector_step(actions) ->
    (obs (N,6) float64, reward (N,) float32, done (N,) bool, infos) with infos[i] == {'zero_start': bool}.
    Extra keyword arguments (not in the reference): device, stream (raw hipStream_t), env_index_base, speculative_resets.

    speculative_resets=True batches RLlib's per-env reset protocol: the first `reset_at(i)` after a `vector_step` resets EVERY env
    that finished on that tick (one device call, draws taken from the global NumPy stream in ascending index order - the order
    RLlib's sampler calls reset_at in) and the following `reset_at` calls are answered from that batch.  Whatever the caller
    does not claim, or claims out of order, is rolled back (device state snapshot + NumPy RNG state), so results and the RNG
    stream equal the plain protocol's - provided nothing else draws from np.random between the reset_at calls of one tick.
    """

    def __init__(self, config, *, device: int = 0, stream: Optional[int] = None, env_index_base: int = 0,
                 speculative_resets: bool = False, numpy_promotion: Optional[str] = None):
        if isinstance(config, dict):
            config = Config(**config)
        self._config = config
        self.num_envs = self._config.num_envs
        self.observation_space = spaces.Box(low=-np.inf, high=np.inf, shape=(6,), dtype=np.float32)
        self.reward_range = (-1000 * self._config.time_delta, 1000 * self._config.time_delta)
        self.metadata = {}
        self._obs_scale = get_obs_scale(self._config)
        # numpy_promotion: how env.py:230's np.float32(720) * time_delta is evaluated - None = like the NumPy running here
        # ("nep50" for NumPy >= 2, "legacy" float64 product for the NumPy 1.18.2 the reference pins); see include/q1env.h
        self._dev = DeviceEnv(self._config, device=device, stream=stream, env_index_base=env_index_base,
                              numpy_promotion=numpy_promotion)
        self._action_decoder = ActionDecoder(self._config, device=device, _shared=self._dev)
        self.action_space = self._action_decoder.action_space
        self._step_num = 0
        self._cache = {}
        self._speculative = bool(speculative_resets)
        self._pending_done = None            # indices that finished on the last vector_step (speculative mode only)
        self._spec = None                    # the speculative batch in flight
        self.speculation_stats = {"batches": 0, "claimed": 0, "rolled_back": 0}
        self.vector_reset()

    # ---- resets: randomness from the global NumPy stream in the reference's order ---------------
    def vector_reset(self):
        self._settle_speculation()
        c, n = self._config, self.num_envs
        zero_start = np.random.random(size=(n,)) < c.zero_start_prob                       # env.py:432
        yaw = np.random.uniform(*c.initial_yaw_range, size=(n,))                            # env.py:436
        time_remaining = np.random.uniform(c.time_limit, size=(n,))                         # env.py:439 (low=limit, high=1)
        speed = np.random.uniform(c.max_initial_speed, size=(n,))                           # env.py:442
        angle = np.random.uniform(2 * np.pi, size=(n,))                                     # env.py:446
        self._cache = {}
        return self._dev.reset_draws(zero_start, yaw, time_remaining, speed, angle)

    def _draw_reset_rows(self, k, consumed=None):
        """The draws of k consecutive `reset_at` calls from the global NumPy stream (env.py:461-471: a zero start consumes no
        yaw / time / speed draw, the angle is always drawn), and the stream left exactly where those k calls leave it.

        Every draw of the reference's reset_at is ONE double of the legacy MT19937 stream (random() is it; uniform(a, b) is
        a + (b - a) * it; the one-argument uniform(x) is uniform(low=x, high=1)).  So instead of up to 5 k scalar np.random calls:
        draw 5 k doubles as one block, follow the data-dependent consumption (2 doubles for a zero start, 5 otherwise) with an
        integer walk over the block, gather the fields with array arithmetic, then rewind the stream and advance it by the
        number of doubles really consumed.  `consumed`, if a list, receives the cumulative count after each reset."""
        c = self._config
        p, (lo, hi), tl, ms, tau = c.zero_start_prob, c.initial_yaw_range, c.time_limit, c.max_initial_speed, 2 * np.pi
        if k <= 2:                                       # one or two resets: the plain calls are cheaper than a state save
            zs, yaw, tm, sp, an = np.zeros(k, bool), np.zeros(k), np.zeros(k), np.zeros(k), np.zeros(k)
            rnd, uni, used = np.random.random, np.random.uniform, 0
            for j in range(k):
                z = zs[j] = rnd() < p
                if not z:
                    yaw[j] = uni(lo, hi)
                    tm[j] = uni(tl)
                    sp[j] = uni(ms)
                an[j] = uni(tau)
                used += 2 if z else 5
                if consumed is not None:
                    consumed.append(used)
            return zs, yaw, tm, sp, an
        state = np.random.get_state()
        block = np.random.random(5 * k)
        step = np.where(block < p, 2, 5)                 # doubles a reset consumes IF it starts at this position
        hop = step.tolist()
        pos, starts = 0, [0] * k
        for j in range(k):                               # the only sequential part: integer hops
            starts[j] = pos
            pos += hop[pos]
        at = np.asarray(starts, dtype=np.intp)
        zs = block[at] < p
        go = ~zs
        yaw, tm, sp = np.zeros(k), np.zeros(k), np.zeros(k)
        a3 = at[go]
        yaw[go] = float(lo) + (float(hi) - float(lo)) * block[a3 + 1]
        tm[go] = float(tl) + (1.0 - float(tl)) * block[a3 + 2]
        sp[go] = float(ms) + (1.0 - float(ms)) * block[a3 + 3]
        an = tau + (1.0 - tau) * block[at + step[at] - 1]
        np.random.set_state(state)
        np.random.random(pos)                            # the stream is now where k reset_at calls leave it
        if consumed is not None:
            consumed.extend((at + step[at]).tolist())
        return zs, yaw, tm, sp, an

    def _settle_speculation(self):
        """Undo the part of a speculative reset batch the caller did not claim: device state back to the snapshot, the claimed
        resets re-applied from the rows drawn for them, NumPy RNG back to where the claimed prefix left it - but only if the
        global stream is still where the speculative draws left it; if something else drew from np.random in between, rewinding
        would replay those foreign draws, so the stream is left alone (RuntimeWarning: it is then ahead of the plain protocol's
        by the unclaimed draws)."""
        sp, self._spec, self._pending_done = self._spec, None, None
        if sp is not None:
            self.speculation_stats["claimed"] += sp["claimed"]
        if sp is None or sp["claimed"] == len(sp["order"]):
            return
        k = sp["claimed"]
        self.speculation_stats["rolled_back"] += len(sp["order"]) - k
        self._dev.restore_state()
        if k:
            self._dev.reset_draws(*[r[:k] for r in sp["rows"]], idx=sp["order"][:k])
        now = np.random.get_state()
        after = sp["rng_after"]
        if now[0] == after[0] and now[2:] == after[2:] and np.array_equal(now[1], after[1]):
            np.random.set_state(sp["rng"])
            if k:
                np.random.random(sp["consumed"][k - 1])   # one double per draw: exactly the claimed prefix's consumption
        else:
            import warnings
            warnings.warn("VectorPhysEnv(speculative_resets=True): the global NumPy RNG was used by other code between the "
                          "reset_at calls of one tick; the unclaimed speculative draws are not rewound", RuntimeWarning,
                          stacklevel=3)
            self.speculation_stats["foreign_rng_use"] = self.speculation_stats.get("foreign_rng_use", 0) + 1
        self._cache = {}

    def reset_at(self, index):
        index = int(index)
        if index < 0:                                   # the reference indexes NumPy arrays, so negatives wrap
            index += self.num_envs
        sp = self._spec
        if sp is not None:
            if sp["claimed"] < len(sp["order"]) and int(sp["order"][sp["claimed"]]) == index:
                sp["claimed"] += 1
                return sp["obs"][sp["claimed"] - 1].copy()
            self._settle_speculation()                  # out-of-order / unexpected index: fall back to the plain protocol
        elif self._pending_done is not None and self._pending_done.size > 1 and int(self._pending_done[0]) == index:
            order, self._pending_done = self._pending_done, None
            rng = np.random.get_state()
            consumed = []
            rows = self._draw_reset_rows(order.size, consumed)
            self._dev.snapshot_state()
            obs = self._dev.reset_draws(*rows, idx=order)
            self._spec = {"order": order, "obs": obs, "claimed": 1, "rng": rng, "rng_after": np.random.get_state(),
                          "rows": rows, "consumed": consumed}
            self.speculation_stats["batches"] += 1
            self._cache = {}
            return obs[0].copy()
        self._pending_done = None
        rows = self._draw_reset_rows(1)
        self._cache = {}
        return self._dev.reset_draws(*rows, idx=[index])[0]

    def reset_many(self, indices):
        """Extension (not in the reference): `[reset_at(i) for i in indices]` as ONE device call - the same draws from the
        global NumPy stream in the same order, the same states and observations, returned as a (len(indices), 6) array.
        RLlib resets finished envs one `reset_at` (one launch + one synchronising copy, ~70 us) at a time; a caller that
        knows the done set up front (`np.flatnonzero(done)`) saves that per-env round trip."""
        c, n = self._config, self.num_envs
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        idx = np.where(idx < 0, idx + n, idx)
        if idx.size == 0:
            return np.empty((0, 6), dtype=np.float64)
        if np.unique(idx).size != idx.size:            # a repeated index: later resets overwrite earlier ones, keep it sequential
            return np.stack([self.reset_at(int(i)) for i in idx])
        self._settle_speculation()
        self._cache = {}
        return self._dev.reset_draws(*self._draw_reset_rows(idx.size), idx=idx)

    # ---- the tick ----------------------------------------------------------------------------
    def vector_step(self, actions):
        rows = _checked_rows(actions, self._dev.action_width, self.num_envs)
        self._settle_speculation()
        obs, reward, done, zero_start = self._dev.step_host(rows)
        self._step_num += 1
        self._cache = {}
        if self._speculative:
            self._pending_done = np.flatnonzero(done)
        return obs, reward, done, _LazyInfos(zero_start)

    def get_unwrapped(self):
        return []

    def _get_obs(self):
        """Current observation of every env, (N, 6) float64 (env.py:392-400), without stepping."""
        self._settle_speculation()
        return self._dev.observe_host()

    def _get_obs_at(self, index):
        """Current observation of one env, (6,) float64 (env.py:402-408)."""
        self._settle_speculation()
        return self._dev.observe_host()[index]

    def close(self):
        self._dev.close()

    # ---- state the reference exposes as attributes (read by analyse.py:199-218) -----------------
    def _state(self):
        self._settle_speculation()
        if "st" not in self._cache:
            self._cache["st"] = self._dev.get_state()
        return self._cache["st"]

    @property
    def player_state(self) -> phys.PlayerState:
        """Immutable snapshot (fresh host arrays) of the reference's PlayerState (phys.py:156-161)."""
        s = self._state()
        return phys.PlayerState(z_pos=s["z_pos"].copy(),
                                vel=np.stack([s["vel_x"], s["vel_y"], s["vel_z"]], axis=1),
                                on_ground=(s["flags"] & _lib.FLAG_ON_GROUND) != 0,
                                jump_released=(s["flags"] & _lib.FLAG_JUMP_RELEASED) != 0)

    @property
    def _yaw(self):
        return self._state()["yaw"].copy()

    @property
    def _time_remaining(self):
        return self._state()["time_remaining"].copy()

    @property
    def _zero_start(self):
        return (self._state()["flags"] & _lib.FLAG_ZERO_START) != 0

    @property
    def distance(self):
        """Extension: float64 integrals of dt*vel_x, dt*vel_y since the last reset, shape (N, 2)."""
        s = self._state()
        return np.stack([s["pos_x"], s["pos_y"]], axis=1)

    def get_state(self):
        """Checkpoint of the full env + decoder state (dict of host arrays); inverse: set_state()."""
        self._settle_speculation()
        return self._dev.get_state()

    def set_state(self, **arrays):
        self._settle_speculation()
        self._cache = {}
        self._dev.set_state(**arrays)


def _register():
    """env.py:516-521: importing the module registers 'Q1PhysEnv-v0' (when gym is installed) and always
    fills q1physrl_amd.registry so `q1physrl_amd.make('Q1PhysEnv-v0')` works without gym."""
    from . import registry
    registry.register('Q1PhysEnv-v0', PhysEnv, {'config': Config.get_default()})
    if _gym is None:
        return
    import gym.envs.registration as reg
    try:
        reg.register(id='Q1PhysEnv-v0', entry_point=f'{__name__}:PhysEnv', nondeterministic=False,
                     kwargs={'config': Config.get_default()})
    except Exception as ex:                          # noqa: BLE001
        # A duplicate id (gym.error.Error('Cannot re-register id: ...'), e.g. the reference package imported first) is expected
        # and silent - the only error that is swallowed.
        dup = False
        try:
            import gym.error as _gerr
            dup = isinstance(ex, _gerr.Error) and ('re-register' in str(ex) or 'already registered' in str(ex).lower())
        except Exception:                            # noqa: BLE001 - gym without gym.error: fall back to the message
            dup = 're-register' in str(ex) or 'already registered' in str(ex).lower()
        if dup:
            return
        # anything else is genuine breakage (wrong entry_point / kwargs, a gym whose register() changed): fail at import like any
        # other import-time error, unless the caller explicitly asks for the lenient behaviour
        import os
        if os.environ.get("Q1PHYSRL_LENIENT_GYM_REGISTER") != "1":
            raise
        import warnings
        warnings.warn(f"q1physrl_amd: gym registration of 'Q1PhysEnv-v0' failed ({ex!r}); "
                      "q1physrl_amd.make('Q1PhysEnv-v0') still works", RuntimeWarning, stacklevel=2)


_register()
