"""NumPy ORACLE for the q1physrl env hot path.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / the timed CPU baseline.  The product (q1physrl_amd) never imports it and has
no CPU fallback: without the HIP library it raises.

What it is: an independent, from-scratch restatement (explicit SoA state, explicit dtypes, one
function per tick) of the arithmetic the reference performs on this path under NumPy 2.2.6:

    VectorPhysEnv.vector_step   reference q1physrl_env/q1physrl_env/env.py:482-510
    ActionDecoder.map           env.py:225-269          (_fix_actions env.py:221-223)
    phys.apply                  q1physrl_env/q1physrl_env/phys.py:184-197
      _air_move phys.py:93-109, _angle_vectors 56-66, _user_friction 83-90, _accelerate 69-80,
      _do_z_physics 112-132
    _get_obs / _round_vel / _round_origin / get_obs_scale   env.py:381-408, 294-296
    vector_reset / reset_at (RNG draw order)                env.py:428-480
    ActionDecoder.vector_reset / reset_at                   env.py:271-291

Parity pin: tests/test_oracle_golden.py checks this module BIT-FOR-BIT against tests/golden/*.npz,
which oracle/gen_golden.py produced by running the reference itself in the build container
(fixtures G1-G5, anchors S1/S2 of SURVEY.md section 8c).  The reference's own test-suite has no
runnable test of this path, so those generated vectors are the pin.

Numerics contract restated: yaw, time_remaining, last_key_press_time, z_pos and
every horizontal intermediate are float64; vel is float32 storage (RNE on store); friction speed /
control, the +270 jump add and the reward multiply are float32; no FMA contraction; true division.
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Sequence, Tuple

import numpy as np

F32 = np.float32
F64 = np.float64

# sv_user.c / sv_phys.c constants, as float32 (phys.py:47-53)
MAX_SPEED = F32(320)
ACCELERATE = F32(10)
FRICTION = F32(4)
STOP_SPEED = F32(100)
JUMP_SPEED = F32(270)
GRAVITY = F32(800)
FLOOR_HEIGHT = F32(24.03125)

INITIAL_Z = F32(32.843201)      # env.py:54
INITIAL_VEL_Z = F32(-12)        # env.py:55
INITIAL_YAW_ZERO = F32(90)      # env.py:58
MAX_YAW_SPEED = F32(720)        # env.py:91
DEFAULT_ACTION_RANGE = F32(720) * F32(0.014)   # env.py:139 (float32 10.08)

KEY_LEFT, KEY_RIGHT, KEY_FORWARD, KEY_JUMP = 0, 1, 2, 3


@dataclasses.dataclass(frozen=True)
class OracleConfig:
    """Field-for-field mirror of reference Config (env.py:94-148); same field defaults."""
    num_envs: Optional[int]
    zero_start_prob: float
    initial_yaw_range: Tuple[float, float]
    max_initial_speed: float
    time_delta: float = 0.014
    time_limit: float = 5
    allow_yaw: bool = True
    action_range: float = DEFAULT_ACTION_RANGE
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
    def get_default(cls, **over):
        """env.py:150-170"""
        kw = dict(num_envs=None, allow_jump=True, allow_yaw=True, auto_jump=False, discrete_yaw_steps=-1,
                  fmove_max=800, smove_max=1060, hover=False, initial_yaw_range=(0, 360), key_press_delay=0.3,
                  max_initial_speed=700, smooth_keys=True, speed_reward=False, time_delta=1. / 72,
                  time_limit=10., zero_start_prob=0.01)
        kw.update(over)
        return cls(**kw)

    @property
    def num_keys(self) -> int:
        return 4 if (self.allow_jump and not self.auto_jump) else 3      # env.py:206-207


def fix_actions(actions) -> np.ndarray:
    """env.py:221-223: list of tuples whose members are scalars or length>=1 arrays -> (N, A) float64."""
    if isinstance(actions, np.ndarray) and actions.ndim == 2:
        return actions.astype(F64, copy=False)
    return np.array([[np.ravel(c)[0] for c in row] for row in actions], dtype=F64)


def max_yaw_delta(cfg, legacy: bool = False):
    """env.py:230 `_MAX_YAW_SPEED * time_delta`.  Under NumPy 2 (NEP 50) float32(720) * python-float is a float32 product (what
    the G1-G5 fixtures pin); under the NumPy 1.18.2 the reference's requirements.txt pins it is a float64 product
    (legacy=True; pinned by tests/golden/g3_legacy_promotion_*.npz, generated with the reference's module constant patched to
    float64, which is what value-based promotion did)."""
    if legacy:
        return F64(720.0) * F64(cfg.time_delta)
    return MAX_YAW_SPEED * F32(cfg.time_delta)


# ------------------------------------------------------------------------------------------ decode
def decode(cfg, dec, actions: np.ndarray, z_vel: np.ndarray, t_rem: np.ndarray):
    """One ActionDecoder.map call (env.py:225-269).  `dec` is a dict with keys last_press (N,K) f64,
    last_keys (N,K) bool, yaw (N) f64 and is updated in place.  Returns (yaw, smove, fmove, jump)."""
    K = cfg.num_keys
    a = fix_actions(actions)
    pressed = (a[:, :K].astype(np.int64) & 1) != 0        # astype(int) then `& (0/1 array)`: bit 0 of the truncated value (env.py:228,243)
    if not cfg.allow_yaw:
        dyaw = np.zeros(a.shape[0], dtype=F64)
    elif cfg.discrete_yaw_steps == -1:
        dyaw = (a[:, K] * F64(max_yaw_delta(cfg, dec.get("legacy", False)))) / F64(cfg.action_range)
    else:
        s = cfg.discrete_yaw_steps
        dyaw = ((a[:, K] - s) * F64(max_yaw_delta(cfg, dec.get("legacy", False)))) / F64(s)

    now = (F64(cfg.time_limit) - t_rem.astype(F64))[:, None]                     # env.py:241,246
    may_press = now >= (dec["last_press"] + F64(cfg.key_press_delay))
    prev = dec["last_keys"]
    keys = pressed & (may_press | prev)
    rising = keys & ~prev
    dec["last_press"] = np.where(rising, now, dec["last_press"])
    if cfg.smooth_keys:
        level = (keys.astype(F64) + prev.astype(F64)) * 0.5
    else:
        level = keys.astype(F64)
    dec["last_keys"] = keys
    dec["yaw"] = dec["yaw"] + dyaw

    smove = (F64(F32(cfg.smove_max)) * (level[:, KEY_RIGHT] - level[:, KEY_LEFT])).astype(np.int64)
    fmove = (F64(F32(cfg.fmove_max)) * level[:, KEY_FORWARD]).astype(np.int64)
    if cfg.auto_jump:
        jump = z_vel <= 16
    elif cfg.allow_jump:
        jump = keys[:, KEY_JUMP].copy()
    else:
        jump = np.zeros(a.shape[0], dtype=bool)
    return dec["yaw"], smove, fmove, jump


# ------------------------------------------------------------------------------------------ physics
def basis_from_yaw(yaw: np.ndarray):
    """phys.py:56-66 with pitch = roll = 0 (what the env always passes): forward = (c, s), right = (s, -c)."""
    rad = (yaw.astype(F64) * np.pi) / 180.
    return np.cos(rad), np.sin(rad)


def friction(vx: np.ndarray, vy: np.ndarray, dt: F64):
    """phys.py:83-90 on float32 velocity components; returns float64 components."""
    assert vx.dtype == F32 and vy.dtype == F32
    speed = np.sqrt(vx * vx + vy * vy)                           # float32 throughout
    control = np.maximum(speed, STOP_SPEED)                     # float32
    drop = (dt * control.astype(F64)) * F64(FRICTION)
    new_speed = np.maximum(F64(0), speed.astype(F64) - drop)
    moving = speed > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        k = new_speed / speed.astype(F64)
    fx = np.where(moving, vx.astype(F64) * k, vx.astype(F64))
    fy = np.where(moving, vy.astype(F64) * k, vy.astype(F64))
    return fx, fy


def horizontal_move(yaw, fmove, smove, on_ground, dt: F64, vx: np.ndarray, vy: np.ndarray):
    """phys.py:93-109 + 69-80: wish direction, ground friction (previous-tick on_ground), accelerate."""
    c, s = basis_from_yaw(yaw)
    f = fmove.astype(F64)
    m = smove.astype(F64)
    # einsum accumulates from +0.0: (0 + a0*b0) + a1*b1 - matters only for the sign of a zero result
    wx = (F64(0) + c * f) + s * m
    wy = (F64(0) + s * f) + (-c) * m
    wlen = np.sqrt(wx * wx + wy * wy)
    has_wish = wlen > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        dx = np.where(has_wish, wx / wlen, wx)
        dy = np.where(has_wish, wy / wlen, wy)
    wish_speed = np.minimum(F64(MAX_SPEED), wlen)

    fx, fy = friction(vx, vy, dt)
    hx = np.where(on_ground, fx, vx.astype(F64))
    hy = np.where(on_ground, fy, vy.astype(F64))

    cur = (F64(0) + hx * dx) + hy * dy                          # einsum again (phys.py:71)
    capped = np.where((wish_speed > 30) & ~on_ground, F64(30), wish_speed)
    add = np.maximum(F64(0), capped - cur)
    acc = np.minimum((F64(ACCELERATE) * dt) * wish_speed, add)
    return hx + acc * dx, hy + acc * dy                            # float64; caller rounds to float32


def vertical_move(jump, dt: F64, z_pos, vz: np.ndarray, on_ground, jump_released):
    """phys.py:112-132."""
    assert vz.dtype == F32
    jump_released = jump_released | ~jump
    do_jump = on_ground & jump & jump_released
    vz = vz + np.where(do_jump, JUMP_SPEED, F32(0))                # float32 add
    vz = (vz.astype(F64) - F64(GRAVITY) * dt).astype(F32)           # float64 subtract, RNE to float32
    z = z_pos.astype(F64) + dt * vz.astype(F64)
    landed = z < F64(FLOOR_HEIGHT)
    z = np.where(landed, F64(FLOOR_HEIGHT), z)
    vz = np.where(landed, F32(0), vz)
    return z, vz, landed, jump_released


def phys_apply(yaw, fmove, smove, jump, dt: float, st: dict) -> dict:
    """phys.py:184-197 on an SoA state dict {z_pos f64, vel (N,3) f32, on_ground, jump_released}."""
    dt = F64(dt)
    vel = st["vel"]
    nx, ny = horizontal_move(yaw, fmove, smove, st["on_ground"], dt, vel[:, 0], vel[:, 1])
    z, vz, og, jr = vertical_move(jump, dt, st["z_pos"], vel[:, 2], st["on_ground"], st["jump_released"])
    new_vel = np.empty_like(vel)
    new_vel[:, 0] = nx.astype(F32)
    new_vel[:, 1] = ny.astype(F32)
    new_vel[:, 2] = vz
    return {"z_pos": z, "vel": new_vel, "on_ground": og, "jump_released": jr}


def phys_apply_general(yaw, pitch, roll, fmove, smove, jump, dt, z_pos, vel, on_ground, jump_released):
    """phys.apply (phys.py:184-197) for ANY velocity dtype and general pitch / roll (phys.py:56-66), per-element time_delta:
    float32 vel follows the env path above; float64 vel (PlayerState.from_df, phys.py:168-170) keeps every intermediate float64
    (np.linalg.norm, the +270 add and the store are all float64 then)."""
    vt = vel.dtype.type
    dt = np.asarray(dt, dtype=F64)
    k = np.pi
    ry, rp, rr = (np.asarray(yaw, F64) * k) / 180., (np.asarray(pitch, F64) * k) / 180., (np.asarray(roll, F64) * k) / 180.
    sy, cy, sp, cp, sr, cr = np.sin(ry), np.cos(ry), np.sin(rp), np.cos(rp), np.sin(rr), np.cos(rr)
    m00, m01 = cp * cy, ((-1 * sr) * sp) * cy + (-1 * cr) * (-sy)
    m10, m11 = cp * sy, ((-1 * sr) * sp) * sy + (-1 * cr) * cy
    f, m = np.asarray(fmove).astype(F64), np.asarray(smove).astype(F64)
    wx = (F64(0) + m00 * f) + m01 * m
    wy = (F64(0) + m10 * f) + m11 * m
    wlen = np.sqrt(wx * wx + wy * wy)
    has_wish = wlen > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        dx = np.where(has_wish, wx / wlen, wx)
        dy = np.where(has_wish, wy / wlen, wy)
        wish_speed = np.minimum(F64(MAX_SPEED), wlen)
        vx, vy = vel[:, 0], vel[:, 1]
        speed = np.sqrt(vx * vx + vy * vy)                          # in vel's dtype (np.linalg.norm)
        control = np.maximum(speed, vt(100))
        new_speed = np.maximum(F64(0), speed.astype(F64) - (dt * control.astype(F64)) * F64(FRICTION))
        ratio = new_speed / speed.astype(F64)
    moving = speed > 0
    fx = np.where(moving, vx.astype(F64) * ratio, vx.astype(F64))
    fy = np.where(moving, vy.astype(F64) * ratio, vy.astype(F64))
    hx = np.where(on_ground, fx, vx.astype(F64))
    hy = np.where(on_ground, fy, vy.astype(F64))
    cur = (F64(0) + hx * dx) + hy * dy
    capped = np.where((wish_speed > 30) & ~on_ground, F64(30), wish_speed)
    add = np.maximum(F64(0), capped - cur)
    acc = np.minimum((F64(ACCELERATE) * dt) * wish_speed, add)
    out = np.empty_like(vel)
    out[:, 0] = (hx + acc * dx).astype(vt)
    out[:, 1] = (hy + acc * dy).astype(vt)
    jump = np.asarray(jump, dtype=bool)
    jr = np.asarray(jump_released, dtype=bool) | ~jump
    do_jump = np.asarray(on_ground, dtype=bool) & jump & jr
    vz = vel[:, 2] + np.where(do_jump, vt(270), vt(0))
    vz = (vz.astype(F64) - F64(GRAVITY) * dt).astype(vt)
    z = np.asarray(z_pos, F64) + dt * vz.astype(F64)
    landed = z < F64(FLOOR_HEIGHT)
    out[:, 2] = np.where(landed, vt(0), vz)
    return np.where(landed, F64(FLOOR_HEIGHT), z), out, landed, jr


# ------------------------------------------------------------------------------------------ observation
def obs_scale(cfg):
    return np.array([cfg.time_limit, 90., 100, 200, 200, 200], dtype=F64)      # env.py:294-296


def observe(cfg, t_rem, yaw, z_pos, vel) -> np.ndarray:
    """env.py:381-400: z rounded to 1/8 (half-even), vel truncated toward zero to multiples of 16."""
    zq = np.round(z_pos.astype(F64) * 8) / 8
    vq = ((vel / F32(16)).astype(np.int64) * 16).astype(F64)
    raw = np.concatenate([t_rem.astype(F64)[:, None], yaw.astype(F64)[:, None], zq[:, None], vq], axis=1)
    return raw / obs_scale(cfg)


# ------------------------------------------------------------------------------------------ env
class OracleVectorEnv:
    """Same call protocol as reference VectorPhysEnv (env.py:369-513) on explicit SoA arrays.

    RNG: draws from the GLOBAL np.random stream in the reference's order (env.py:432-446, 461-471),
    including the one-argument uniform(x) == uniform(low=x, high=1.0) quirk.
    """

    def __init__(self, cfg, legacy_promotion: bool = False):
        if isinstance(cfg, dict):
            cfg = OracleConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items()})
        self.cfg = cfg
        self.legacy_promotion = bool(legacy_promotion)
        self.n = cfg.num_envs
        self.step_num = 0
        self.vector_reset()

    # -- reset paths
    def _fresh_state(self):
        n = self.n
        vel = np.zeros((n, 3), dtype=F32)
        vel[:, 2] = INITIAL_VEL_Z
        return {"z_pos": np.full(n, INITIAL_Z, dtype=F64), "vel": vel,
                "on_ground": np.zeros(n, dtype=bool), "jump_released": np.ones(n, dtype=bool)}

    def vector_reset(self):
        cfg, n = self.cfg, self.n
        self.st = self._fresh_state()
        self.zero_start = np.random.random(size=(n,)) < cfg.zero_start_prob
        yaw_draw = np.random.uniform(cfg.initial_yaw_range[0], cfg.initial_yaw_range[1], size=(n,))
        time_draw = np.random.uniform(cfg.time_limit, size=(n,))          # low=time_limit, high=1.0 (quirk)
        speed_draw = np.random.uniform(cfg.max_initial_speed, size=(n,))
        self.yaw = np.where(self.zero_start, F64(INITIAL_YAW_ZERO), yaw_draw)
        self.t_rem = np.where(self.zero_start, F64(cfg.time_limit), time_draw)
        speed = np.where(self.zero_start, F64(0), speed_draw)
        angle = np.random.uniform(2 * np.pi, size=(n,))
        if cfg.hover:
            speed[:] = 320
            angle[:] = np.pi / 2
        self.st["vel"][:, 0] = (speed * np.cos(angle)).astype(F32)
        self.st["vel"][:, 1] = (speed * np.sin(angle)).astype(F32)
        k = cfg.num_keys
        self.dec = {"last_press": np.full((n, k), -F64(cfg.key_press_delay)),
                    "last_keys": np.zeros((n, k), dtype=bool), "yaw": self.yaw.copy()}
        return self.observation()

    def reset_at(self, i: int):
        cfg = self.cfg
        self.st["z_pos"][i] = INITIAL_Z
        self.st["vel"][i] = (0, 0, INITIAL_VEL_Z)
        self.st["on_ground"][i] = False
        self.st["jump_released"][i] = True
        z0 = bool(np.random.random() < cfg.zero_start_prob)
        self.zero_start[i] = z0
        # zero starts consume NO yaw/time/speed draws (conditional expressions, env.py:462-467)
        self.yaw[i] = F64(INITIAL_YAW_ZERO) if z0 else np.random.uniform(cfg.initial_yaw_range[0], cfg.initial_yaw_range[1])
        self.t_rem[i] = cfg.time_limit if z0 else np.random.uniform(cfg.time_limit)
        speed = 0 if z0 else np.random.uniform(cfg.max_initial_speed)
        angle = np.random.uniform(2 * np.pi)                              # always drawn (env.py:471)
        if cfg.hover:
            speed, angle = 320, np.pi / 2
        self.st["vel"][i, 0] = F32(speed * np.cos(angle))
        self.st["vel"][i, 1] = F32(speed * np.sin(angle))
        self.dec["last_press"][i] = -F64(cfg.key_press_delay)
        self.dec["last_keys"][i] = False
        self.dec["yaw"][i] = self.yaw[i]
        return self.observation()[i]

    # -- the tick
    def observation(self):
        return observe(self.cfg, self.t_rem, self.yaw, self.st["z_pos"], self.st["vel"])

    def vector_step(self, actions):
        cfg = self.cfg
        if cfg.hover:                                                      # env.py:483-485
            self.st["vel"][:, 2] = 0
            self.st["z_pos"][:] = 100
        self.dec["yaw"] = self.yaw
        self.dec["legacy"] = getattr(self, "legacy_promotion", False)
        yaw, smove, fmove, jump = decode(cfg, self.dec, actions, self.st["vel"][:, 2], self.t_rem)
        self.yaw = yaw
        self.last_cmd = (smove, fmove, jump)
        self.st = phys_apply(yaw, fmove, smove, jump, cfg.time_delta, self.st)
        vel = self.st["vel"]
        if cfg.speed_reward:                                               # env.py:500-503 (float32)
            reward = F32(cfg.time_delta) * np.sqrt(vel[:, 0] * vel[:, 0] + vel[:, 1] * vel[:, 1])
        else:
            reward = F32(cfg.time_delta) * vel[:, 1]
        self.t_rem = self.t_rem - F64(cfg.time_delta)
        done = self.t_rem < 0
        self.step_num += 1
        return self.observation(), reward, done, self.zero_start.copy()
