"""Mirror of the reference physics module's public surface (q1physrl_env/q1physrl_env/phys.py):
`Inputs` (phys.py:135-153), `PlayerState` (156-181) and `apply` (184-197), with `apply` executed by the
stateless HIP kernel behind q1phys_apply_host (include/q1env.h).  No CPU fallback."""
import dataclasses

import numpy as np

from . import _lib

__all__ = ('apply', 'Inputs', 'PlayerState')


@dataclasses.dataclass
class Inputs:
    """Per-frame move command, vectorised over players; fields as sent over Quake's network layer."""
    yaw: np.ndarray
    pitch: np.ndarray
    roll: np.ndarray
    fmove: np.ndarray
    smove: np.ndarray
    button2: np.ndarray
    time_delta: np.ndarray

    @classmethod
    def from_df(cls, df):
        g = lambda c: df[c].to_numpy()   # noqa: E731
        return cls(g("yaw"), g("pitch"), g("roll"), g("fmove"), g("smove"), g("button2") > 0, g("host_frametime"))

    def to_df(self):
        import pandas as pd
        return pd.DataFrame({"yaw": self.yaw, "pitch": self.pitch, "roll": self.roll, "fmove": self.fmove,
                             "smove": self.smove, "button2": self.button2, "host_frametime": self.time_delta})


@dataclasses.dataclass
class PlayerState:
    z_pos: np.ndarray
    vel: np.ndarray             # (N, 3) float32 (env) or float64 (from_df)
    on_ground: np.ndarray
    jump_released: np.ndarray

    @classmethod
    def from_df(cls, df):
        g = lambda c: df[c].to_numpy()   # noqa: E731
        return cls(g("z"), np.stack([g("velx"), g("vely"), g("velz")], axis=1), g("onground") > 0, g("jumpreleased") > 0)

    def to_df(self):
        import pandas as pd
        return pd.DataFrame({"z": self.z_pos, "velx": self.vel[:, 0], "vely": self.vel[:, 1], "velz": self.vel[:, 2],
                             "onground": self.on_ground, "jumpreleased": self.jump_released})

    @classmethod
    def concatenate(cls, player_states):
        names = [f.name for f in dataclasses.fields(cls)]
        return cls(**{n: np.concatenate([getattr(ps, n) for ps in player_states]) for n in names})


def _f64(a, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (n,)))


def _u8(a, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(a).astype(bool), (n,))).view(np.uint8)


def apply(inputs: Inputs, player_state: PlayerState, *, device: int = 0) -> PlayerState:
    """One frame of Quake player physics for N players (phys.py:184-197): returns a NEW PlayerState.

    As in the reference, the arithmetic follows the dtype of `player_state.vel`: float32 (what the env stores: friction
    speed, the +270 jump add and the stored result are float32) or float64 (what PlayerState.from_df yields, phys.py:168-170,
    the demo-analysis path: nothing is rounded to float32) - the returned vel has the same dtype.  Other dtypes are treated as
    float64, like NumPy's promotion would.  pitch / roll of all zeros take the yaw-only basis the env uses.
    """
    lib = _lib.load()
    v_in = np.asarray(player_state.vel)
    vdt = np.float32 if v_in.dtype == np.float32 else np.float64
    vel = np.ascontiguousarray(v_in, dtype=vdt)
    n = vel.shape[0]
    assert vel.shape == (n, 3)
    pitch = np.asarray(inputs.pitch)
    roll = np.asarray(inputs.roll)
    pitch = _f64(pitch, n) if np.any(pitch != 0) else None
    roll = _f64(roll, n) if np.any(roll != 0) else None
    out_z = np.empty((n,), np.float64)
    out_vel = np.empty((n, 3), vdt)
    out_og = np.empty((n,), np.uint8)
    out_jr = np.empty((n,), np.uint8)
    args = [_f64(inputs.yaw, n), pitch, roll, _f64(inputs.fmove, n), _f64(inputs.smove, n), _u8(inputs.button2, n),
            _f64(inputs.time_delta, n), _f64(player_state.z_pos, n), vel, _u8(player_state.on_ground, n),
            _u8(player_state.jump_released, n), out_z, out_vel, out_og, out_jr]
    fn = lib.q1phys_apply_host if vdt == np.float32 else lib.q1phys_apply_host_f64
    _lib.check(fn(int(device), n, *[_lib.ptr(a) for a in args]))
    return PlayerState(out_z, out_vel, out_og.view(np.bool_), out_jr.view(np.bool_))
