"""ctypes wrapper of the plain-C oracle (oracle/q1_oracle.c).  TEST INFRASTRUCTURE - used by tests/ and by
bench.py's cpu_baseline leg only.  Same call protocol as OracleVectorEnv; resets are delegated to the NumPy
oracle's reset code (they are not on the timed path)."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import np_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libq1oracle.so")


class Params(C.Structure):
    _fields_ = [("num_keys", C.c_int32), ("yaw_mode", C.c_int32), ("jump_mode", C.c_int32), ("smooth_keys", C.c_int32),
                ("hover", C.c_int32), ("speed_reward", C.c_int32), ("dt", C.c_double), ("time_limit", C.c_double),
                ("key_press_delay", C.c_double), ("yaw_num", C.c_double), ("yaw_den", C.c_double), ("yaw_steps", C.c_double),
                ("fmove_max", C.c_double), ("smove_max", C.c_double)]


class State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("vel", "z_pos", "yaw", "t_rem", "last_press", "on_ground", "jump_released", "last_keys")]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "q1_oracle.c")):
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        _lib = C.CDLL(LIB)
        _lib.q1o_has_openmp.restype = C.c_int
    return _lib


class COracleVectorEnv(O.OracleVectorEnv):
    """OracleVectorEnv whose tick is computed by q1_oracle.c (state arrays shared in place)."""

    def __init__(self, cfg, threads=1):
        self.threads = threads
        self.lib = load()
        super().__init__(cfg)

    def _bind(self):
        cfg = self.cfg
        self.p = Params()
        self.lib.q1o_make_params(C.byref(self.p), int(cfg.allow_yaw), int(cfg.discrete_yaw_steps), int(cfg.auto_jump),
                                 int(cfg.allow_jump), int(cfg.smooth_keys), int(cfg.hover), int(cfg.speed_reward),
                                 C.c_double(cfg.time_delta), C.c_double(cfg.time_limit), C.c_double(cfg.key_press_delay),
                                 C.c_double(float(cfg.action_range)), C.c_double(cfg.fmove_max), C.c_double(cfg.smove_max))

    def _c_state(self):
        n, k = self.n, self.cfg.num_keys
        # C side keeps 4 key slots per env
        if getattr(self, "_lp4", None) is None or self._lp4.shape[0] != n:
            self._lp4 = np.full((n, 4), -np.float64(self.cfg.key_press_delay))
            self._lk4 = np.zeros((n, 4), dtype=np.uint8)
        self._lp4[:, :k] = self.dec["last_press"]
        self._lk4[:, :k] = self.dec["last_keys"]
        self._og = self.st["on_ground"].astype(np.uint8)
        self._jr = self.st["jump_released"].astype(np.uint8)
        s = State()
        for name, arr in (("vel", self.st["vel"]), ("z_pos", self.st["z_pos"]), ("yaw", self.yaw), ("t_rem", self.t_rem),
                          ("last_press", self._lp4), ("on_ground", self._og), ("jump_released", self._jr), ("last_keys", self._lk4)):
            assert arr.flags["C_CONTIGUOUS"]
            setattr(s, name, arr.ctypes.data)
        return s

    def vector_step(self, actions):
        if not hasattr(self, "p"):
            self._bind()
        a = np.ascontiguousarray(O.fix_actions(actions), dtype=np.float64)
        n, k = self.n, self.cfg.num_keys
        self.st["z_pos"] = np.ascontiguousarray(self.st["z_pos"], dtype=np.float64)
        self.yaw = np.ascontiguousarray(self.yaw, dtype=np.float64)
        self.t_rem = np.ascontiguousarray(self.t_rem, dtype=np.float64)
        s = self._c_state()
        obs = np.empty((n, 6), np.float64)
        rew = np.empty((n,), np.float32)
        done = np.empty((n,), np.uint8)
        self.lib.q1o_step(C.byref(self.p), C.byref(s), C.c_int64(n), a.ctypes.data_as(C.c_void_p), obs.ctypes.data_as(C.c_void_p),
                          rew.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p), C.c_int(self.threads))
        self.dec["last_press"] = self._lp4[:, :k].copy()
        self.dec["last_keys"] = self._lk4[:, :k].astype(bool)
        self.dec["yaw"] = self.yaw
        self.st["on_ground"] = self._og.astype(bool)
        self.st["jump_released"] = self._jr.astype(bool)
        self.step_num += 1
        return obs, rew, done.astype(bool), self.zero_start.copy()


def time_rollout(n, ticks, threads, action_range, seed=0, period=32):
    """Throughput leg for bench.py: `ticks` ticks of n zero-start envs inside C (no Python per tick); the action
    tensor holds `period` ticks and is cycled."""
    import time
    np.random.seed(seed)
    env = COracleVectorEnv(O.OracleConfig.get_default(num_envs=n, zero_start_prob=1.0), threads=threads)
    env._bind()
    rng = np.random.default_rng(seed)
    period = min(period, ticks)
    a = np.concatenate([(rng.random((period, n, 4)) < 0.5).astype(np.float64),
                        rng.uniform(-action_range, action_range, (period, n, 1)).astype(np.float32).astype(np.float64)], axis=2)
    a = np.ascontiguousarray(a)
    s = env._c_state()
    obs = np.empty((n, 6), np.float64)
    rew = np.empty((n,), np.float32)
    done = np.empty((n,), np.uint8)
    t0 = time.perf_counter()
    env.lib.q1o_rollout(C.byref(env.p), C.byref(s), C.c_int64(n), C.c_int(ticks), C.c_int(period), a.ctypes.data_as(C.c_void_p),
                        obs.ctypes.data_as(C.c_void_p), rew.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return time.perf_counter() - t0
