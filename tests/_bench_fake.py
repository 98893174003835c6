"""Oracle-backed stand-in for q1physrl_amd.device.DeviceEnv, used ONLY by tests/test_bench_launcher.py (through bench.py's
Q1_BENCH_ENV_FACTORY hook) so that bench.py's launcher, rendezvous, timed region, rank reduction and JSON line can run with
world size 2 over gloo in a container without a GPU.  It implements exactly the handle methods bench.py's main path calls;
"device pointers" are host addresses of CPU torch tensors."""
import ctypes
import time

import numpy as np

from oracle import np_oracle as O


def _view(ptr, count, dtype):
    buf = (ctypes.c_char * (count * np.dtype(dtype).itemsize)).from_address(int(ptr))
    return np.frombuffer(buf, dtype=dtype, count=count)


class FakeDeviceEnv:
    def __init__(self, config, num_envs=None, device=0, stream=None, env_index_base=0):
        kw = dict(config.__dict__)
        if num_envs is not None:
            kw["num_envs"] = num_envs
        np.random.seed(1000 + int(env_index_base))
        self._env = O.OracleVectorEnv(O.OracleConfig(**kw))
        self.n = kw["num_envs"]
        self.env_index_base = int(env_index_base)
        self.calls = []
        self._t0 = self._t1 = 0.0

    def _tick(self, keys, mouse, obs, reward, done):
        a = np.concatenate([((keys[:, None] >> np.arange(4)[None, :]) & 1).astype(np.float64),
                            mouse.astype(np.float64)[:, None]], axis=1)
        o, r, d, _ = self._env.vector_step(a)
        if obs is not None:
            obs[:] = o.astype(np.float32).reshape(-1)
        if reward is not None:
            reward[:] = r
        if done is not None:
            done[:] = d.astype(np.uint8)

    def step_many_dev(self, ticks, action_format, act_a, act_b=0, obs_format=1, obs=0, reward=0, done=0, out_stride_ticks=0,
                      use_graph=True):
        self.calls.append(("step_many", ticks, int(use_graph)))
        flags, use_graph = int(use_graph) & 124, int(use_graph) & 3
        if use_graph == 2:                           # prepare only
            return
        if flags & 4:
            self.timer_start()
        if flags & 16:                               # Q1ENV_STAMP_START: the "device" start stamp
            self._s0 = time.perf_counter()
        n = self.n
        for t in range(ticks):
            ot = t if out_stride_ticks else 0
            self._tick(_view(act_a + t * n, n, np.uint8), _view(act_b + 4 * t * n, n, np.float32),
                       _view(obs + 24 * ot * n, 6 * n, np.float32) if obs else None,
                       _view(reward + 4 * ot * n, n, np.float32) if reward else None,
                       _view(done + ot * n, n, np.uint8) if done else None)
        if flags & 8:
            self.timer_mark()
        if flags & (32 | 64):                        # Q1ENV_SIGNAL / _WAIT: end stamp + sequence number (synchronous stand-in)
            self.signal_mark()

    def persistent_start(self, *a):
        raise RuntimeError("the oracle stand-in has no tick server")

    def snapshot_state(self):
        import copy
        self._snap = copy.deepcopy(self._env)

    def restore_state(self):
        import copy
        self._env = copy.deepcopy(self._snap)

    def rollout_dev(self, ticks, action_format, act_a=0, act_b=0, rng_seed=0, obs_format=1, obs=0, reward=0, done=0,
                    auto_reset=False, return_sum=0):
        self.calls.append(("rollout", ticks))
        self.step_many_dev(ticks, action_format, act_a, act_b, obs_format, obs, reward, done, out_stride_ticks=1, use_graph=int(auto_reset) & 124)
        self.calls.pop()

    def prepare_rollout(self, *a, **k):
        import functools
        return functools.partial(self.rollout_dev, *a, **k)

    def reset_philox_dev(self, seed, mask=0, done_only=False, obs_format=1, obs=0, counter_dev=0):
        self.calls.append(("reset", bool(done_only)))
        for i in np.flatnonzero(self._env.t_rem < 0) if done_only else range(self.n):
            self._env.reset_at(int(i))

    def sync(self):
        pass

    # completion signal of the real handle (include/q1env.h, ABI v4): the stand-in runs synchronously, so the signal has always arrived
    def signal_mark(self):
        self._s1 = time.perf_counter()
        self._signalled = True

    def signal_wait(self, timeout_s=30.0):
        assert getattr(self, "_signalled", False), "signal_wait without a signalled launch"

    def signal_elapsed(self):
        return self._s1 - self._s0

    def timer_start(self):
        self._t0 = time.perf_counter()

    def timer_mark(self):
        self._t1 = time.perf_counter()

    def timer_elapsed(self):
        return (self._t1 - self._t0) * 1e3

    def timer_stop(self):
        self.timer_mark()
        return self.timer_elapsed()

    def close(self):
        pass


class FakeNoRollout(FakeDeviceEnv):
    """A stand-in whose multi-tick entry point is missing: bench.py's default mode must then fall back - on EVERY rank, at the
    same point - to per-tick launches and say so in the JSON line."""
    def rollout_dev(self, *a, **k):
        raise RuntimeError("this stand-in has no fused rollout")
