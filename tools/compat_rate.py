"""PCIe-inclusive rate of the NumPy-compatible API (host arrays in / out through q1env_step_host), for DESIGN.md section 7."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from q1physrl_amd import env as E
for n, style in ((100, "rllib"), (4096, "rllib"), (65536, "rllib"), (4096, "ndarray"), (65536, "ndarray"), (1048576, "ndarray")):
    cfg = dict(E.Config.get_default().__dict__, num_envs=n)
    e = E.VectorPhysEnv(cfg)
    rng = np.random.default_rng(0)
    a = np.concatenate([(rng.random((n, 4)) < 0.5).astype(np.float64), rng.uniform(-10, 10, (n, 1))], axis=1)
    if style == "rllib":
        a = [tuple([int(x) for x in r[:4]] + [np.array([r[4]], dtype=np.float32)]) for r in a]
    e.vector_step(a)
    k = 20 if n >= 65536 else 200
    t0 = time.perf_counter()
    for _ in range(k):
        obs, rew, done, infos = e.vector_step(a)
    dt = (time.perf_counter() - t0) / k
    t0 = time.perf_counter()
    for i in range(50):
        e.reset_at(i % n)
    dr = (time.perf_counter() - t0) / 50
    print(f"n={n:8d} {style:8s} vector_step {dt*1e3:9.3f} ms = {n/dt/1e6:9.3f} M env-steps/s   reset_at {dr*1e6:8.1f} us")
    e.close()

# BASELINE configs[0]: the gym single-env loop (PhysEnv.step / reset on done), 1000 steps
np.random.seed(0)
g = E.PhysEnv(E.Config.get_default())
g.reset()
rng = np.random.default_rng(1)
acts = [tuple(int(x) for x in rng.integers(0, 2, 4)) + (np.array([rng.uniform(-10, 10)], dtype=np.float32),) for _ in range(1000)]
g.step(acts[0])
t0 = time.perf_counter()
for a in acts:
    _, _, d, _ = g.step(a)
    if d:
        g.reset()
dt = (time.perf_counter() - t0) / len(acts)
print(f"PhysEnv (gym, 1 env): {dt*1e6:.1f} us per step incl. resets = {1/dt/1e3:.1f} k env-steps/s")
g.close()
# reset_many vs reset_at
e = E.VectorPhysEnv(dict(E.Config.get_default().__dict__, num_envs=65536))
idx = np.arange(0, 65536, 720)
t0 = time.perf_counter(); [e.reset_at(int(i)) for i in idx]; t1 = time.perf_counter(); e.reset_many(idx); t2 = time.perf_counter()
print(f"{len(idx)} resets at 65536 envs: reset_at loop {1e3*(t1-t0):.2f} ms, reset_many {1e3*(t2-t1):.2f} ms")
e.close()

# RLlib-style loop with per-env resets: plain protocol vs speculative_resets=True (ndarray actions, 65 536 envs, short episodes)
for spec in (False, True):
    np.random.seed(3)
    e = E.VectorPhysEnv(dict(E.Config.get_default().__dict__, num_envs=65536, time_limit=0.3, zero_start_prob=0.0), speculative_resets=spec)
    rng = np.random.default_rng(0)
    a = np.concatenate([(rng.random((65536, 4)) < 0.5).astype(np.float64), rng.uniform(-10, 10, (65536, 1))], axis=1)
    t0 = time.perf_counter(); resets = 0
    for _ in range(80):
        _, _, done, _ = e.vector_step(a)
        for i in np.flatnonzero(done):
            e.reset_at(int(i)); resets += 1
    dt = (time.perf_counter() - t0) / 80
    print(f"65536 envs, vector_step + reset_at of every finished env (speculative_resets={spec}): {dt*1e3:.2f} ms/tick = {65536/dt/1e6:.1f} M env-steps/s, "
          f"{resets/80:.0f} resets/tick")
    e.close()
