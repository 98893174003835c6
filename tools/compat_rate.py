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
