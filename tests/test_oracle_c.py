"""CPU: the plain-C oracle (oracle/q1_oracle.c) against the reference-generated golden traces.
Integers/booleans bit-exact; floats within 1e-5 relative and >= 99.9 % bit-identical (glibc sin/cos vs NumPy's)."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from tests import _replay as R

GETTERS = {
    "vel": lambda e: e.st["vel"], "z_pos": lambda e: e.st["z_pos"], "on_ground": lambda e: e.st["on_ground"],
    "jump_released": lambda e: e.st["jump_released"], "yaw": lambda e: e.yaw, "time_remaining": lambda e: e.t_rem,
    "last_key_press_time": lambda e: e.dec["last_press"], "last_keys": lambda e: e.dec["last_keys"].astype(np.uint8),
}


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("name", R.TRACE_FIXTURES)
def test_c_oracle_trace(name, threads):
    fx = R.load(name)
    res, env = R.replay(lambda kw: CO.COracleVectorEnv(kw, threads=threads), fx, name, GETTERS)
    for k in ("done", "zero_start", "on_ground", "jump_released", "last_keys"):
        assert np.array_equal(np.asarray(res[k]).astype(np.int64), np.asarray(fx[k]).astype(np.int64)), (name, k)
    for k in ("obs", "reward", "vel", "z_pos", "yaw", "time_remaining", "last_key_press_time"):
        a, b = np.ascontiguousarray(res[k]), np.ascontiguousarray(fx[k])
        assert a.dtype == b.dtype
        rel = np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1.0)
        assert float(rel.max()) <= 1e-5, (name, k, float(rel.max()))
        u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
        assert float(np.mean(a.view(u) == b.view(u))) >= 0.999, (name, k)
