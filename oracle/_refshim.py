"""Harness-side shims that let the read-only reference at /root/reference be imported in THIS
container (no gym, NumPy >= 1.24).  Test infrastructure only: used by oracle/gen_golden.py to make
tests/golden/*.npz.  Nothing here travels to, or is needed on, the GPU box.

Shims:
  * a stub `gym` module (gym.Env, gym.spaces.{Box,Discrete,Tuple}, gym.envs.registration.register)
  * np.bool / np.int / np.float aliases (removed in NumPy 1.24; used at env.py:56-57,228,265,269)
  * sys.dont_write_bytecode so nothing is written under /root/reference
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference/q1physrl_env"


def _install_gym_stub():
    if "gym" in sys.modules:
        return
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")

    class Env:
        pass

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Discrete:
        def __init__(self, n):
            self.n = n

    class Tuple:
        def __init__(self, spaces_):
            self.spaces = list(spaces_)

    registry = {}

    def register(id, **kwargs):
        registry[id] = kwargs

    gym.Env = Env
    spaces.Box, spaces.Discrete, spaces.Tuple = Box, Discrete, Tuple
    registration.register = register
    registration.registry = registry
    envs.registration = registration
    gym.spaces, gym.envs = spaces, envs
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.envs": envs,
                        "gym.envs.registration": registration})


def import_reference():
    """Returns (env_module, phys_module) of the reference, imported in place."""
    sys.dont_write_bytecode = True
    for name in ("bool", "int", "float"):
        if not hasattr(np, name):
            setattr(np, name, {"bool": np.bool_, "int": int, "float": float}[name])
    _install_gym_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    ref_env = importlib.import_module("q1physrl_env.env")
    ref_phys = importlib.import_module("q1physrl_env.phys")
    assert ref_env.__file__.startswith("/root/reference/"), ref_env.__file__
    return ref_env, ref_phys
