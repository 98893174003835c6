"""q1physrl_amd - the q1physrl env hot path (VectorPhysEnv.vector_step -> ActionDecoder.map -> phys.apply)
as hand-written HIP kernels for MI355X (gfx950) behind the reference's own env API.

    from q1physrl_amd import env, phys          # same surface as q1physrl_env.env / q1physrl_env.phys
    e = env.VectorPhysEnv(dict(env_config))     # state lives on the GPU; every tick is one kernel

The native library (libq1env.so, C ABI in include/q1env.h) is loaded on first use; there is no CPU path.
"""
from . import _lib  # noqa: F401
from .registry import make  # noqa: F401

__version__ = "0.1.0"
