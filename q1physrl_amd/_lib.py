"""ctypes binding of libq1env.so (include/q1env.h).  This is the whole Python <-> HIP boundary.

There is deliberately NO fallback: if the library is missing or no MI355X is visible the calls raise.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# Q1ENV_LIB_PATH selects another build of the SAME library (tools/asan_check.sh: the AddressSanitizer build of the host side)
LIB_PATH = os.environ.get("Q1ENV_LIB_PATH") or os.path.join(_PKG, "libq1env.so")

ABI_VERSION = 6
ACT_F64_ROWS, ACT_F32_ROWS, ACT_PACKED, ACT_RANDOM = 0, 1, 2, 3
OBS_F64, OBS_F32 = 0, 1
TIMER_START, TIMER_STOP = 4, 8     # q1env_step_many use_graph flags: record the handle's start / stop timer event around the launches
STAMP_START, SIGNAL, SIGNAL_WAIT = 16, 32, 64   # q1env_rollout auto_reset flags: the kernel-written completion signal (include/q1env.h)
FLAG_ON_GROUND, FLAG_JUMP_RELEASED, FLAG_ZERO_START, FLAG_LAST_KEY0 = 1, 2, 4, 8


class Q1EnvError(RuntimeError):
    """A libq1env call returned a negative status (message from q1env_last_error)."""


class Q1Config(C.Structure):          # q1env_config
    _fields_ = [("num_envs", C.c_int32), ("allow_yaw", C.c_int32), ("discrete_yaw_steps", C.c_int32),
                ("speed_reward", C.c_int32), ("hover", C.c_int32), ("smooth_keys", C.c_int32),
                ("auto_jump", C.c_int32), ("allow_jump", C.c_int32),
                ("zero_start_prob", C.c_double), ("initial_yaw_lo", C.c_double), ("initial_yaw_hi", C.c_double),
                ("max_initial_speed", C.c_double), ("time_delta", C.c_double), ("time_limit", C.c_double),
                ("action_range", C.c_double), ("fmove_max", C.c_double), ("smove_max", C.c_double),
                ("key_press_delay", C.c_double), ("env_index_base", C.c_int64),
                ("legacy_promotion", C.c_int32), ("reserved0", C.c_int32)]


class Q1State(C.Structure):           # q1env_state
    _fields_ = [("vel_x", C.c_void_p), ("vel_y", C.c_void_p), ("vel_z", C.c_void_p),
                ("pos_x", C.c_void_p), ("pos_y", C.c_void_p), ("z_pos", C.c_void_p),
                ("yaw", C.c_void_p), ("time_remaining", C.c_void_p),
                ("last_key_press_time", C.c_void_p), ("flags", C.c_void_p)]


class Q1Mlp(C.Structure):             # q1env_mlp
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w23_image", C.c_void_p), ("b2", C.c_void_p), ("b3", C.c_void_p),
                ("out", C.c_void_p), ("out_dim", C.c_int)]


class Q1ResidentArgs(C.Structure):   # q1env_resident_args
    _fields_ = [("ticks", C.c_int), ("deterministic", C.c_int), ("pi", C.POINTER(Q1Mlp)), ("seed", C.c_uint64), ("counter_dev", C.c_void_p),
                ("counter_offset", C.c_uint64), ("keys_dev", C.c_void_p), ("mouse_dev", C.c_void_p),
                ("logp_dev", C.c_void_p), ("obs_dev", C.c_void_p), ("reward_dev", C.c_void_p), ("done_dev", C.c_void_p),
                ("zero_start_dev", C.c_void_p), ("ep_return_dev", C.c_void_p), ("partials_dev", C.c_void_p), ("status_dev", C.c_void_p),
                ("timeout_s", C.c_double)]


class Q1LearnerNet(C.Structure):     # q1env_learner_net
    _fields_ = [(k, C.c_void_p) for k in ("w1", "b1", "w2", "b2", "w3", "b3", "gw1", "gb1", "gw2", "gb2", "gw3", "gb3")] + [("out_dim", C.c_int)]


class Q1LearnerBatch(C.Structure):   # q1env_learner_batch
    _fields_ = [("minibatch", C.c_int64), ("idx_dev", C.c_void_p), ("idx_cursor_dev", C.c_void_p), ("obs_dev", C.c_void_p), ("old_logits_dev", C.c_void_p), ("old_stride", C.c_int),
                ("keys_dev", C.c_void_p), ("mouse_dev", C.c_void_p), ("logp_old_dev", C.c_void_p), ("adv_dev", C.c_void_p),
                ("value_old_dev", C.c_void_p), ("vtarg_dev", C.c_void_p),
                ("clip_param", C.c_float), ("vf_clip_param", C.c_float), ("vf_loss_coeff", C.c_float), ("entropy_coeff", C.c_float),
                ("kl_coeff_dev", C.c_void_p), ("stats_partials_dev", C.c_void_p), ("skip_reduce", C.c_int), ("saturation_dev", C.c_void_p)]


STATE_FIELDS = (("vel_x", np.float32, 1), ("vel_y", np.float32, 1), ("vel_z", np.float32, 1),
                ("pos_x", np.float64, 1), ("pos_y", np.float64, 1), ("z_pos", np.float64, 1),
                ("yaw", np.float64, 1), ("time_remaining", np.float64, 1),
                ("last_key_press_time", np.float64, 4), ("flags", np.uint8, 1))

_P = C.c_void_p
_SIGNATURES = {
    "q1env_abi_version": (C.c_int, []),
    "q1env_last_error": (C.c_char_p, []),
    "q1env_device_count": (C.c_int, []),
    "q1env_create": (C.c_int, [C.POINTER(Q1Config), C.c_int, _P, C.POINTER(_P)]),
    "q1env_destroy": (C.c_int, [_P]),
    "q1env_set_stream": (C.c_int, [_P, _P]),
    "q1env_sync": (C.c_int, [_P]),
    "q1env_num_keys": (C.c_int, [_P]),
    "q1env_action_width": (C.c_int, [_P]),
    "q1env_tick_count": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "q1env_debug_counters": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int]),
    "q1env_reset_draws_host": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "q1env_reset_philox": (C.c_int, [_P, C.c_uint64, _P, _P, C.c_int, C.c_int, _P]),
    "q1env_step": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P]),
    "q1env_step_autoreset": (C.c_int, [_P, C.c_int, _P, _P, C.c_uint64, _P, _P, _P, _P, _P]),
    "q1env_step_autoreset_many": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_uint64, _P, _P, _P, _P, _P, C.c_int, C.c_int]),
    "q1env_step_host": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P]),
    "q1env_step_many": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int]),
    "q1env_rollout": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_uint64, C.c_int, _P, _P, _P, C.c_int, _P]),
    "q1env_observe": (C.c_int, [_P, C.c_int, _P]),
    "q1env_observe_host": (C.c_int, [_P, C.c_int, _P]),
    "q1env_get_state_host": (C.c_int, [_P, C.POINTER(Q1State)]),
    "q1env_set_state_host": (C.c_int, [_P, C.POINTER(Q1State)]),
    "q1env_state_device_ptrs": (C.c_int, [_P, C.POINTER(Q1State)]),
    "q1env_decode_host": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "q1env_decoder_reset_host": (C.c_int, [_P, C.c_int64, _P, _P]),
    "q1phys_apply_host": (C.c_int, [C.c_int, C.c_int64] + [_P] * 15),
    "q1phys_apply_host_f64": (C.c_int, [C.c_int, C.c_int64] + [_P] * 15),
    "q1env_host_alloc": (C.c_void_p, [C.c_uint64]),
    "q1env_host_free": (C.c_int, [_P]),
    "q1env_policy_sample": (C.c_int, [_P, _P, C.c_int, C.c_uint64, C.c_uint64, _P, C.c_int, _P, _P, _P]),
    "q1env_gae": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_float, C.c_float, _P, _P]),
    "q1env_snapshot_state": (C.c_int, [_P]),
    "q1env_restore_state": (C.c_int, [_P]),
    "q1env_policy_forward": (C.c_int, [_P] * 7 + [C.c_int, _P]),
    "q1env_policy_value_forward": (C.c_int, [_P, _P, _P, _P]),
    "q1env_ppo_loss_grad": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int] + [_P] * 7 + [C.c_float] * 4 + [_P] * 4),
    "q1env_learner_workspace_bytes": (C.c_uint64, [C.c_int64, C.c_int, C.c_int]),
    "q1env_learner_images": (C.c_int, [_P, C.POINTER(Q1LearnerNet), C.POINTER(Q1LearnerNet), _P, C.c_int64, C.c_int]),
    "q1env_learner_forward": (C.c_int, [_P, C.POINTER(Q1LearnerNet), C.POINTER(Q1LearnerNet), _P, C.c_int64, C.c_int, _P, _P, _P, _P]),
    "q1env_learner_backward": (C.c_int, [_P, C.POINTER(Q1LearnerNet), C.POINTER(Q1LearnerNet), _P, C.c_int64, C.c_int, _P, _P, _P, _P, C.c_float]),
    "q1env_learner_step": (C.c_int, [_P, C.POINTER(Q1LearnerNet), C.POINTER(Q1LearnerNet), _P, C.c_int, C.POINTER(Q1LearnerBatch)]),
    "q1env_learner_adam_state_bytes": (C.c_uint64, [C.c_int]),
    "q1env_learner_adam": (C.c_int, [_P, C.POINTER(Q1LearnerNet), C.POINTER(Q1LearnerNet), _P, C.c_int64, C.c_int] + [C.c_float] * 5 + [_P, _P]),
    "q1env_learner_sgd_step": (C.c_int, [_P, C.POINTER(Q1LearnerNet), C.POINTER(Q1LearnerNet), _P, C.c_int, C.POINTER(Q1LearnerBatch)] + [C.c_float] * 4 + [_P]),
    "q1env_learner_set_loss_scale": (C.c_int, [_P, C.c_float, C.c_float]),
    "q1env_learner_persistent_bytes": (C.c_uint64, [C.c_int64]),
    "q1env_learner_sgd_epochs": (C.c_int, [_P, C.POINTER(Q1LearnerNet), C.POINTER(Q1LearnerNet), _P, C.POINTER(Q1LearnerBatch)] + [C.c_int64] * 5
                                 + [C.c_float] * 4 + [_P, C.c_double]),
    "q1env_learner_sgd_epochs_f32": (C.c_int, [_P, C.POINTER(Q1LearnerNet), C.POINTER(Q1LearnerNet), _P, C.POINTER(Q1LearnerBatch)] + [C.c_int64] * 5
                                     + [C.c_float] * 4 + [_P, C.c_double]),
    "q1env_learner_set_exchange_mode": (C.c_int, [_P, C.c_int]),
    "q1env_learner_set_step_mode": (C.c_int, [_P, C.c_int]),
    "q1env_learner_set_profiling": (C.c_int, [_P, C.c_int]),
    "q1env_learner_persistent_layout": (C.c_int, [C.c_int64, C.c_int, C.POINTER(C.c_uint64)]),
    "q1env_learner_debug_counters": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "q1env_learner_persistent_status": (C.c_int, [_P, _P, C.POINTER(C.c_uint32)]),
    "q1env_sample_step": (C.c_int, [_P, _P, C.c_int, C.c_uint64, _P, C.c_uint64, C.c_int] + [_P] * 9),
    "q1env_episode_stats": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "q1env_step_persistent_start": (C.c_int, [_P, C.c_int, C.c_uint32, _P, _P, _P, C.c_uint64, C.c_int, _P, C.c_double]),
    "q1env_step_persistent_drive": (C.c_int, [_P, _P, C.c_int, C.c_uint32, _P, _P, _P, _P, _P, _P, C.c_double]),
    "q1env_step_persistent_publish": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "q1env_step_persistent_collect": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P, C.c_double]),
    "q1env_step_persistent_pair": (C.c_int, [_P, C.c_int, C.c_uint32, _P, _P, _P, _P, _P, C.c_uint64, C.c_int, _P, _P, C.c_double]),
    "q1env_policy_forward_rows": (C.c_int, [_P, C.c_uint64, _P, _P]),
    "q1env_sample_resident": (C.c_int, [_P, _P]),
    "q1env_selftest_division": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.POINTER(C.c_uint64)]),
    "q1env_selftest_trig": (C.c_int, [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "q1env_calibrate_traffic": (C.c_int, [_P, C.c_int]),
    "q1env_diag_signal_reader": (C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_double, _P]),
    "q1env_signal_mark": (C.c_int, [_P]),
    "q1env_signal_wait": (C.c_int, [_P, C.c_double]),
    "q1env_signal_read": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "q1env_build_id": (C.c_char_p, []),
    "q1env_timer_start": (C.c_int, [_P]),
    "q1env_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "q1env_timer_mark": (C.c_int, [_P]),
    "q1env_timer_elapsed": (C.c_int, [_P, C.POINTER(C.c_float)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so and load it by path; libq1env.so finds
    ROCm's copy through its RUNPATH.  If this library comes first, a later `import torch` brings a SECOND runtime into the process and
    torch.cuda then reports "No HIP GPUs are available".  So when a torch with a bundled runtime is installed (looked up without
    importing it) and not loaded yet, that copy is loaded first: same SONAME, so libq1env binds to it and torch re-uses it."""
    if "torch" in sys.modules:
        return                                            # torch's runtime is in the process already; the SONAME resolves to it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                                          # fall back to the system runtime


def load():
    """dlopen libq1env.so (built by q1physrl_amd.build / __graft_entry__.build()) and type its symbols."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Q1EnvError(f"{LIB_PATH} is missing: build it with `python -m q1physrl_amd.build` "
                         "(hipcc, gfx950).  q1physrl_amd has no CPU fallback.")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = the .so does not match include/q1env.h
        fn.restype, fn.argtypes = res, args
    if lib.q1env_abi_version() != ABI_VERSION:
        raise Q1EnvError(f"libq1env ABI {lib.q1env_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def lib_sha16():
    """First 16 hex digits of the sha256 of the library file this process loads (bench.py prints it as `lib_sha16`)."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def build_id():
    """The source hash compiled into the loaded library (q1physrl_amd/build.py sources_sha16)."""
    return load().q1env_build_id().decode()


def check(rc):
    if rc < 0:
        raise Q1EnvError(f"libq1env error {rc}: {load().q1env_last_error().decode(errors='replace')}")
    return rc


def ptr(a):
    """Host address of a C-contiguous NumPy array as an int (None -> NULL); every `void*` parameter of the binding accepts it.
    (`a.ctypes.data_as(c_void_p)` costs ~2.7 us per array - more than half of a small-batch vector_step's host time in round 3.)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data


# ---- page-locked host arrays -------------------------------------------------------------------------------------------------
PACK_MAX_ENVS = 16384      # q1env_core.hip: batches up to this size are packed through the handle's own pinned staging


class _PinnedBlock:
    """Owner of one q1env_host_alloc block; NumPy arrays made from it keep it alive through `.base`, and when the last of them is
    garbage-collected the block returns to the pool (so callers still own what they were handed, as with np.empty)."""
    __slots__ = ("ptr", "nbytes", "pool", "__array_interface__")

    def __init__(self, pool, ptr, nbytes):
        self.ptr, self.nbytes, self.pool = ptr, nbytes, pool
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            self.pool._release(self.ptr, self.nbytes)
        except Exception:   # noqa: BLE001 - interpreter shutdown
            pass


class PinnedPool:
    """np.empty for page-locked memory: size classes (powers of two >= 4 KiB), freed blocks are cached (up to `cache_bytes`) because
    hipHostMalloc costs milliseconds for the 48 MB observation block of a million envs."""

    def __init__(self, cache_bytes=2 << 30):
        self._free, self._cached, self._cap = {}, 0, int(cache_bytes)

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        need = max(int(np.prod(shape)) * dtype.itemsize, 1)
        size = 4096
        while size < need:
            size *= 2
        stack = self._free.get(size)
        if stack:
            ptr = stack.pop()
            self._cached -= size
        else:
            ptr = load().q1env_host_alloc(size)
            if not ptr:
                raise Q1EnvError(f"q1env_host_alloc({size}) failed: {load().q1env_last_error().decode(errors='replace')}")
        raw = np.asarray(_PinnedBlock(self, ptr, size))
        return raw[:need].view(dtype).reshape(shape)

    def _release(self, ptr, size):
        if self._cached + size <= self._cap:
            self._free.setdefault(size, []).append(ptr)
            self._cached += size
        elif _lib is not None:
            _lib.q1env_host_free(C.c_void_p(ptr))

    def trim(self):
        for size, stack in self._free.items():
            while stack:
                load().q1env_host_free(C.c_void_p(stack.pop()))
        self._cached = 0


_pool = None


def pinned_pool():
    global _pool
    if _pool is None:
        _pool = PinnedPool()
    return _pool


def host_empty(shape, dtype, n_envs):
    """An uninitialised host array for a *_host entry point: page-locked for batches the library does not pack itself."""
    if n_envs > PACK_MAX_ENVS:
        return pinned_pool().empty(shape, dtype)
    return np.empty(shape, dtype=dtype)
