"""DeviceEnv: thin object wrapper over one q1env_t handle (one shard of envs on one MI355X).

Everything numerical happens in libq1env.so; this file only marshals NumPy arrays / raw device
pointers across the C ABI (include/q1env.h).
"""
import ctypes as C

import numpy as np

from . import _lib


def legacy_promotion_default():
    """NumPy < 2 multiplies np.float32(720) by a Python float in float64 (env.py:230); NumPy >= 2 (NEP 50) in float32.  The
    default follows the NumPy the caller runs - i.e. what the reference itself would compute in this interpreter - and
    Q1PHYSRL_NUMPY_PROMOTION=legacy|nep50 overrides it."""
    import os
    v = os.environ.get("Q1PHYSRL_NUMPY_PROMOTION", "").lower()
    if v in ("legacy", "nep50"):
        return v == "legacy"
    return int(np.__version__.split(".")[0]) < 2


def make_c_config(config, num_envs=None, env_index_base=0, numpy_promotion=None):
    """reference Config (env.py:94-148) -> q1env_config POD.  numpy_promotion: None (follow the running NumPy), "nep50", "legacy"."""
    c = _lib.Q1Config()
    c.num_envs = int(config.num_envs if num_envs is None else num_envs)
    c.allow_yaw = int(bool(config.allow_yaw))
    c.discrete_yaw_steps = int(config.discrete_yaw_steps)
    c.speed_reward = int(bool(config.speed_reward))
    c.hover = int(bool(config.hover))
    c.smooth_keys = int(bool(config.smooth_keys))
    c.auto_jump = int(bool(config.auto_jump))
    c.allow_jump = int(bool(config.allow_jump))
    c.zero_start_prob = float(config.zero_start_prob)
    c.initial_yaw_lo = float(config.initial_yaw_range[0])
    c.initial_yaw_hi = float(config.initial_yaw_range[1])
    c.max_initial_speed = float(config.max_initial_speed)
    c.time_delta = float(config.time_delta)
    c.time_limit = float(config.time_limit)
    c.action_range = float(config.action_range)
    c.fmove_max = float(config.fmove_max)
    c.smove_max = float(config.smove_max)
    c.key_press_delay = float(config.key_press_delay)
    c.env_index_base = int(env_index_base)
    if numpy_promotion not in (None, "nep50", "legacy"):
        raise ValueError(f"numpy_promotion must be None, 'nep50' or 'legacy', got {numpy_promotion!r}")
    c.legacy_promotion = int(legacy_promotion_default() if numpy_promotion is None else numpy_promotion == "legacy")
    c.reserved0 = 0
    return c


class DeviceEnv:
    def __init__(self, config, num_envs=None, device=0, stream=None, env_index_base=0, numpy_promotion=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        cc = make_c_config(config, num_envs, env_index_base, numpy_promotion)
        self.legacy_promotion = bool(cc.legacy_promotion)
        self._pin_in = None
        self.n = cc.num_envs
        self.device = device
        _lib.check(self._lib.q1env_create(C.byref(cc), int(device), C.c_void_p(stream or 0), C.byref(self._h)))
        self.num_keys = _lib.check(self._lib.q1env_num_keys(self._h))
        self.action_width = _lib.check(self._lib.q1env_action_width(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.q1env_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 - interpreter shutdown
            pass

    # ---- host (NumPy) paths -----------------------------------------------------------------
    def step_host(self, actions_f64, obs_format=_lib.OBS_F64, want_zero_start=True):
        n = self.n
        a = np.ascontiguousarray(actions_f64, dtype=np.float64)
        if a.shape != (n, self.action_width):
            raise ValueError(f"actions must have shape ({n}, {self.action_width}), got {a.shape}")
        if n > _lib.PACK_MAX_ENVS:
            # large batches: page-locked arrays, so every copy is one direct DMA instead of the runtime's staged pageable path
            # (the returned arrays are the caller's, as always: their block goes back to the pool when they are collected)
            if self._pin_in is None:
                self._pin_in = _lib.pinned_pool().empty(a.shape, np.float64)
            np.copyto(self._pin_in, a)
            a = self._pin_in
        if n <= _lib.PACK_MAX_ENVS:
            # small batches: the four results are views of ONE fresh block (one allocation, one address lookup; the caller owns it
            # through the views' .base) - at 100 envs the call itself is ~12 us, so every microsecond of wrapper counts
            odt = np.float64 if obs_format == _lib.OBS_F64 else np.float32
            no = n * 6 * np.dtype(odt).itemsize
            blk = np.empty(no + n * 4 + 2 * n, dtype=np.uint8)
            base = blk.ctypes.data
            obs = blk[:no].view(odt).reshape(n, 6)
            reward = blk[no:no + 4 * n].view(np.float32)
            done = blk[no + 4 * n:no + 5 * n]
            zs = blk[no + 5 * n:] if want_zero_start else None
            _lib.check(self._lib.q1env_step_host(self._h, _lib.ACT_F64_ROWS, a.ctypes.data, None, obs_format, base, base + no, base + no + 4 * n,
                                                 (base + no + 5 * n) if want_zero_start else None))
            return obs, reward, done.view(np.bool_), (zs.view(np.bool_) if zs is not None else None)
        obs = _lib.host_empty((n, 6), np.float64 if obs_format == _lib.OBS_F64 else np.float32, n)
        reward = _lib.host_empty((n,), np.float32, n)
        done = _lib.host_empty((n,), np.uint8, n)
        zs = _lib.host_empty((n,), np.uint8, n) if want_zero_start else None
        _lib.check(self._lib.q1env_step_host(self._h, _lib.ACT_F64_ROWS, _lib.ptr(a), None, obs_format,
                                             _lib.ptr(obs), _lib.ptr(reward), _lib.ptr(done), _lib.ptr(zs)))
        return obs, reward, done.view(np.bool_), (zs.view(np.bool_) if zs is not None else None)

    def reset_draws(self, zero_start, yaw, time_remaining, speed, angle, idx=None, obs_format=_lib.OBS_F64):
        zs = np.ascontiguousarray(zero_start, dtype=np.uint8)
        cnt = zs.shape[0]
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (yaw, time_remaining, speed, angle)]
        assert all(x.shape == (cnt,) for x in arrs)
        ix = None if idx is None else np.ascontiguousarray(idx, dtype=np.int32)
        obs = _lib.host_empty((cnt, 6), np.float64 if obs_format == _lib.OBS_F64 else np.float32, cnt)
        _lib.check(self._lib.q1env_reset_draws_host(self._h, cnt, _lib.ptr(ix), _lib.ptr(zs), *[_lib.ptr(x) for x in arrs],
                                                    obs_format, _lib.ptr(obs)))
        return obs

    def snapshot_state(self):
        _lib.check(self._lib.q1env_snapshot_state(self._h))

    def restore_state(self):
        _lib.check(self._lib.q1env_restore_state(self._h))

    def observe_host(self, obs_format=_lib.OBS_F64):
        obs = _lib.host_empty((self.n, 6), np.float64 if obs_format == _lib.OBS_F64 else np.float32, self.n)
        _lib.check(self._lib.q1env_observe_host(self._h, obs_format, _lib.ptr(obs)))
        return obs

    def get_state(self, fields=None):
        """dict of fresh host arrays; last_key_press_time comes back as (N, 4) like the reference's."""
        out, st = {}, _lib.Q1State()
        for name, dt, mult in _lib.STATE_FIELDS:
            if fields is not None and name not in fields:
                continue
            out[name] = _lib.host_empty((mult * self.n,), dt, self.n)
            setattr(st, name, out[name].ctypes.data)
        _lib.check(self._lib.q1env_get_state_host(self._h, C.byref(st)))
        if "last_key_press_time" in out:
            out["last_key_press_time"] = np.ascontiguousarray(out["last_key_press_time"].reshape(4, self.n).T)
        return out

    def set_state(self, **arrays):
        st, keep = _lib.Q1State(), []
        for name, dt, mult in _lib.STATE_FIELDS:
            if name not in arrays:
                continue
            a = np.asarray(arrays[name], dtype=dt)
            if name == "last_key_press_time":
                a = a.reshape(self.n, 4).T
            a = np.ascontiguousarray(a).reshape(-1)
            assert a.shape == (mult * self.n,), (name, a.shape)
            keep.append(a)
            setattr(st, name, a.ctypes.data)
        _lib.check(self._lib.q1env_set_state_host(self._h, C.byref(st)))

    def device_ptrs(self):
        st = _lib.Q1State()
        _lib.check(self._lib.q1env_state_device_ptrs(self._h, C.byref(st)))
        return {name: getattr(st, name) for name, _, _ in _lib.STATE_FIELDS}

    def decode_host(self, actions_f64, z_vel, time_remaining):
        n = self.n
        a = np.ascontiguousarray(actions_f64, dtype=np.float64)
        if a.shape != (n, self.action_width):
            raise ValueError(f"actions must have shape ({n}, {self.action_width}), got {a.shape}")
        zv = np.ascontiguousarray(np.broadcast_to(np.asarray(z_vel, dtype=np.float32), (n,)))
        tr = np.ascontiguousarray(np.broadcast_to(np.asarray(time_remaining, dtype=np.float64), (n,)))
        yaw = np.empty((n,), np.float64)
        smove = np.empty((n,), np.int64)
        fmove = np.empty((n,), np.int64)
        jump = np.empty((n,), np.uint8)
        _lib.check(self._lib.q1env_decode_host(self._h, _lib.ACT_F64_ROWS, _lib.ptr(a), None, _lib.ptr(zv), _lib.ptr(tr),
                                               _lib.ptr(yaw), _lib.ptr(smove), _lib.ptr(fmove), _lib.ptr(jump)))
        return yaw, smove, fmove, jump.view(np.bool_)

    def decoder_reset(self, yaw, idx=None):
        y = np.ascontiguousarray(np.atleast_1d(np.asarray(yaw, dtype=np.float64)))
        ix = None if idx is None else np.ascontiguousarray(np.atleast_1d(idx), dtype=np.int32)
        _lib.check(self._lib.q1env_decoder_reset_host(self._h, y.shape[0], _lib.ptr(ix), _lib.ptr(y)))

    # ---- device-pointer paths (zero copy; pointers are ints, e.g. tensor.data_ptr()) ----------
    def step_dev(self, action_format, act_a, act_b=0, obs_format=_lib.OBS_F32, obs=0, reward=0, done=0, zero_start=0):
        _lib.check(self._lib.q1env_step(self._h, action_format, act_a or None, act_b or None, obs_format,
                                        obs or None, reward or None, done or None, zero_start or None))

    def step_autoreset_dev(self, action_format, act_a, act_b, seed, obs=0, reward=0, done=0, zero_start=0, counter_dev=0):
        _lib.check(self._lib.q1env_step_autoreset(self._h, action_format, act_a or None, act_b or None, int(seed), counter_dev or None,
                                                  obs or None, reward or None, done or None, zero_start or None))

    def step_autoreset_many_dev(self, ticks, action_format, act_a, act_b, seed, counter_dev, obs=0, reward=0, done=0, zero_start=0,
                                out_stride_ticks=0, use_graph=True):
        """`ticks` auto-reset ticks (one launch each + one counter node) from a cached hipGraph (q1env_step_autoreset_many)."""
        _lib.check(self._lib.q1env_step_autoreset_many(self._h, int(ticks), action_format, act_a or None, act_b or None, int(seed), counter_dev,
                                                       obs or None, reward or None, done or None, zero_start or None, int(out_stride_ticks),
                                                       int(use_graph)))

    def step_many_dev(self, ticks, action_format, act_a, act_b=0, obs_format=_lib.OBS_F32, obs=0, reward=0, done=0,
                      out_stride_ticks=0, use_graph=True):
        _lib.check(self._lib.q1env_step_many(self._h, ticks, action_format, act_a or None, act_b or None, obs_format,
                                             obs or None, reward or None, done or None, int(out_stride_ticks), int(use_graph)))   # use_graph: 0 eager, 1 replay, 2 prepare only

    def rollout_dev(self, ticks, action_format, act_a=0, act_b=0, rng_seed=0, obs_format=_lib.OBS_F32, obs=0, reward=0,
                    done=0, auto_reset=False, return_sum=0):
        _lib.check(self._lib.q1env_rollout(self._h, ticks, action_format, act_a or None, act_b or None, rng_seed, obs_format,
                                           obs or None, reward or None, done or None, int(auto_reset),     # bit 0 + timer flags
                                           return_sum or None))

    def prepare_rollout(self, ticks, action_format, act_a=0, act_b=0, rng_seed=0, obs_format=_lib.OBS_F32, obs=0, reward=0, done=0,
                        auto_reset=False, return_sum=0):
        """rollout_dev with the arguments converted ONCE: returns a zero-argument callable that issues the same q1env_rollout call.
        For loops that launch the same rollout repeatedly (a benchmark's timed region, a sampler replaying fixed buffers): ctypes'
        per-call conversion of twelve Python values is ~1.5 us of a ~9 us enqueue."""
        fn = self._lib.q1env_rollout
        p = lambda v: C.c_void_p(v) if v else None                 # noqa: E731
        args = (self._h, C.c_int(int(ticks)), C.c_int(int(action_format)), p(act_a), p(act_b), C.c_uint64(int(rng_seed)), C.c_int(int(obs_format)),
                p(obs), p(reward), p(done), C.c_int(int(auto_reset)), p(return_sum))

        def call():
            r = fn(*args)
            if r:
                _lib.check(r)
        return call

    def reset_philox_dev(self, seed, mask=0, done_only=False, obs_format=_lib.OBS_F32, obs=0, counter_dev=0):
        _lib.check(self._lib.q1env_reset_philox(self._h, seed, counter_dev or None, mask or None, int(done_only), obs_format, obs or None))

    def policy_sample_dev(self, logits, row_stride, seed, counter, keys, mouse, logp=0, deterministic=False, counter_dev=0):
        _lib.check(self._lib.q1env_policy_sample(self._h, logits, int(row_stride), int(seed), int(counter), counter_dev or None,
                                                 int(deterministic), keys, mouse or None, logp or None))

    def gae_dev(self, ticks, reward, value, done, gamma, lam, adv, vtarg):
        _lib.check(self._lib.q1env_gae(self._h, int(ticks), reward, value, done, float(gamma), float(lam), adv, vtarg))

    def policy_forward_dev(self, obs, w1, b1, w23_image, b2, b3, out_dim, out):
        _lib.check(self._lib.q1env_policy_forward(self._h, obs, w1, b1, w23_image, b2, b3, int(out_dim), out))

    def ppo_loss_grad_dev(self, batch, logits, old_logits, row_stride, keys, mouse, logp_old, adv, value, value_old, vtarg, clip_param,
                          vf_clip_param, vf_loss_coeff, entropy_coeff, kl_coeff_dev, dlogits, dvalue, partials):
        _lib.check(self._lib.q1env_ppo_loss_grad(self._h, int(batch), logits, old_logits, int(row_stride), keys, mouse or None, logp_old,
                                                 adv, value, value_old, vtarg, float(clip_param), float(vf_clip_param),
                                                 float(vf_loss_coeff), float(entropy_coeff), kl_coeff_dev, dlogits, dvalue, partials))

    # ---- native learner step (q1env_learner_*): nets are _lib.Q1LearnerNet, ws a device pointer of learner_workspace_bytes() bytes
    def learner_workspace_bytes(self, minibatch, out_dim_pi, splits):
        n = int(self._lib.q1env_learner_workspace_bytes(int(minibatch), int(out_dim_pi), int(splits)))
        if n == 0:
            raise _lib.Q1EnvError("q1env_learner_workspace_bytes: bad shape")
        return n

    def learner_images_dev(self, pi, vf, ws, minibatch, splits):
        _lib.check(self._lib.q1env_learner_images(self._h, C.byref(pi), C.byref(vf), ws, int(minibatch), int(splits)))

    def learner_forward_dev(self, pi, vf, ws, minibatch, splits, obs, idx=0, logits_out=0, value_out=0):
        _lib.check(self._lib.q1env_learner_forward(self._h, C.byref(pi), C.byref(vf), ws, int(minibatch), int(splits), obs, idx or None,
                                                   logits_out or None, value_out or None))

    def learner_backward_dev(self, pi, vf, ws, minibatch, splits, obs, idx, dlogits, dvalue, grad_scale):
        _lib.check(self._lib.q1env_learner_backward(self._h, C.byref(pi), C.byref(vf), ws, int(minibatch), int(splits), obs, idx or None,
                                                    dlogits, dvalue, float(grad_scale)))

    def learner_adam_state_bytes(self, out_dim_pi):
        return int(self._lib.q1env_learner_adam_state_bytes(int(out_dim_pi)))

    def learner_adam_dev(self, pi, vf, ws, minibatch, splits, grad_scale, lr, beta1, beta2, eps, state, stats_partials=0):
        _lib.check(self._lib.q1env_learner_adam(self._h, C.byref(pi), C.byref(vf), ws, int(minibatch), int(splits), float(grad_scale), float(lr),
                                                float(beta1), float(beta2), float(eps), state, stats_partials or None))

    def learner_sgd_step_dev(self, pi, vf, ws, splits, batch, lr, beta1, beta2, eps, state):
        _lib.check(self._lib.q1env_learner_sgd_step(self._h, C.byref(pi), C.byref(vf), ws, int(splits), C.byref(batch), float(lr), float(beta1),
                                                    float(beta2), float(eps), state))

    def learner_set_loss_scale(self, pi_upscale=0.0, value_downscale=0.0):
        """float16 loss scales of the learner calls that follow (exact powers of two; 0 = default): include/q1env.h."""
        _lib.check(self._lib.q1env_learner_set_loss_scale(self._h, float(pi_upscale), float(value_downscale)))

    def learner_persistent_bytes(self, batch_rows):
        return int(self._lib.q1env_learner_persistent_bytes(int(batch_rows)))

    def learner_sgd_epochs_dev(self, pi, vf, pws, batch, batch_rows, idx_rows, steps, steps_per_epoch, epoch_stride, lr, beta1, beta2, eps, state, timeout_s=5.0,
                               f32=False):
        """`steps` SGD steps of 128-sample minibatches as ONE dispatch (include/q1env.h q1env_learner_sgd_epochs; f32: ..._f32, float32 arithmetic)."""
        fn = self._lib.q1env_learner_sgd_epochs_f32 if f32 else self._lib.q1env_learner_sgd_epochs
        _lib.check(fn(self._h, C.byref(pi), C.byref(vf), C.c_void_p(int(pws)), C.byref(batch), int(batch_rows), int(idx_rows), int(steps),
                                                      int(steps_per_epoch), int(epoch_stride), float(lr), float(beta1), float(beta2), float(eps),
                                                      C.c_void_p(int(state)), float(timeout_s)))

    EXCHANGE_MODES = {"auto": 0, "agent": 1, "census_fail": 2}

    def learner_set_exchange_mode(self, mode):
        """Exchange mode of the persistent learner on this handle: "auto" (L2-local when the XCD census agrees, else agent scope), "agent",
        "census_fail" (tests: automatic, with a census made to disagree).  include/q1env.h."""
        _lib.check(self._lib.q1env_learner_set_exchange_mode(self._h, self.EXCHANGE_MODES[mode] if isinstance(mode, str) else int(mode)))

    STEP_MODES = {"auto": 0, "four_launch": 1, "fused": 2, "fused_dw1": 3, "fused_dw1_q": 4, "fused_dw1_r4wgrad": 5}

    def learner_set_step_mode(self, mode):
        """Kernel sequence of q1env_learner_sgd_step on this handle: "auto" (fused forward + backward kernel from 2 048 samples on), "four_launch"
        (round 4's path), "fused" (bit-identical to it), "fused_dw1" (dZ1 replaced by its per-tile products).  include/q1env.h."""
        _lib.check(self._lib.q1env_learner_set_step_mode(self._h, self.STEP_MODES[mode] if isinstance(mode, str) else int(mode)))

    def learner_set_profiling(self, wave_of_group=-1):
        _lib.check(self._lib.q1env_learner_set_profiling(self._h, int(wave_of_group)))

    def learner_persistent_layout(self, batch_rows, net):
        """{name: byte offset inside the persistent workspace} of network `net`'s exchange buffers + "bytes" (the group's workspace size)."""
        out = (C.c_uint64 * 9)()
        _lib.check(self._lib.q1env_learner_persistent_layout(int(batch_rows), int(net), out))
        return dict(zip(("bar", "b3x", "h1x", "h1tx", "dz2x", "w2tx", "yp", "w2st", "bytes"), (int(x) for x in out)))

    def learner_debug_counters(self):
        """(built with -DQ1_CHECK, exchange accesses checked, row indices checked, barrier readings checked, assertions failed)."""
        out = (C.c_uint64 * 5)()
        _lib.check(self._lib.q1env_learner_debug_counters(self._h, out))
        return tuple(int(x) for x in out)

    def learner_persistent_status(self, pws):
        st = (C.c_uint32 * 4)()
        _lib.check(self._lib.q1env_learner_persistent_status(self._h, C.c_void_p(int(pws)), st))
        return [int(x) for x in st]

    def learner_step_dev(self, pi, vf, ws, splits, batch):
        _lib.check(self._lib.q1env_learner_step(self._h, C.byref(pi), C.byref(vf), ws, int(splits), C.byref(batch)))

    def sample_step_dev(self, logits, row_stride, seed, counter_dev, counter_offset, deterministic, keys, mouse, logp, obs, reward,
                        done, zero_start, ep_return, partials):
        """policy_sample + step_autoreset + episode_stats of one sampler tick in one launch (q1env_sample_step)."""
        _lib.check(self._lib.q1env_sample_step(self._h, logits, int(row_stride), int(seed) & (2 ** 64 - 1), counter_dev,
                                               int(counter_offset), int(bool(deterministic)), keys, mouse, logp, obs, reward, done,
                                               zero_start, ep_return, partials))

    def policy_forward_rows_dev(self, rows, obs, net: "_lib.Q1Mlp"):
        """One network over `rows` observation rows of any buffer (q1env_policy_forward_rows)."""
        _lib.check(self._lib.q1env_policy_forward_rows(self._h, int(rows), obs, C.byref(net)))

    def sample_resident_dev(self, ticks, pi: "_lib.Q1Mlp", seed, counter_dev, counter_offset, deterministic, keys, mouse, logp, obs,
                            reward, done, zero_start, ep_return, partials, status, timeout_s=2.0):
        """A whole sampling horizon as one dispatch (q1env_sample_resident); all arguments are device addresses."""
        a = _lib.Q1ResidentArgs(int(ticks), int(bool(deterministic)), C.pointer(pi), int(seed) & 0xFFFFFFFFFFFFFFFF, counter_dev or None,
                                int(counter_offset), keys, mouse or None, logp, obs, reward, done, zero_start or None,
                                ep_return, partials, status, float(timeout_s))
        _lib.check(self._lib.q1env_sample_resident(self._h, C.byref(a)))

    def policy_value_forward_dev(self, obs, pi: "_lib.Q1Mlp", vf: "_lib.Q1Mlp"):
        """Both networks of a sampler tick in one launch (q1env_policy_value_forward); pi / vf are _lib.Q1Mlp structs."""
        _lib.check(self._lib.q1env_policy_value_forward(self._h, obs, C.byref(pi), C.byref(vf)))

    def episode_stats_dev(self, reward, done, zero_start, ep_return, partials):
        _lib.check(self._lib.q1env_episode_stats(self._h, reward, done, zero_start, ep_return, partials))

    def persistent_start(self, ticks, tag0, mailbox, results, obs_final, seed, auto_reset, status, timeout_s=2.0):
        """Launch the resident tick server on the handle's stream (q1env_step_persistent_start); pointers are device ints."""
        _lib.check(self._lib.q1env_step_persistent_start(self._h, int(ticks), int(tag0) & 0xFFFFFFFF, mailbox, results, obs_final or None,
                                                         int(seed) & (2 ** 64 - 1), int(bool(auto_reset)), status, float(timeout_s)))

    def persistent_drive(self, producer_stream, ticks, tag0, keys, mouse, mailbox, results, checksum, status, timeout_s=2.0):
        """Launch the reference dependent producer on `producer_stream` (raw hipStream_t int, not the handle's stream)."""
        _lib.check(self._lib.q1env_step_persistent_drive(self._h, C.c_void_p(int(producer_stream)), int(ticks), int(tag0) & 0xFFFFFFFF,
                                                         keys, mouse, mailbox, results, checksum or None, status, float(timeout_s)))

    def persistent_publish(self, producer_stream, tag0, tick, keys, mouse, mailbox):
        _lib.check(self._lib.q1env_step_persistent_publish(self._h, C.c_void_p(int(producer_stream)), int(tag0) & 0xFFFFFFFF, int(tick), keys,
                                                           mouse or None, mailbox))

    def persistent_collect(self, producer_stream, tag0, tick, results, obs, reward, done, zero_start, status, timeout_s=2.0):
        _lib.check(self._lib.q1env_step_persistent_collect(self._h, C.c_void_p(int(producer_stream)), int(tag0) & 0xFFFFFFFF, int(tick), results,
                                                           obs, reward or None, done or None, zero_start or None, status, float(timeout_s)))

    def persistent_pair(self, ticks, tag0, keys, mouse, mailbox, results, obs_final, seed, auto_reset, checksum, status, timeout_s=2.0):
        """Server + reference driver as one dispatch on the handle's stream (q1env_step_persistent_pair)."""
        _lib.check(self._lib.q1env_step_persistent_pair(self._h, int(ticks), int(tag0) & 0xFFFFFFFF, keys, mouse, mailbox, results,
                                                        obs_final or None, int(seed) & (2 ** 64 - 1), int(auto_reset),   # bit 0 + timer flags
                                                        checksum or None, status, float(timeout_s)))

    def observe_dev(self, obs, obs_format=_lib.OBS_F32):
        _lib.check(self._lib.q1env_observe(self._h, obs_format, obs))

    def set_stream(self, stream):
        """Rebind to a raw hipStream_t (int); 0 / None = the default (null) stream."""
        _lib.check(self._lib.q1env_set_stream(self._h, C.c_void_p(stream or 0)))

    def sync(self):
        _lib.check(self._lib.q1env_sync(self._h))

    def debug_counters(self, clear=False):
        """(built_with_Q1_CHECK, granule-pair stores checked, mismatches) of the assertion build (q1env_debug_counters)."""
        out = (C.c_uint64 * 4)()
        _lib.check(self._lib.q1env_debug_counters(self._h, out, int(bool(clear))))
        return int(out[0]), int(out[1]), int(out[2])

    def tick_count(self):
        """Ticks stepped since create: the handle-side Philox counter (q1env_tick_count)."""
        out = C.c_uint64()
        _lib.check(self._lib.q1env_tick_count(self._h, C.byref(out)))
        return int(out.value)

    def calibrate_traffic(self, launches=10):
        _lib.check(self._lib.q1env_calibrate_traffic(self._h, int(launches)))

    def diag_signal_reader(self, reader_stream, out_dev, expect_dev, off_reward, off_done, ticks, result_dev, workgroups=256, timeout_s=2.0):
        """Enqueue the visibility reader (include/q1env.h q1env_diag_signal_reader) on reader_stream for the NEXT signalled launch."""
        _lib.check(self._lib.q1env_diag_signal_reader(self._h, C.c_void_p(int(reader_stream)), C.c_void_p(int(out_dev)), C.c_void_p(int(expect_dev)),
                                                      int(off_reward), int(off_done), int(ticks), int(workgroups), float(timeout_s),
                                                      C.c_void_p(int(result_dev))))

    # ---- completion signal (include/q1env.h, ABI v4): stamps + sequence number written by the kernels themselves
    def signal_mark(self):
        """End stamp + sequence number behind whatever the stream holds (for regions that do not end in a rollout launch)."""
        _lib.check(self._lib.q1env_signal_mark(self._h))

    def signal_wait(self, timeout_s=30.0):
        """Poll the host-coherent sequence word until the last requested signal has arrived."""
        _lib.check(self._lib.q1env_signal_wait(self._h, float(timeout_s)))

    def signal_elapsed(self):
        """Seconds between the start stamp and the end stamp of the last signalled region (the device's constant-rate wall clock)."""
        a, b, hz = C.c_uint64(), C.c_uint64(), C.c_double()
        _lib.check(self._lib.q1env_signal_read(self._h, C.byref(a), C.byref(b), C.byref(hz)))
        return (int(b.value) - int(a.value)) / hz.value

    def timer_start(self):
        _lib.check(self._lib.q1env_timer_start(self._h))

    def timer_mark(self):
        """Record the stop event without waiting for it (read it later with timer_elapsed)."""
        _lib.check(self._lib.q1env_timer_mark(self._h))

    def timer_elapsed(self):
        ms = C.c_float()
        _lib.check(self._lib.q1env_timer_elapsed(self._h, C.byref(ms)))
        return ms.value

    def timer_stop(self):
        ms = C.c_float()
        _lib.check(self._lib.q1env_timer_stop(self._h, C.byref(ms)))
        return ms.value
