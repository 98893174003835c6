"""GPU-resident fast path: torch-ROCm tensors in and out of the env with zero copies.

torch is plumbing here (device memory + the current stream); all arithmetic is the same HIP kernels the
NumPy-compatible path uses (include/q1env.h: q1env_step / q1env_step_many / q1env_rollout / q1env_reset_philox).

    env = TensorVectorEnv(config, device=0, seed=1)
    obs = env.reset()                                   # (N, 6) float32 on the GPU
    obs, reward, done = env.step_tensor(actions)        # actions: (N, A) float32 policy output, or (keys u8, mouse f32)
    env.reset_done()                                    # masked device-RNG reset of finished episodes

Outputs are views of env-owned buffers, valid until the next call (SURVEY.md section 8b ownership rule).
"""
from typing import Optional, Tuple

import torch

from . import _lib
from .device import DeviceEnv


class _DevArray:
    """__cuda_array_interface__ shim so torch.as_tensor can alias a raw device pointer of the SoA state."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class TensorVectorEnv:
    def __init__(self, config, device: int = 0, env_index_base: int = 0, seed: int = 0):
        if isinstance(config, dict):
            from .env import Config
            config = Config(**config)
        self.config = config
        self.num_envs = int(config.num_envs)
        self.device = torch.device("cuda", device)
        self.seed = int(seed)
        with torch.cuda.device(self.device):
            self._dev = DeviceEnv(config, device=device, env_index_base=env_index_base)
            self._dev.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        n = self.num_envs
        self.obs = torch.empty((n, 6), dtype=torch.float32, device=self.device)
        self.reward = torch.empty((n,), dtype=torch.float32, device=self.device)
        self.done = torch.empty((n,), dtype=torch.uint8, device=self.device)
        self.zero_start = torch.empty((n,), dtype=torch.uint8, device=self.device)
        self.num_keys = self._dev.num_keys
        self.action_width = self._dev.action_width

    # ---- stream plumbing -------------------------------------------------------------------
    def use_current_stream(self):
        """Re-bind to torch's current stream (call after entering a `torch.cuda.stream(...)` context)."""
        self._dev.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- API ---------------------------------------------------------------------------------
    def reset(self, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        m = 0
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == self.num_envs
            m = mask.data_ptr()
        self._dev.reset_philox_dev(self.seed, mask=m, done_only=False, obs_format=_lib.OBS_F32, obs=self.obs.data_ptr())
        return self.obs

    def reset_done(self) -> torch.Tensor:
        """Reset exactly the envs whose episode has ended (time_remaining < 0, env.py:506); returns fresh obs of all envs."""
        self._dev.reset_philox_dev(self.seed, mask=0, done_only=True, obs_format=_lib.OBS_F32, obs=self.obs.data_ptr())
        return self.obs

    def _act_ptrs(self, actions, lead: Tuple[int, ...]):
        if isinstance(actions, (tuple, list)):
            keys, mouse = actions
            assert keys.dtype == torch.uint8 and keys.is_contiguous() and tuple(keys.shape) == lead
            if self.config.allow_yaw:
                assert mouse.dtype == torch.float32 and mouse.is_contiguous() and tuple(mouse.shape) == lead
            return _lib.ACT_PACKED, keys.data_ptr(), (mouse.data_ptr() if mouse is not None else 0)
        a = actions
        assert a.is_contiguous() and tuple(a.shape) == lead + (self.action_width,), (a.shape, lead, self.action_width)
        if a.dtype == torch.float32:
            return _lib.ACT_F32_ROWS, a.data_ptr(), 0
        if a.dtype == torch.float64:
            return _lib.ACT_F64_ROWS, a.data_ptr(), 0
        raise TypeError(f"actions must be float32/float64 rows or (uint8 keys, float32 mouse); got {a.dtype}")

    def step_tensor(self, actions) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        fmt, a, b = self._act_ptrs(actions, (self.num_envs,))
        self._dev.step_dev(fmt, a, b, _lib.OBS_F32, self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                           self.zero_start.data_ptr())
        return self.obs, self.reward, self.done

    def step_autoreset(self, actions) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """One tick with in-kernel reset of finished episodes (one launch instead of step_tensor + reset_done): reward /
        done / zero_start describe the finished step, obs rows of finished envs are the new episodes' first observations."""
        fmt, a, b = self._act_ptrs(actions, (self.num_envs,))
        self._dev.step_autoreset_dev(fmt, a, b, self.seed, self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                                     self.zero_start.data_ptr())
        return self.obs, self.reward, self.done

    def step_many(self, actions, ticks: int, outputs: bool = False, use_graph: bool = True, auto_reset: bool = False):
        """`ticks` single-tick launches over tick-major actions; outputs=True returns tick-major (T,N,..) tensors.
        auto_reset=True: every tick also resets, in-kernel, the envs it finished (q1env_step_autoreset_many; the Philox counter
        lives in self.reset_counter, a device int64 advanced by the graph itself)."""
        fmt, a, b = self._act_ptrs(actions, (ticks, self.num_envs))
        if auto_reset:
            # ONE logical Philox counter per handle: the device-resident count the replayable graph reads is re-synchronised from
            # the handle's tick count before every call (a 8-byte fill on the stream), so this path, reset(), step_autoreset() and
            # the serve_* / rollout paths - which read the handle's own count - never reuse a (seed, env, counter) triple, in any
            # interleaving.  (q1env_step_autoreset_many advances both by `ticks`.)  The GpuSampler keeps its OWN counter (sampler.tick)
            # for its trajectory's draws: do not step a sampler's env through other entry points between its horizons.
            if not hasattr(self, "reset_counter"):
                self.reset_counter = torch.zeros((1,), dtype=torch.int64, device=self.device)
            self.reset_counter.fill_(self._dev.tick_count())
            n = self.num_envs
            if outputs:
                obs = torch.empty((ticks, n, 6), dtype=torch.float32, device=self.device)
                rew = torch.empty((ticks, n), dtype=torch.float32, device=self.device)
                done = torch.empty((ticks, n), dtype=torch.uint8, device=self.device)
            else:
                obs, rew, done = self.obs, self.reward, self.done
            self._dev.step_autoreset_many_dev(ticks, fmt, a, b, self.seed, self.reset_counter.data_ptr(), obs.data_ptr(), rew.data_ptr(),
                                              done.data_ptr(), self.zero_start.data_ptr() if not outputs else 0, int(outputs), use_graph)
            return obs, rew, done
        if outputs:
            n = self.num_envs
            obs = torch.empty((ticks, n, 6), dtype=torch.float32, device=self.device)
            rew = torch.empty((ticks, n), dtype=torch.float32, device=self.device)
            done = torch.empty((ticks, n), dtype=torch.uint8, device=self.device)
            self._dev.step_many_dev(ticks, fmt, a, b, _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), 1, use_graph)
            return obs, rew, done
        self._dev.step_many_dev(ticks, fmt, a, b, _lib.OBS_F32, self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(), 0, use_graph)
        return self.obs, self.reward, self.done

    def rollout(self, ticks: int, actions=None, outputs: bool = True, auto_reset: bool = False, obs_dtype=torch.float32,
                return_sum: Optional[torch.Tensor] = None):
        """Fused multi-tick kernel.  actions=None -> on-device iid random actions (Q1ENV_ACT_RANDOM, seed = self.seed)."""
        n = self.num_envs
        if actions is None:
            fmt, a, b = _lib.ACT_RANDOM, 0, 0
        else:
            fmt, a, b = self._act_ptrs(actions, (ticks, n))
        obs = rew = done = None
        if outputs:
            obs = torch.empty((ticks, n, 6), dtype=obs_dtype, device=self.device)
            rew = torch.empty((ticks, n), dtype=torch.float32, device=self.device)
            done = torch.empty((ticks, n), dtype=torch.uint8, device=self.device)
        self._dev.rollout_dev(ticks, fmt, a, b, self.seed, _lib.OBS_F32 if obs_dtype == torch.float32 else _lib.OBS_F64,
                              obs.data_ptr() if outputs else 0, rew.data_ptr() if outputs else 0,
                              done.data_ptr() if outputs else 0, auto_reset,
                              return_sum.data_ptr() if return_sum is not None else 0)
        return obs, rew, done

    def serve_ticks(self, keys: torch.Tensor, mouse: torch.Tensor, auto_reset: bool = True, timeout_s: float = 2.0, sync: bool = True,
                    two_streams: bool = False):
        """Experimental: `ticks` ticks on the RESIDENT tick server (q1env_step_persistent_*: no kernel boundary per tick, state in
        registers) fed by the reference dependent producer on a side stream, which hands tick t+1's action (keys uint8 (T,N),
        mouse float32 (T,N)) over only after tick t's results arrived.  Bit-identical to T step_autoreset calls.  Returns a dict:
        obs (N,6) / reward / done / zero_start of the LAST tick (decoded from the result granules), checksum (float64 (2,N): sums
        of the rewards / first observation column of ticks 0..T-2 as the producer received them) and status (the five uint32 of
        include/q1env.h for this launch: all zero = success; status[1] / status[3] != 0 = a side timed out, status[2] / status[4] = ticks left unserved).
        two_streams=False: server and producer are blocks of ONE dispatch (co-resident by construction); True: the producer runs on a
        high-priority side stream (its own hardware-queue pool) - the arrangement an external producer has."""
        n, d = self.num_envs, self.device
        ticks = int(keys.shape[0])
        assert keys.dtype == torch.uint8 and keys.is_contiguous() and tuple(keys.shape) == (ticks, n)
        assert mouse.dtype == torch.float32 and mouse.is_contiguous() and tuple(mouse.shape) == (ticks, n)
        if not hasattr(self, "_srv"):
            self._srv = {"mailbox": torch.zeros((n,), dtype=torch.int64, device=d), "results": torch.zeros((4, n, 2), dtype=torch.int64, device=d),
                         "status": torch.zeros((5,), dtype=torch.int32, device=d), "checksum": torch.zeros((2, n), dtype=torch.float64, device=d),
                         "stream": torch.cuda.Stream(device=d, priority=-1), "tag": 0}
        sv = self._srv
        sv["checksum"].zero_()
        sv["status"].zero_()
        cur = torch.cuda.current_stream(d)
        if two_streams:
            sv["stream"].wait_stream(cur)                   # the producer starts after the inputs exist
            self._dev.persistent_start(ticks, sv["tag"], sv["mailbox"].data_ptr(), sv["results"].data_ptr(), self.obs.data_ptr(), self.seed,
                                       auto_reset, sv["status"].data_ptr(), timeout_s)
            self._dev.persistent_drive(sv["stream"].cuda_stream, ticks, sv["tag"], keys.data_ptr(), mouse.data_ptr(), sv["mailbox"].data_ptr(),
                                       sv["results"].data_ptr(), sv["checksum"].data_ptr(), sv["status"].data_ptr(), timeout_s)
            cur.wait_stream(sv["stream"])
        else:
            self._dev.persistent_pair(ticks, sv["tag"], keys.data_ptr(), mouse.data_ptr(), sv["mailbox"].data_ptr(), sv["results"].data_ptr(),
                                      self.obs.data_ptr(), self.seed, auto_reset, sv["checksum"].data_ptr(), sv["status"].data_ptr(), timeout_s)
        sv["tag"] = (sv["tag"] + ticks) % 0xFFFFFF
        if not sync:
            return None
        torch.cuda.synchronize(d)
        res = sv["results"].permute(0, 2, 1).reshape(8, n)      # granule k of env i (stored as pairs: uint64[4][N][2])
        low = (res & 0xFFFFFFFF).to(torch.int32).view(torch.float32)
        self.reward.copy_(low[6])
        self.done.copy_(((res[6] >> 32) & 1).to(torch.uint8))
        self.zero_start.copy_(((res[6] >> 33) & 1).to(torch.uint8))
        return {"obs": self.obs, "obs_from_granules": low[:6].t().contiguous(), "reward": self.reward, "done": self.done,
                "zero_start": self.zero_start, "checksum": sv["checksum"], "status": sv["status"].cpu().numpy().astype("uint32")}

    def serve_with_policy(self, act_fn, ticks: int, auto_reset: bool = True, timeout_s: float = 2.0):
        """Experimental: `ticks` ticks on the resident tick server with a TORCH producer in a policy's place.  act_fn(obs, t) ->
        (keys uint8 (N,), mouse float32 (N,)) is ordinary torch code; it runs on a high-priority side stream between
        q1env_step_persistent_collect (tick t-1's results -> obs / reward / done tensors) and q1env_step_persistent_publish (tick t's
        action), while the env side is ONE launch for the whole run.  Bit-identical to `obs = observe(); for t: step_autoreset(act_fn(obs, t))`.
        Returns (reward (T,N) float32, done (T,N) uint8, status uint32[5]); self.obs / reward / done hold the last tick."""
        n, d = self.num_envs, self.device
        if not hasattr(self, "_srv"):
            self._srv = {"mailbox": torch.zeros((n,), dtype=torch.int64, device=d), "results": torch.zeros((4, n, 2), dtype=torch.int64, device=d),
                         "status": torch.zeros((5,), dtype=torch.int32, device=d), "checksum": torch.zeros((2, n), dtype=torch.float64, device=d),
                         "stream": torch.cuda.Stream(device=d, priority=-1), "tag": 0}
        sv = self._srv
        sv["status"].zero_()
        obs = self.observe().clone()                           # the observation the first action is computed from
        rew = torch.empty((ticks, n), dtype=torch.float32, device=d)
        don = torch.empty((ticks, n), dtype=torch.uint8, device=d)
        cur, side = torch.cuda.current_stream(d), sv["stream"]
        side.wait_stream(cur)
        tag0 = sv["tag"]
        self._dev.persistent_start(ticks, tag0, sv["mailbox"].data_ptr(), sv["results"].data_ptr(), 0, self.seed, auto_reset,
                                   sv["status"].data_ptr(), timeout_s)
        with torch.cuda.stream(side):
            for t in range(ticks):
                keys, mouse = act_fn(obs, t)
                assert keys.dtype == torch.uint8 and mouse.dtype == torch.float32 and keys.is_contiguous() and mouse.is_contiguous()
                self._dev.persistent_publish(side.cuda_stream, tag0, t, keys.data_ptr(), mouse.data_ptr(), sv["mailbox"].data_ptr())
                self._dev.persistent_collect(side.cuda_stream, tag0, t, sv["results"].data_ptr(), obs.data_ptr(), rew[t].data_ptr(),
                                             don[t].data_ptr(), self.zero_start.data_ptr(), sv["status"].data_ptr(), timeout_s)
        sv["tag"] = (tag0 + ticks) % 0xFFFFFF
        cur.wait_stream(side)
        torch.cuda.synchronize(d)
        self.obs.copy_(obs)
        self.reward.copy_(rew[-1])
        self.done.copy_(don[-1])
        return rew, don, sv["status"].cpu().numpy().astype("uint32")

    def observe(self) -> torch.Tensor:
        self._dev.observe_dev(self.obs.data_ptr(), _lib.OBS_F32)
        return self.obs

    def state_tensors(self):
        """Zero-copy torch views of the live SoA state arrays (dict name -> tensor)."""
        p = self._dev.device_ptrs()
        n = self.num_envs
        spec = {"vel_x": ("<f4", (n,)), "vel_y": ("<f4", (n,)), "vel_z": ("<f4", (n,)), "pos_x": ("<f8", (n,)), "pos_y": ("<f8", (n,)),
                "z_pos": ("<f8", (n,)), "yaw": ("<f8", (n,)), "time_remaining": ("<f8", (n,)),
                "last_key_press_time": ("<f8", (4, n)), "flags": ("|u1", (n,))}
        return {k: torch.as_tensor(_DevArray(p[k], shape, ts), device=self.device) for k, (ts, shape) in spec.items()}

    def get_state(self):
        torch.cuda.current_stream(self.device).synchronize()
        return self._dev.get_state()

    def set_state(self, **arrays):
        self._dev.set_state(**arrays)

    def sync(self):
        self._dev.sync()

    def close(self):
        self._dev.close()
