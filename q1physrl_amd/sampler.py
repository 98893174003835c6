"""GPU-resident sampler loop ("next" row 1 of SURVEY.md section 8f): the counterpart of RLlib's per-worker
sampler with nothing leaving the device between ticks:

    obs (N,6) f32 --policy forward (torch modules, or the fused MFMA kernel writing straight into the trajectory rows)-->
    logits (N,10), value --q1env_sample_step (HIP, ONE launch: sample the action + log-prob, step, reset finished episodes,
    episode statistics)--> packed action, logp, reward, done, next obs (fresh first observation for the envs it reset)

A tick is two launches with the fused policy (q1env_policy_value_forward + q1env_sample_step); fused_tick=False keeps the
three separate kernels (q1env_policy_sample, q1env_step_autoreset, q1env_episode_stats) - same bits, for tests.

Trajectories are stored tick-major ([T][N]...) in preallocated device tensors; episode statistics follow the
reference's metric hook (train.py:54-57: the return of zero-start episodes, `zero_start_total_reward`).

The loop has no host<->device synchronisation and no host-side state that changes per tick (the Philox counter lives
in device memory), so with use_graph=True the whole horizon is captured once into a hipGraph (torch.cuda.CUDAGraph) and
replayed: a tick then costs its GPU time, not ~40 Python-driven launches.
"""
import torch

from . import _lib
from .tensor_env import TensorVectorEnv


class GpuSampler:
    def __init__(self, env: TensorVectorEnv, policy, horizon: int, autocast_dtype=None, use_graph: bool = False,
                 fused_tick: bool = True, resident: bool = False):
        """resident=True: the whole horizon is ONE dispatch (q1env_sample_resident: policy blocks with the network's weights in LDS
        and env blocks with the state in registers, talking through tagged granules) followed by one batched value-network
        forward over the stored observations - bit-identical trajectories, no per-tick launches.  Needs a FusedPolicyForward
        policy (continuous, discrete - up to 24 outputs - or no mouse); any batch size (q1env.h); use_graph is
        ignored.  Every wait inside the dispatch is bounded; a wave that gives up sets the status words and the rows after its last
        completed tick are NOT written, so collect() checks the status after every horizon (check_status=True; one 20-byte D2H copy
        at a point where the caller synchronises anyway) and raises instead of handing stale memory to the learner."""
        self.env, self.policy, self.T = env, policy, int(horizon)
        self.fused_tick = bool(fused_tick)
        self.resident = bool(resident)
        if self.resident and not hasattr(policy, "_mlp"):
            raise ValueError("GpuSampler(resident=True) needs a policy.FusedPolicyForward")
        n, d, t = env.num_envs, env.device, self.T
        self.obs = torch.empty((t + 1, n, 6), dtype=torch.float32, device=d)
        self.keys = torch.empty((t, n), dtype=torch.uint8, device=d)
        self.mouse = torch.empty((t, n), dtype=torch.float32, device=d)
        self.logp = torch.empty((t, n), dtype=torch.float32, device=d)
        from .policy import policy_row_width
        width = policy_row_width(env.num_keys, env.config.discrete_yaw_steps, env.config.allow_yaw)
        self.logits = torch.empty((t, n, width), dtype=torch.float32, device=d)   # behaviour-policy outputs
        self.value = torch.empty((t + 1, n), dtype=torch.float32, device=d)
        self.reward = torch.empty((t, n), dtype=torch.float32, device=d)
        self.done = torch.empty((t, n), dtype=torch.uint8, device=d)
        self.ep_return = torch.zeros((n,), dtype=torch.float64, device=d)
        self.autocast_dtype = autocast_dtype
        self.tick = torch.zeros((1,), dtype=torch.int64, device=d)      # Philox counter, advanced on the device
        # device-resident episode statistics, one slot per wave of envs: [episodes, zero_start_episodes, return_sum,
        # zero_start_return_sum] (q1env_episode_stats; summed on the host on demand)
        self._stats = torch.zeros(((n + 63) // 64, 4), dtype=torch.float64, device=d)
        self.use_graph = bool(use_graph) and not self.resident
        self._graphs = {}
        if self.resident:
            self._status = torch.zeros((5,), dtype=torch.int32, device=d)       # written by the resident dispatch on failure only
        self.obs[0].copy_(env.reset())

    def _forward(self, obs):
        if self.autocast_dtype is not None:
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                logits, value = self.policy(obs)
            return logits.float().contiguous(), value.float()
        logits, value = self.policy(obs)
        return logits.contiguous(), value

    def _horizon(self, deterministic):
        """T ticks for all envs: no host<->device synchronisation, no per-tick host state (the RNG counter of tick t is
        the device-resident tick count + t; the count advances once per horizon)."""
        env, dev = self.env, self.env._dev
        cnt = self.tick.data_ptr()
        into = getattr(self.policy, "forward_into", None)
        for t in range(self.T):
            logits = self.logits[t]                   # what the actions are really sampled from (the learner's "old" policy)
            if into is not None:
                into(self.obs[t], logits, self.value[t])
            else:
                lg, value = self._forward(self.obs[t])
                self.value[t].copy_(value)
                logits.copy_(lg)
            if self.fused_tick:
                # sample + tick with in-kernel reset of finished episodes + episode bookkeeping (train.py:54-57): one kernel
                dev.sample_step_dev(logits.data_ptr(), logits.shape[1], env.seed, cnt, t, deterministic, self.keys[t].data_ptr(),
                                    self.mouse[t].data_ptr(), self.logp[t].data_ptr(), self.obs[t + 1].data_ptr(),
                                    self.reward[t].data_ptr(), self.done[t].data_ptr(), env.zero_start.data_ptr(),
                                    self.ep_return.data_ptr(), self._stats.data_ptr())
            else:
                dev.policy_sample_dev(logits.data_ptr(), logits.shape[1], env.seed, 0, self.keys[t].data_ptr(),
                                      self.mouse[t].data_ptr(), self.logp[t].data_ptr(), deterministic, counter_dev=cnt)
                dev.step_autoreset_dev(_lib.ACT_PACKED, self.keys[t].data_ptr(), self.mouse[t].data_ptr(), env.seed,
                                       self.obs[t + 1].data_ptr(), self.reward[t].data_ptr(), self.done[t].data_ptr(),
                                       env.zero_start.data_ptr(), counter_dev=cnt)
                dev.episode_stats_dev(self.reward[t].data_ptr(), self.done[t].data_ptr(), env.zero_start.data_ptr(),
                                      self.ep_return.data_ptr(), self._stats.data_ptr())
                self.tick.add_(1)
        if self.fused_tick:
            self.tick.add_(self.T)
        if into is not None:
            into(self.obs[self.T], self._scratch_logits(), self.value[self.T])
        else:
            _, v_last = self._forward(self.obs[self.T])
            self.value[self.T].copy_(v_last)

    def _horizon_resident(self, deterministic, timeout_s=5.0):
        """T ticks as one dispatch + the value network over the T + 1 stored observation rows as one batched launch."""
        env, dev = self.env, self.env._dev
        n, t = env.num_envs, self.T
        pi = self.policy._mlp("pi", self.logits.view(t * n, -1))
        dev.sample_resident_dev(t, pi, env.seed, self.tick.data_ptr(), 0, deterministic, self.keys.data_ptr(),
                                self.mouse.data_ptr(), self.logp.data_ptr(), self.obs.data_ptr(),
                                self.reward.data_ptr(), self.done.data_ptr(), env.zero_start.data_ptr(), self.ep_return.data_ptr(),
                                self._stats.data_ptr(), self._status.data_ptr(), timeout_s)
        self.tick.add_(t)
        dev.policy_forward_rows_dev((t + 1) * n, self.obs.data_ptr(), self.policy._mlp("vf", self.value.view((t + 1) * n, 1)))

    def resident_status(self):
        """uint32[5] as q1env_step_persistent_*: all zero = every wave served / handed over every tick of every horizon so far."""
        return self._status.cpu().numpy().astype("uint32")

    def check_resident_status(self):
        """Raise if any env / policy wave of a resident horizon gave up (time-out): trajectory rows after its last completed tick
        are stale memory.  Synchronises with the dispatch (a 20-byte copy)."""
        if self.resident:
            st = self.resident_status()
            if st.any():
                raise RuntimeError(f"resident sampler: a wave timed out, the trajectory of the last horizon is incomplete "
                                   f"(status words {st.tolist()}; see q1env_sample_resident in include/q1env.h)")

    def _scratch_logits(self):
        if not hasattr(self, "_scratch"):
            self._scratch = torch.empty_like(self.logits[0])
        return self._scratch

    @torch.no_grad()
    def collect(self, deterministic=False, check_status=True):
        """One horizon; returns the trajectory buffers (views, valid until the next collect).  obs[T] is carried over
        to obs[0] at the START of the next collect, so the returned buffers are complete (T+1 observation rows).
        resident=True: check_status=True (default) reads the dispatch's status words back and raises on a time-out;
        pass False in latency-critical loops and call check_resident_status() before the trajectory is consumed."""
        if getattr(self, "_carry", False):
            self.obs[0].copy_(self.obs[self.T])
        self._carry = True
        if self.resident:
            self._horizon_resident(deterministic)
            if check_status:
                self.check_resident_status()
        elif not self.use_graph:
            self._horizon(deterministic)
        else:
            key = bool(deterministic)
            if key not in self._graphs:
                self._capture(key)
            else:
                self._graphs[key].replay()
        return {"obs": self.obs, "keys": self.keys, "mouse": self.mouse, "logp": self.logp, "logits": self.logits,
                "value": self.value, "reward": self.reward, "done": self.done}

    def _capture(self, deterministic):
        """Capture the horizon into a hipGraph.  Capture only records: the state a replay starts from must be the state
        the eager path would start from, so everything the horizon mutates (env state, counters, statistics) is saved
        before a warm-up pass (hipBLASLt workspaces, autotuning) and restored, and the graph is replayed once for real."""
        env = self.env
        torch.cuda.synchronize(env.device)
        saved_env = env.get_state()
        saved = [x.clone() for x in (self.tick, self._stats, self.ep_return, self.obs[0])]
        side = torch.cuda.Stream(device=env.device)
        with torch.cuda.stream(side):
            env.use_current_stream()
            self._horizon(deterministic)                      # warm-up on the side stream
        torch.cuda.synchronize(env.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
            env.use_current_stream()
            self._horizon(deterministic)
        env.use_current_stream()                              # back on the caller's stream for replays and later calls
        torch.cuda.synchronize(env.device)
        env.set_state(**saved_env)
        for dst, src in zip((self.tick, self._stats, self.ep_return, self.obs[0]), saved):
            dst.copy_(src)
        self._graphs[deterministic] = g
        g.replay()

    def advantages(self, traj, gamma, lam):
        """GAE over the last collected trajectory on the device (q1env_gae): returns (adv, vtarg), each (T, N) float32."""
        adv = torch.empty_like(traj["reward"])
        vtarg = torch.empty_like(traj["reward"])
        self.env._dev.gae_dev(self.T, traj["reward"].data_ptr(), traj["value"].data_ptr(), traj["done"].data_ptr(), gamma, lam,
                              adv.data_ptr(), vtarg.data_ptr())
        return adv, vtarg

    @property
    def stats(self):
        e, z, r, zr = self._stats.sum(dim=0).tolist()          # the only synchronisation, on demand
        return {"episodes": int(e), "zero_start_episodes": int(z), "return_sum": r, "zero_start_return_sum": zr}

    def zero_start_total_reward_mean(self):
        s = self.stats
        return s["zero_start_return_sum"] / s["zero_start_episodes"] if s["zero_start_episodes"] else float("nan")
