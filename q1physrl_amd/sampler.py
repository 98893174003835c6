"""GPU-resident sampler loop ("next" row 1 of SURVEY.md section 8f): the counterpart of RLlib's per-worker
sampler with nothing leaving the device between ticks:

    obs (N,6) f32 --torch MLP--> logits (N,10) --q1env_policy_sample (HIP)--> packed action + logp
        --q1env_step (HIP)--> obs', reward, done --q1env_reset_philox(done_only) (HIP)--> next obs

Trajectories are stored tick-major ([T][N]...) in preallocated device tensors; episode statistics follow the
reference's metric hook (train.py:54-57: the return of zero-start episodes, `zero_start_total_reward`)."""
import torch

from . import _lib
from .tensor_env import TensorVectorEnv


class GpuSampler:
    def __init__(self, env: TensorVectorEnv, policy, horizon: int, autocast_dtype=None):
        self.env, self.policy, self.T = env, policy, int(horizon)
        n, d, t = env.num_envs, env.device, self.T
        self.obs = torch.empty((t + 1, n, 6), dtype=torch.float32, device=d)
        self.keys = torch.empty((t, n), dtype=torch.uint8, device=d)
        self.mouse = torch.empty((t, n), dtype=torch.float32, device=d)
        self.logp = torch.empty((t, n), dtype=torch.float32, device=d)
        self.value = torch.empty((t + 1, n), dtype=torch.float32, device=d)
        self.reward = torch.empty((t, n), dtype=torch.float32, device=d)
        self.done = torch.empty((t, n), dtype=torch.uint8, device=d)
        self.ep_return = torch.zeros((n,), dtype=torch.float64, device=d)
        self.autocast_dtype = autocast_dtype
        self.counter = 0
        # device-resident episode statistics: [episodes, zero_start_episodes, return_sum, zero_start_return_sum]
        self._stats = torch.zeros((4,), dtype=torch.float64, device=d)
        self.obs[0].copy_(env.reset())

    @torch.no_grad()
    def _forward(self, obs):
        if self.autocast_dtype is not None:
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                logits, value = self.policy(obs)
            return logits.float().contiguous(), value.float()
        logits, value = self.policy(obs)
        return logits.contiguous(), value

    @torch.no_grad()
    def collect(self, deterministic=False):
        """One horizon of T ticks for all envs, without a single host<->device synchronisation; returns the trajectory
        buffers (views, valid until the next collect)."""
        env, dev = self.env, self.env._dev
        zero = torch.zeros((), dtype=torch.float64, device=env.device)
        for t in range(self.T):
            logits, value = self._forward(self.obs[t])
            self.value[t].copy_(value)
            dev.policy_sample_dev(logits.data_ptr(), logits.shape[1], env.seed, self.counter, self.keys[t].data_ptr(),
                                  self.mouse[t].data_ptr(), self.logp[t].data_ptr(), deterministic)
            self.counter += 1
            # tick: reward / done / zero_start only - the observation comes from the reset kernel below, which writes
            # the row of EVERY env (fresh first observation for the envs it resets, current observation for the others)
            dev.step_dev(_lib.ACT_PACKED, self.keys[t].data_ptr(), self.mouse[t].data_ptr(), _lib.OBS_F32, 0,
                         self.reward[t].data_ptr(), self.done[t].data_ptr(), env.zero_start.data_ptr())
            # episode bookkeeping (train.py:54-57: return of finished episodes, split by zero_start), all on device
            self.ep_return += self.reward[t]
            fin = self.done[t].bool()
            zs = fin & env.zero_start.bool()
            finished_ret = torch.where(fin, self.ep_return, zero)
            self._stats += torch.stack([fin.sum(), zs.sum(), finished_ret.sum(), torch.where(zs, self.ep_return, zero).sum()])
            self.ep_return = torch.where(fin, zero, self.ep_return)
            dev.reset_philox_dev(env.seed, 0, True, _lib.OBS_F32, self.obs[t + 1].data_ptr())   # masked: done envs only
        _, v_last = self._forward(self.obs[self.T])
        self.value[self.T].copy_(v_last)
        out = {"obs": self.obs, "keys": self.keys, "mouse": self.mouse, "logp": self.logp, "value": self.value,
               "reward": self.reward, "done": self.done}
        self.obs[0].copy_(self.obs[self.T])
        return out

    @property
    def stats(self):
        e, z, r, zr = self._stats.tolist()          # the only synchronisation, on demand
        return {"episodes": int(e), "zero_start_episodes": int(z), "return_sum": r, "zero_start_return_sum": zr}

    def zero_start_total_reward_mean(self):
        z = self.stats["zero_start_episodes"]
        return self.stats["zero_start_return_sum"] / z if z else float("nan")

    def advantages(self, traj, gamma, lam):
        """GAE over the last collected trajectory on the device (q1env_gae): returns (adv, vtarg), each (T, N) float32."""
        adv = torch.empty_like(traj["reward"])
        vtarg = torch.empty_like(traj["reward"])
        self.env._dev.gae_dev(self.T, traj["reward"].data_ptr(), traj["value"].data_ptr(), traj["done"].data_ptr(), gamma, lam,
                              adv.data_ptr(), vtarg.data_ptr())
        return adv, vtarg
