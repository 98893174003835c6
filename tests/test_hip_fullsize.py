"""GPU, driver-run: BASELINE.json configs[3] and configs[4] at their stated sizes.

  configs[3]  1 048 576 envs sharded over 8 GPUs (131 072 per GPU): the LAST shard (env_index_base = 7 x 131 072), built as its
              own handle the way rank 7 builds it, must equal the tail of one 1 048 576-env handle - fused rollout with on-device
              random actions and in-kernel resets, and the per-tick step kernels with packed actions.
  configs[4]  262 144 envs driving the sampler loop (fused matrix-core policy forward + q1env_sample_step, the whole horizon in
              one hipGraph): a 64-env slice of the stored trajectory is replayed tick by tick through the NumPy env oracle
              (bit-exact) and the stored actions / log-probabilities are re-derived from the stored logits by oracle/dist_oracle.py
              on the same Philox draws.
"""
import dataclasses

import numpy as np
import pytest

from oracle import dist_oracle as DO
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

PARAMS_YML = dict(action_range=10, allow_jump=True, allow_yaw=True, auto_jump=False, discrete_yaw_steps=-1, fmove_max=800,
                  smove_max=1060, hover=False, initial_yaw_range=(0, 360), key_press_delay=0.3, max_initial_speed=700,
                  smooth_keys=True, speed_reward=False, time_delta=0.013888888888888, time_limit=10, zero_start_prob=0.01)


def oracle_from_state(cfg, st, sl):
    """An OracleVectorEnv holding the envs `sl` of a device state dict (the inverse of test_hip_fastpath.inject)."""
    m = len(range(*sl.indices(st["flags"].shape[0])))
    np.random.seed(0)
    ora = O.OracleVectorEnv(dataclasses.replace(cfg, num_envs=m))
    f = st["flags"][sl]
    ora.st = {"vel": np.stack([st["vel_x"][sl], st["vel_y"][sl], st["vel_z"][sl]], axis=1).astype(np.float32),
              "z_pos": st["z_pos"][sl].astype(np.float64), "on_ground": (f & 1) != 0, "jump_released": (f & 2) != 0}
    ora.yaw, ora.t_rem, ora.zero_start = st["yaw"][sl].copy(), st["time_remaining"][sl].copy(), (f & 4) != 0
    k = cfg.num_keys
    ora.dec = {"last_press": st["last_key_press_time"][sl][:, :k].copy(),
               "last_keys": ((f[:, None] >> (3 + np.arange(k))[None, :]) & 1) != 0, "yaw": ora.yaw}
    return ora


def test_config4_last_shard_of_1048576_envs_equals_tail_of_one_handle():
    import torch
    from q1physrl_amd.env import Config
    from q1physrl_amd.sharding import shard_plan
    from q1physrl_amd.tensor_env import TensorVectorEnv
    total, world, seed = 1048576, 8, 17
    start, count = shard_plan(total, world)[7]
    assert (start, count) == (7 * 131072, 131072)
    base = O.OracleConfig.get_default(num_envs=total, zero_start_prob=0.25)         # get_default Config, mixed starts
    ticks_a, ticks_b = 800, 96                                                     # > one 720-tick episode, then per-tick kernels
    g = torch.Generator(device="cuda").manual_seed(5)
    keys = torch.randint(0, 16, (ticks_b, total), dtype=torch.uint8, device="cuda", generator=g)
    mouse = (torch.rand((ticks_b, total), device="cuda", generator=g) * 2 - 1) * float(np.float32(base.action_range))

    def run(n, first):
        e = TensorVectorEnv(Config(**dataclasses.replace(base, num_envs=n).__dict__), seed=seed, env_index_base=first)
        e.reset()
        ret = torch.zeros((n,), dtype=torch.float64, device="cuda")
        e.rollout(ticks_a, None, outputs=False, auto_reset=True, return_sum=ret)   # on-device Philox actions keyed by GLOBAL env index
        obs = rew = done = None
        for t in range(ticks_b):                                                    # per-tick step + in-kernel reset, packed actions
            obs, rew, done = e.step_autoreset((keys[t, first:first + n].contiguous(), mouse[t, first:first + n].contiguous()))
        torch.cuda.synchronize()
        out = (e.get_state(), ret.cpu().numpy(), obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), done.cpu().numpy().copy())
        e.close()
        return out
    full_st, full_ret, full_obs, full_rew, full_done = run(total, 0)
    sh_st, sh_ret, sh_obs, sh_rew, sh_done = run(count, start)
    for k in full_st:
        assert np.array_equal(full_st[k][start:], sh_st[k]), k
    assert np.array_equal(full_ret[start:], sh_ret) and np.array_equal(full_obs[start:], sh_obs)
    assert np.array_equal(full_rew[start:], sh_rew) and np.array_equal(full_done[start:], sh_done)
    # the run did something: every env finished at least one episode, starts are out of phase, returns are spread
    assert full_st["time_remaining"].min() >= -0.02 and len(np.unique(sh_st["time_remaining"])) > 1000
    assert np.isfinite(sh_ret).all() and np.abs(sh_ret).max() > 50.0
    # ... and the first shard differs from the last (the RNG really is keyed by the global index)
    assert not np.array_equal(full_st["yaw"][:count], sh_st["yaw"])


def test_config5_sampler_262144_envs_slice_replayed_by_the_oracles():
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.env import Config
    from q1physrl_amd.sampler import GpuSampler
    from q1physrl_amd.tensor_env import TensorVectorEnv
    n, T, seed, base = 262144, 24, 31, 3 * 262144          # as the 4th of several 262 144-env sampler shards would be keyed
    cfg = O.OracleConfig(num_envs=n, **PARAMS_YML)
    env = TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=seed, env_index_base=base)
    torch.manual_seed(2)
    pol = P.Q1Policy().cuda()
    with torch.no_grad():                                    # a policy with opinions: near-uniform logits would test little
        pol.pi[-1].weight.mul_(80.0)
        pol.pi[-1].bias.normal_(0, 0.5)
    fused = P.FusedPolicyForward(pol, env)
    s = GpuSampler(env, fused, horizon=T, use_graph=True)
    st0 = env.get_state()
    tr = s.collect()                                         # capture + first replay: starts from st0, RNG counter 0
    torch.cuda.synchronize()
    assert int(s.tick.item()) == T
    pick = slice(131072 + 7, 131072 + 7 + 64)                # a slice in the middle of the batch, not wave-aligned
    m = 64
    ora = oracle_from_state(cfg, st0, pick)
    keys = tr["keys"][:, pick].cpu().numpy()
    mouse = tr["mouse"][:, pick].cpu().numpy()
    logits = tr["logits"][:, pick].cpu().numpy()
    logp = tr["logp"][:, pick].cpu().numpy()
    obs = tr["obs"][:, pick].cpu().numpy()
    rew, done = tr["reward"][:, pick].cpu().numpy(), tr["done"][:, pick].cpu().numpy()
    assert np.array_equal(obs[0], ora.observation().astype(np.float32))
    assert not done.any()                                    # starts have >= 1 s left, the horizon is 1/3 s: no reset inside the slice
    genv = np.arange(pick.start, pick.stop, dtype=np.uint64) + np.uint64(base)
    low, high = -float(np.float32(cfg.action_range)), float(np.float32(cfg.action_range))
    sure_total = 0
    for t in range(T):
        # (a) the env: stored action -> oracle tick == stored reward / done / next observation, bit for bit
        a = np.concatenate([((keys[t][:, None] >> np.arange(4)) & 1).astype(np.float64), mouse[t][:, None].astype(np.float64)], axis=1)
        o, r, d, _ = ora.vector_step(a)
        assert np.array_equal(r, rew[t]) and np.array_equal(d, done[t].astype(bool)), t
        assert np.array_equal(o.astype(np.float32), obs[t + 1]), t
        # (b) the distribution: same Philox draws (counter = tick index) through the float64 restatement
        k2, m2, lp2, margin = DO.sample_from_philox(cfg, logits[t], seed, genv, t)
        sure = margin > 1e-6
        sure_total += int(sure.sum())
        assert np.array_equal(keys[t][sure], k2[sure]), t
        assert np.max(np.abs(mouse[t] - m2)) < 1e-4, t
        # (c) the stored log-probability re-derived for the STORED action from the stored logits (no sampling involved)
        lp3 = DO.mouse_logp(mouse[t].astype(np.float64), logits[t][:, 8].astype(np.float64), logits[t][:, 9].astype(np.float64), low, high)
        for j in range(4):
            lp0, lp1 = DO.key_logprobs(logits[t][:, 2 * j].astype(np.float64), logits[t][:, 2 * j + 1].astype(np.float64))
            lp3 = lp3 + np.where((keys[t] >> j) & 1, lp1, lp0)
        inner = np.abs(mouse[t]) < 0.99 * high               # away from the 1e-6 clip, where ndtri amplifies float32 rounding
        assert np.max(np.abs(logp[t][inner] - lp3[inner]) / np.maximum(np.abs(lp3[inner]), 1.0)) < 1e-3, t
    assert sure_total > 0.999 * T * m
    # (d) the fused matrix-core forward at this size against the float32 torch modules on the slice
    with torch.no_grad():
        lg32, v32 = pol(tr["obs"][0, pick])
    assert float((lg32 - tr["logits"][0, pick]).abs().max()) < 2e-2 * max(1.0, float(lg32.abs().max()))
    assert float((v32 - tr["value"][0, pick]).abs().max()) < 2e-2 * max(1.0, float(v32.abs().max()))
    # (e) whole-batch sanity of the horizon: finite, every key used, episode bookkeeping consistent with the stored done flags
    assert torch.isfinite(tr["logp"]).all() and torch.isfinite(tr["obs"]).all()
    assert s.stats["episodes"] == int(tr["done"].sum().item())
    # a second replay of the captured graph continues the same episodes with fresh randomness (counter advanced on the device)
    k_first = tr["keys"].clone()
    tr2 = s.collect()
    torch.cuda.synchronize()
    assert int(s.tick.item()) == 2 * T and not torch.equal(tr2["keys"], k_first)
    env.close()
