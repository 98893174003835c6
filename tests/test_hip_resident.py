"""GPU: the resident sampler (q1env_sample_resident + q1env_policy_forward_rows: a sampling horizon as ONE dispatch + one batched
value forward) against the two-launch-per-tick sampler it replaces - every trajectory tensor, the env state, the episode statistics and
the Philox counter must be identical, bit for bit, over several horizons (resets, ragged batches, 3-key / no-mouse / generic Configs,
one and two tiles per policy wave, deterministic actions), and its failure mode must be a bounded time-out."""
import numpy as np
import pytest

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def make_env(n, seed, **over):
    from q1physrl_amd.env import Config
    from q1physrl_amd.tensor_env import TensorVectorEnv
    cfg = O.OracleConfig.get_default(num_envs=n, **over)
    return cfg, TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=seed)


def run_sampler(n, horizons, T, resident, over, deterministic=False, policy_seed=0):
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.sampler import GpuSampler
    torch.manual_seed(policy_seed)
    cfg, env = make_env(n, seed=9, **over)
    pol = P.Q1Policy(num_keys=env.num_keys, allow_yaw=cfg.allow_yaw, discrete_yaw_steps=cfg.discrete_yaw_steps).cuda()
    with torch.no_grad():                                          # (weights large enough for the actions to depend on the observation)
        for p_ in pol.parameters():
            p_.mul_(3.0)
    fused = P.FusedPolicyForward(pol, env)
    s = GpuSampler(env, fused, horizon=T, resident=resident)
    runs = []
    for _ in range(horizons):
        tr = s.collect(deterministic=deterministic)
        torch.cuda.synchronize()
        runs.append({k: v.clone() for k, v in tr.items()})
    out = (runs, s.stats, env.get_state(), s.tick.clone(), s.ep_return.clone(), env.zero_start.clone(),
           s.resident_status() if resident else None)
    env.close()
    return out


@pytest.mark.parametrize("n,T,over,det", [
    (2048, 40, dict(zero_start_prob=0.5, time_limit=0.3), False),                       # one tile per policy wave, resets
    (4096 + 37, 24, dict(zero_start_prob=0.3, time_limit=0.2), False),                  # ragged: dead lanes in env and policy waves
    (32768, 16, dict(zero_start_prob=0.1, time_limit=0.15), False),                     # BASELINE configs[4]'s shard
    (65536, 8, dict(zero_start_prob=1.0), False),                                       # two tiles per policy wave
    (1000, 30, dict(time_limit=0.25, allow_jump=False), False),                         # three keys (8 logits + mouse)
    (1500, 30, dict(time_limit=0.25, allow_yaw=False), False),                          # no mouse
    (777, 30, dict(time_limit=0.25, auto_jump=True, speed_reward=True), False),         # generic (SPEC = false) kernels
    (2048, 30, dict(zero_start_prob=0.5, time_limit=0.3), True),                        # deterministic actions
    (3000, 30, dict(time_limit=0.25, zero_start_prob=0.4, discrete_yaw_steps=5), False),        # discrete mouse: 19 logits, categorical head
    (900, 30, dict(time_limit=0.25, discrete_yaw_steps=7, hover=True), False),                  # 23 logits, generic kernels
    (1300, 24, dict(time_limit=0.25, discrete_yaw_steps=1, allow_jump=False), True),            # 3 keys + a 3-way mouse: 9 logits, still the discrete variant
    (32768, 8, dict(zero_start_prob=0.3, time_limit=0.1, discrete_yaw_steps=5), False),         # the wide variant's co-resident capacity
    # round 4: more workgroups than CUs - the workgroups share nothing, so the grid runs as successive sets (before: refused)
    (40000, 6, dict(zero_start_prob=0.3, time_limit=0.05, discrete_yaw_steps=5), False),         # wide variant, 313 workgroups, ragged
    ((1 << 17) + 77, 6, dict(zero_start_prob=0.5, time_limit=0.05), False),                     # 513 workgroups of 256 envs, ragged tail
    (262144, 4, dict(zero_start_prob=0.1, time_limit=0.05), False),                             # BASELINE configs[4] on ONE GPU: 1 024 workgroups
])
def test_resident_sampler_equals_the_two_launch_sampler(n, T, over, det):
    import torch
    ref = run_sampler(n, 3, T, False, over, det)
    res = run_sampler(n, 3, T, True, over, det)
    assert res[6] is not None and not res[6].any(), res[6]
    for h in range(3):
        for k in ("obs", "keys", "mouse", "logp", "logits", "value", "reward", "done"):
            a, b = ref[0][h][k], res[0][h][k]
            assert a.shape == b.shape and torch.equal(a, b), (h, k, float((a != b).float().mean()))
    assert ref[1] == res[1]                                          # episode statistics
    if not det:
        assert not torch.equal(res[0][0]["keys"], res[0][1]["keys"])
    for k in ref[2]:
        assert np.array_equal(ref[2][k], res[2][k]), k              # env state
    assert torch.equal(ref[3], res[3]) and torch.equal(ref[4], res[4]) and torch.equal(ref[5], res[5])
    assert int(ref[0][0]["done"].sum()) > 0 or over.get("zero_start_prob") == 1.0


def test_resident_sampler_refuses_what_it_cannot_run():
    import torch
    from q1physrl_amd import _lib, policy as P
    from q1physrl_amd.sampler import GpuSampler
    cfg, env = make_env(512, seed=1, discrete_yaw_steps=10)
    pol = P.Q1Policy(discrete_yaw_steps=10).cuda()                  # 8 + 21 = 29 outputs: more than the resident workgroup's LDS holds
    s = GpuSampler(env, P.FusedPolicyForward(pol, env), horizon=4, resident=True)
    with pytest.raises(_lib.Q1EnvError, match="more than 24"):
        s.collect()
    env.close()
    with pytest.raises(ValueError, match="FusedPolicyForward"):
        cfg, env = make_env(512, seed=1)
        GpuSampler(env, P.Q1Policy().cuda(), horizon=4, resident=True)


def test_policy_forward_rows_equals_the_per_batch_forward():
    """q1env_policy_forward_rows over T stacked observation batches == T q1env_policy_value_forward calls, bit for bit (what the
    resident sampler's batched value forward relies on)."""
    import torch
    from q1physrl_amd import policy as P
    n, T = 3000, 5
    cfg, env = make_env(n, seed=2)
    torch.manual_seed(1)
    pol = P.Q1Policy().cuda()
    with torch.no_grad():
        for p_ in pol.parameters():
            p_.mul_(2.0)
    fused = P.FusedPolicyForward(pol, env)
    obs = (torch.rand((T, n, 6), device="cuda") * 2 - 1).contiguous()
    per_batch_l, per_batch_v = [], []
    for t in range(T):
        lg, v = fused(obs[t])
        per_batch_l.append(lg.clone()); per_batch_v.append(v.clone())
    value = torch.empty((T * n, 1), device="cuda")
    logits = torch.empty((T * n, 10), device="cuda")
    env._dev.policy_forward_rows_dev(T * n, obs.data_ptr(), fused._mlp("vf", value))
    env._dev.policy_forward_rows_dev(T * n, obs.data_ptr(), fused._mlp("pi", logits))
    torch.cuda.synchronize()
    assert torch.equal(value.view(T, n), torch.stack(per_batch_v)) and torch.equal(logits.view(T, n, 10), torch.stack(per_batch_l))
    env.close()


def test_resident_sampler_shard_equals_the_slice_of_the_whole_batch():
    """Multi-GPU is a batch split: the trajectories a shard (env_index_base = first global env) samples are the corresponding slice
    of the whole batch's - the Philox draws of sampling and resets are keyed by the GLOBAL env index, the workgroup boundaries are not."""
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.env import Config
    from q1physrl_amd.sampler import GpuSampler
    from q1physrl_amd.tensor_env import TensorVectorEnv
    n, T, base, count = 6000, 20, 2100, 1700                       # a shard that starts and ends inside workgroups of the whole batch
    over = dict(time_limit=0.2, zero_start_prob=1.0)               # (zero starts: the initial state is the same everywhere)
    outs = []
    for num, b in ((n, 0), (count, base)):
        cfg = O.OracleConfig.get_default(num_envs=num, **over)
        env = TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=4, env_index_base=b)
        torch.manual_seed(3)
        pol = P.Q1Policy().cuda()
        with torch.no_grad():
            for p_ in pol.parameters():
                p_.mul_(3.0)
        s = GpuSampler(env, P.FusedPolicyForward(pol, env), horizon=T, resident=True)
        runs = [{k: v.clone() for k, v in s.collect().items()} for _ in range(2)]
        torch.cuda.synchronize()
        assert not s.resident_status().any()
        outs.append(runs)
        env.close()
    whole, shard = outs
    for h in range(2):
        for k in ("obs", "keys", "mouse", "logp", "logits", "value", "reward", "done"):
            assert torch.equal(whole[h][k][:, base:base + count], shard[h][k]), (h, k)
    assert int(whole[1]["done"].sum()) > 0


def test_collect_raises_when_a_resident_wave_reports_a_time_out():
    """a resident horizon whose status words are non-zero left trajectory rows unwritten; collect() must raise instead of
    handing stale memory to the learner.  (The status words are forced here: a real time-out needs a dead wave.)"""
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.env import Config
    from q1physrl_amd.sampler import GpuSampler
    from q1physrl_amd.tensor_env import TensorVectorEnv
    env = TensorVectorEnv(Config(**dict(Config.get_default().__dict__, num_envs=512)), device=0, seed=2)
    s = GpuSampler(env, P.FusedPolicyForward(P.Q1Policy().cuda(), env), horizon=4, resident=True)
    s.collect()                                   # a healthy horizon: no exception, status all zero
    assert not s.resident_status().any()
    s._status[1] = 1
    s._status[2] = 3
    with pytest.raises(RuntimeError, match="timed out"):
        s.collect()
    s._status.zero_()
    s.collect(check_status=False)
    s.check_resident_status()
    env.close()
