"""GPU: the fused policy-sampling kernel (q1env_policy_sample) against the float64 NumPy/SciPy restatement fed with the
same Philox draws, the torch distribution, and an end-to-end sampler loop."""
import os

import numpy as np
import pytest

from oracle import dist_oracle as DO
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def make_env(n, seed=5, base=0, **over):
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    cfg = O.OracleConfig.get_default(num_envs=n, **over)
    return cfg, TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=seed, env_index_base=base)


def test_policy_sample_kernel_matches_restatement():
    import torch
    n, seed, counter, base = 50_000, 5, 17, 123_456_789_012
    cfg, env = make_env(n, seed, base)
    rng = np.random.default_rng(1)
    logits = rng.normal(0, 1.5, (n, 10)).astype(np.float32)
    logits[:, 8] = rng.uniform(-4, 4, n)
    logits[:, 9] = rng.uniform(-3, 2.5, n)
    lt = torch.from_numpy(logits).cuda()
    keys = torch.empty(n, dtype=torch.uint8, device="cuda")
    mouse = torch.empty(n, dtype=torch.float32, device="cuda")
    logp = torch.empty(n, dtype=torch.float32, device="cuda")
    env._dev.policy_sample_dev(lt.data_ptr(), 10, seed, counter, keys.data_ptr(), mouse.data_ptr(), logp.data_ptr())
    torch.cuda.synchronize()
    k2, m2, lp2, margin = DO.sample_from_philox(cfg, logits, seed, np.arange(n, dtype=np.uint64) + np.uint64(base), counter)
    sure = margin > 1e-6                    # a uniform within 1e-6 of its float32 threshold may legitimately flip
    assert sure.mean() > 0.9999 and np.array_equal(keys.cpu().numpy()[sure], k2[sure])
    # float32 kernel vs float64 restatement: absolute tolerance on the action (range 20), relative on logp
    # (measured on MI355X, 200 000 rows: 0 key mismatches, |mouse diff| <= 5.3e-5, logp 2.9e-4 / 2.6e-5 relative)
    assert np.max(np.abs(mouse.cpu().numpy() - m2)) < 1e-4
    lp = logp.cpu().numpy()
    same = sure & (np.abs(m2) < 10.0)       # away from the 1e-6 clip, where ndtri amplifies float32 rounding
    assert np.max(np.abs(lp[same] - lp2[same]) / np.maximum(np.abs(lp2[same]), 1.0)) < 6e-4
    inner = sure & (np.abs(m2) < 9.5)
    assert np.max(np.abs(lp[inner] - lp2[inner]) / np.maximum(np.abs(lp2[inner]), 1.0)) < 1e-4
    # torch distribution agrees with the kernel's logp on the kernel's own samples
    from q1physrl_amd import policy as P
    dist = P.Q1PhysActionDist(lt.double(), float(np.float32(cfg.action_range)), 4)
    kbits = ((keys[:, None].long() >> torch.arange(4, device="cuda")) & 1)
    lp_t = dist.logp(kbits, mouse.double().view(-1, 1)).cpu().numpy()
    assert np.max(np.abs(lp[same] - lp_t[same]) / np.maximum(np.abs(lp_t[same]), 1.0)) < 6e-4
    # statistics: empirical key frequency equals sigmoid(l1 - l0) on average
    p1 = 1 / (1 + np.exp(-(logits[:, 1].astype(np.float64) - logits[:, 0])))
    assert abs(((keys.cpu().numpy() & 1) != 0).mean() - p1.mean()) < 0.01
    # deterministic mode: argmax keys, squashed mean
    env._dev.policy_sample_dev(lt.data_ptr(), 10, seed, counter, keys.data_ptr(), mouse.data_ptr(), logp.data_ptr(), True)
    torch.cuda.synchronize()
    kd, md, _, _ = DO.sample_from_philox(cfg, logits, seed, np.arange(n, dtype=np.uint64), counter, deterministic=True)
    assert np.array_equal(keys.cpu().numpy(), kd) and np.max(np.abs(mouse.cpu().numpy() - md)) < 1e-4
    env.close()


def test_sampler_loop_end_to_end():
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.sampler import GpuSampler
    torch.manual_seed(0)
    cfg, env = make_env(4096, seed=9, zero_start_prob=0.5, time_limit=1.0)
    pol = P.Q1Policy().cuda()
    s = GpuSampler(env, pol, horizon=100)
    traj = s.collect()
    torch.cuda.synchronize()
    assert traj["obs"].shape == (101, 4096, 6) and traj["reward"].shape == (100, 4096)
    assert torch.isfinite(traj["logp"]).all() and torch.isfinite(traj["obs"]).all()
    # every stored transition is reproducible by the oracle: replay env 0..63 of the first episode segment
    assert s.stats["episodes"] >= 4096 and s.stats["zero_start_episodes"] > 0
    assert np.isfinite(s.zero_start_total_reward_mean())
    # q1env_episode_stats against a NumPy replay of the stored trajectory (zero_start of an episode = flag at its first obs:
    # obs column 0 (time left / limit) == 1 and yaw column == 1 identify zero starts only loosely, so use the env's own flags)
    rew = traj["reward"].cpu().numpy().astype(np.float64)
    dn = traj["done"].cpu().numpy().astype(bool)
    run = np.zeros(4096)
    episodes, ret_sum = 0, 0.0
    for t in range(100):
        run += rew[t]
        episodes += int(dn[t].sum())
        ret_sum += float(run[dn[t]].sum())
        run[dn[t]] = 0.0
    assert s.stats["episodes"] == episodes and abs(s.stats["return_sum"] - ret_sum) < 1e-6 * max(1.0, abs(ret_sum))
    assert 0 < s.stats["zero_start_episodes"] < episodes
    # stored actions really are what was applied: re-step a fresh env with the stored packed actions (no resets in 20 ticks)
    cfg2, env2 = make_env(4096, seed=9, zero_start_prob=0.5, time_limit=1.0)
    env2.reset()
    for t in range(20):
        o, r, d = env2.step_tensor((traj["keys"][t], traj["mouse"][t]))
        assert torch.equal(r, traj["reward"][t]) and torch.equal(d, traj["done"][t])
    env.close(); env2.close()


def test_gae_kernel_matches_numpy():
    import torch
    from oracle import ppo_oracle as PO
    from q1physrl_amd.sampler import GpuSampler
    from q1physrl_amd import policy as P
    cfg, env = make_env(3000, seed=2)
    rng = np.random.default_rng(0)
    t, n = 50, 3000
    s = GpuSampler(env, P.Q1Policy().cuda(), horizon=t)
    traj = {"reward": torch.from_numpy(rng.normal(0, 1, (t, n)).astype(np.float32)).cuda(),
            "value": torch.from_numpy(rng.normal(0, 1, (t + 1, n)).astype(np.float32)).cuda(),
            "done": torch.from_numpy((rng.random((t, n)) < 0.05).astype(np.uint8)).cuda()}
    adv, vtarg = s.advantages(traj, 0.99, 0.95)
    torch.cuda.synchronize()
    a2, v2 = PO.gae(traj["reward"].cpu().numpy(), traj["value"].cpu().numpy(), traj["done"].cpu().numpy(), 0.99, 0.95)
    assert np.allclose(adv.cpu().numpy(), a2, rtol=1e-5, atol=1e-5) and np.allclose(vtarg.cpu().numpy(), v2, rtol=1e-5, atol=1e-5)
    env.close()


def test_wr_policy_reproduces_published_reward_and_oracle_trajectory():
    """Known answer from the reference's published artefacts: its world-record policy (weights fixture
    tests/golden/wr_policy.npz, exported from data/checkpoints/wr by oracle/export_wr_weights.py) trained to
    zero_start_total_reward_mean ~ 5700 (README.md:54, stochastic policy).  Played in the HIP env under the run's own
    env_config it must score that, and the expert trajectory it produces (strafe-jumping, key rate-limiting in play)
    must be bit-identical to the oracle driven with the same actions."""
    import json
    import os
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.env import Config
    from q1physrl_amd.sampler import GpuSampler
    from q1physrl_amd.tensor_env import TensorVectorEnv
    w = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wr_policy.npz")))
    ec = json.loads(str(w["env_config_json"]))
    ec["initial_yaw_range"] = tuple(ec["initial_yaw_range"])
    pol = P.load_rllib_fcnet_weights(P.Q1Policy(), w).cuda()
    assert pol.num_parameters() == 137995
    cfg = Config(**{**ec, "num_envs": 2048, "zero_start_prob": 1.0})
    env = TensorVectorEnv(cfg, seed=7)
    tr = GpuSampler(env, pol, horizon=720).collect()
    total = tr["reward"].double().sum(0)
    assert bool(tr["done"][-1].all()) and not bool(tr["done"][:-1].any())
    assert 5600.0 < float(total.mean()) < 5800.0, float(total.mean())          # README: ~5700
    # oracle replay of the first 32 envs' expert trajectories
    k = 32
    keys = tr["keys"][:, :k].cpu().numpy()
    mouse = tr["mouse"][:, :k].cpu().numpy()
    ocfg = O.OracleConfig(**{**ec, "num_envs": k, "zero_start_prob": 1.0})
    np.random.seed(0)
    ora = O.OracleVectorEnv(ocfg)
    obs_gpu = tr["obs"][1:, :k].cpu().numpy()
    rew_gpu = tr["reward"][:, :k].cpu().numpy()
    for t in range(720):
        a = np.concatenate([((keys[t][:, None] >> np.arange(4)[None, :]) & 1).astype(np.float64), mouse[t][:, None].astype(np.float64)], axis=1)
        o, r, d, _ = ora.vector_step(a)
        assert np.array_equal(r, rew_gpu[t]), t
        if t < 719:                      # after the last tick the sampler's obs row is the post-reset observation
            assert np.array_equal(o.astype(np.float32), obs_gpu[t]), t
    assert float(np.sum(rew_gpu.astype(np.float64), axis=0).mean()) > 5600
    env.close()


def test_graph_captured_sampler_equals_eager_sampler():
    """The whole horizon captured into one hipGraph (device-resident Philox counter) must produce the same trajectories
    as the eager loop, on the first horizon AND on replays (fresh random numbers each replay, resets included)."""
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.sampler import GpuSampler
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        cfg, env = make_env(2048, seed=9, zero_start_prob=0.5, time_limit=0.5)
        pol = P.Q1Policy().cuda()
        s = GpuSampler(env, pol, horizon=48, use_graph=use_graph)
        runs = []
        for _ in range(3):
            tr = s.collect()
            torch.cuda.synchronize()
            runs.append({k: v.clone() for k, v in tr.items()})
        outs.append((runs, s.stats, env.get_state()))
        env.close()
    (eager, st_e, env_e), (graph, st_g, env_g) = outs
    for h in range(3):
        for k in ("keys", "reward", "done", "obs", "mouse"):
            same = (eager[h][k] == graph[h][k]).float().mean().item()
            assert same == 1.0, (h, k, same)
    assert st_e == st_g and st_e["episodes"] > 2048
    assert not torch.equal(eager[0]["keys"], eager[1]["keys"])           # replays draw fresh numbers
    for k in env_e:
        assert np.array_equal(env_e[k], env_g[k]), k


def test_eval_sim_with_wr_policy_through_the_compat_api():
    """analyse.eval_sim's protocol (one-env VectorPhysEnv + a parallel stand-alone ActionDecoder + player_state snapshots)
    over the NumPy-compatible surface, with the reference's WR policy: ~5750 units in 720 frames, and the move commands
    (yaw, smove, fmove, jump) it would send to the real game equal the oracle's decoder outputs frame for frame."""
    import json
    import os
    from q1physrl_amd import evaluate as EV, policy as P
    from q1physrl_amd.env import Config
    w = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wr_policy.npz")))
    ec = json.loads(str(w["env_config_json"]))
    ec["initial_yaw_range"] = tuple(ec["initial_yaw_range"])
    cfg = Config(**{**ec, "num_envs": 1, "zero_start_prob": 1.0})
    pol = P.load_rllib_fcnet_weights(P.Q1Policy(), w)
    res = EV.eval_sim(EV.TorchTrainer(pol, ec["action_range"]), cfg)
    assert res.obs.shape == (720, 6) and res.player_state.vel.shape == (720, 3) and res.smove.dtype == np.int64
    total = float(np.sum(res.reward.astype(np.float64)))
    assert 5600 < total < 5900, total
    # the oracle, driven with the recorded actions: same env trajectory AND same decoder outputs
    np.random.seed(0)
    ora = O.OracleVectorEnv(O.OracleConfig(**{**ec, "num_envs": 1, "zero_start_prob": 1.0}))
    dec = {"last_press": np.full((1, 4), -cfg.key_press_delay), "last_keys": np.zeros((1, 4), bool), "yaw": ora.yaw.copy()}
    for t in range(720):
        o = ora.observation()
        assert np.array_equal(o[0], res.obs[t]), t
        y, sm, fm, j = O.decode(ora.cfg, dec, res.action[t][None, :], o[:, 5], ora.t_rem)
        assert y[0] == res.yaw[t] and sm[0] == res.smove[t] and fm[0] == res.fmove[t] and bool(j[0]) == bool(res.jump[t]), t
        _, r, _, _ = ora.vector_step(res.action[t][None, :])
        assert r[0] == res.reward[t]
    cmds = res.move_commands()
    assert set(np.unique(cmds["buttons"])) <= {0, 2} and np.abs(cmds["side"]).max() <= 1060 and cmds["forward"].max() <= 800
    assert np.isfinite(res.wish_angle).all() and res.move_angle.shape == (720,)


def _emulate_fused_forward(net, obs):
    """torch restatement of q1env_policy_forward's arithmetic: float16 weights with float32 accumulation; layer 1 takes the inputs
    and its bias split into two float16 (hi + lo); tanh(z) = 1 - 2 / (2^(c z) + 1) with c = 2 log2(e) folded into W1, b1, W2, b2
    BEFORE their float16 rounding; hidden activations rounded to float16."""
    import torch
    from q1physrl_amd.policy import TANH_PRESCALE as C
    l1, l2, l3 = net[0], net[2], net[4]
    bf = lambda w: w.to(torch.float16).float()

    def split(x):
        hi = bf(x)
        return hi + bf(x - hi)

    def act(a):                                            # a = c z
        return bf(1.0 - 2.0 / (torch.exp2(a) + 1.0))
    h1 = act(split(obs) @ bf(C * l1.weight).T + split(C * l1.bias))
    h2 = act(h1 @ bf(C * l2.weight).T + C * l2.bias)
    return h2 @ bf(l3.weight).T + l3.bias


@pytest.mark.parametrize("n", [32768, 1000, 37, 70001])
def test_fused_mfma_policy_forward(n):
    """The matrix-core forward pass against (a) a torch emulation of its own mixed precision - this pins the MFMA operand /
    accumulator layout, any mistake there is an O(1) error - and (b) the float32 torch modules (bf16 rounding only)."""
    import json
    import os
    import torch
    from q1physrl_amd import policy as P
    torch.manual_seed(n)
    cfg, env = make_env(n, seed=3)
    pol = P.Q1Policy().cuda()
    with torch.no_grad():                                   # asymmetric, well-scaled random weights (default init is tiny on the head)
        for net in (pol.pi, pol.vf):
            for layer in (net[0], net[2], net[4]):
                layer.weight.copy_(torch.randn_like(layer.weight) * (1.5 / layer.in_features ** 0.5))
                layer.bias.copy_(torch.randn_like(layer.bias) * 0.3)
    fused = P.FusedPolicyForward(pol, env)
    obs = (torch.randn((n, 6), device="cuda") * torch.tensor([0.5, 3.0, 0.3, 2.0, 2.0, 1.0], device="cuda")).contiguous()
    l2, v2 = (t.clone() for t in fused(obs, separate_launches=True))   # one q1env_policy_forward per network
    logits, value = fused(obs)                                         # q1env_policy_value_forward: both in one launch
    torch.cuda.synchronize()
    assert torch.equal(l2, logits) and torch.equal(v2, value)
    with torch.no_grad():
        emu_l, emu_v = _emulate_fused_forward(pol.pi, obs), _emulate_fused_forward(pol.vf, obs)[:, 0]
        ref_l, ref_v = pol(obs)
    assert torch.isfinite(logits).all() and logits.shape == (n, 10) and value.shape == (n,)
    # (two float16 roundings: a float32-level difference in a pre-activation can flip one rounding -> up to ~1e-3)
    assert float((logits - emu_l).abs().max()) < 2e-3 and float((value - emu_v).abs().max()) < 2e-3
    assert float((logits - emu_l).abs().mean()) < 5e-5
    # float16 operands: the behaviour policy stays within ~2e-3 of the float32 learner policy (bfloat16 operands: ~1e-1)
    assert float((logits - ref_l).abs().max()) < 6e-3 and float((value - ref_v).abs().max()) < 6e-3
    # the WR policy through the fused forward still plays at its published level
    w = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wr_policy.npz")))
    env.close()
    if n == 1000:
        from q1physrl_amd.env import Config
        from q1physrl_amd.sampler import GpuSampler
        from q1physrl_amd.tensor_env import TensorVectorEnv
        ec = json.loads(str(w["env_config_json"]))
        ec["initial_yaw_range"] = tuple(ec["initial_yaw_range"])
        env2 = TensorVectorEnv(Config(**{**ec, "num_envs": 1024, "zero_start_prob": 1.0}), seed=7)
        wr = P.load_rllib_fcnet_weights(P.Q1Policy(), w).cuda()
        tr = GpuSampler(env2, P.FusedPolicyForward(wr, env2), horizon=720).collect()
        total = float(tr["reward"].double().sum(0).mean())
        assert 5600.0 < total < 5800.0, total
        env2.close()


def test_fused_policy_refresh_keeps_buffer_addresses():
    """A captured sampler graph holds the kernel's weight-buffer addresses: refresh() after an optimiser step must update
    them in place (re-allocating made graph replays read freed memory -> NaN logits)."""
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.sampler import GpuSampler
    cfg, env = make_env(2048, seed=4, zero_start_prob=0.5)
    pol = P.Q1Policy().cuda()
    fused = P.FusedPolicyForward(pol, env)
    ptrs = [t.data_ptr() for w in fused._w.values() for t in w]
    s = GpuSampler(env, fused, horizon=16, use_graph=True)
    s.collect()
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.01 * torch.randn_like(p))
    fused.refresh()
    assert ptrs == [t.data_ptr() for w in fused._w.values() for t in w]
    tr = s.collect()                                           # graph replay with the new weights
    torch.cuda.synchronize()
    assert torch.isfinite(tr["logits"]).all() and torch.isfinite(tr["logp"]).all()
    with torch.no_grad():
        ref, _ = pol(tr["obs"][3])
    assert float((tr["logits"][3] - ref).abs().max()) < 0.05   # replay really used the refreshed weights
    env.close()


def test_graph_captured_sgd_step_matches_eager_learner():
    """PPOLearner(use_graph=True) replays one captured SGD step (loss, backward, Adam) per minibatch: after two updates on
    the same trajectories the parameters agree with the eager learner's (same minibatch permutation, same Adam arithmetic;
    the capture's warm-up steps are rolled back)."""
    import copy
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(3)
    base = P.Q1Policy().cuda()
    cfg, env = make_env(512, time_limit=1.0)
    smp = S.GpuSampler(env, base, horizon=32)
    trajs = []
    for _ in range(2):
        tr = {k: v.clone() for k, v in smp.collect().items()}
        adv, vt = smp.advantages(tr, 0.99, 0.95)
        trajs.append((tr, adv.clone(), vt.clone()))
    env.close()
    res = []
    for use_graph in (False, True):
        pol = copy.deepcopy(base)
        lr = ppo.PPOLearner(pol, cfg.action_range, lr=1e-3, num_sgd_iter=3, minibatch_size=2048, seed=11, use_graph=use_graph)
        assert lr.use_graph == use_graph
        stats = [lr.update(*t) for t in trajs]
        res.append(([p.detach().clone() for p in pol.parameters()], stats))
    (pe, se), (pg, sg) = res
    moved = max((a - b).abs().max().item() for a, b in zip(pe, base.parameters()))
    assert moved > 1e-3                                   # the updates did something
    for a, b in zip(pe, pg):
        assert torch.allclose(a, b, rtol=0, atol=2e-5), (a - b).abs().max().item()
    for a, b in zip(se, sg):
        assert a["sgd_steps"] == b["sgd_steps"] == 3 * (512 * 32 // 2048) and a["kl_coeff"] == b["kl_coeff"]
        for k in ppo.STAT_KEYS:
            assert abs(a[k] - b[k]) <= 1e-3 * max(1.0, abs(a[k])), (k, a[k], b[k])


@pytest.mark.parametrize("n,fused_policy", [(4096, False), (1000, True)])
def test_fused_sampler_tick_equals_three_kernels(n, fused_policy):
    """q1env_sample_step (sample + step + autoreset + episode statistics in one launch, RNG counter = device tick count + tick
    index) against the q1env_policy_sample / q1env_step_autoreset / q1env_episode_stats sequence: identical trajectories, env
    state and statistics over three horizons with episode ends and resets (ragged tail wave included for n = 1000)."""
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.sampler import GpuSampler
    res = []
    for fused_tick in (False, True):
        torch.manual_seed(1)
        cfg, env = make_env(n, seed=9, time_limit=0.5, zero_start_prob=0.3)
        pol = P.Q1Policy().cuda()
        s = GpuSampler(env, P.FusedPolicyForward(pol, env) if fused_policy else pol, horizon=48, fused_tick=fused_tick)
        trajs = [{k: v.clone() for k, v in s.collect().items()} for _ in range(3)]
        res.append((trajs, s.stats, env.get_state(), int(s.tick.item())))
        env.close()
    (ta, sa, ea, ca), (tb, sb, eb, cb) = res
    assert ca == cb == 3 * 48 and sa == sb and sa["episodes"] > n
    for h in range(3):
        for k in ta[h]:
            assert torch.equal(ta[h][k], tb[h][k]), (h, k)
    for k in ea:
        assert np.array_equal(ea[k], eb[k]), k


def test_ppo_loss_grad_kernel_matches_autograd():
    """q1env_ppo_loss_grad (closed-form derivatives, one kernel) against torch autograd of q1physrl_amd.ppo.ppo_loss on the same
    minibatch: statistics and d loss / d(logits, value), including clamped mean / log_std rows, ratios outside the clip range
    and clipped value errors."""
    import torch
    from q1physrl_amd import ppo
    from q1physrl_amd.policy import Q1PhysActionDist
    torch.manual_seed(5)
    bsz = 5000                                               # ragged last block
    cfg, env = make_env(64)
    g = torch.Generator(device="cuda").manual_seed(2)
    rnd = lambda *sh, **kw: torch.randn(*sh, device="cuda", generator=g, **kw)
    old_logits = rnd(bsz, 10) * torch.tensor([1, 1, 1, 1, 1, 1, 1, 1, 1.5, 0.7], device="cuda")
    logits = (old_logits + 0.3 * rnd(bsz, 10)).requires_grad_(True)
    with torch.no_grad():
        logits[:200, 8] = 3.5                                 # mean outside +-3 -> clamp gates the gradient
        logits[200:300, 9] = 2.5                             # log_std above its clamp
        logits[300:400, 9] = -21.0
        dist_old = Q1PhysActionDist(old_logits, cfg.action_range)
        keys, mouse = dist_old.sample(generator=g)
        logp_old = dist_old.logp(keys, mouse) + 0.4 * rnd(bsz)           # ratios spread well beyond 1 +- 0.3
    value_old = 50.0 * rnd(bsz)
    value = (value_old + 60.0 * rnd(bsz) * (torch.rand(bsz, device="cuda", generator=g) < 0.5)).requires_grad_(True)
    vtarg = value_old + 80.0 * rnd(bsz)
    adv = rnd(bsz)
    klc = 0.37

    class Fixed(torch.nn.Module):                            # ppo_loss calls policy(obs): hand it the leaf tensors
        def forward(self, obs):
            return logits, value
    mb = {"obs": None, "old_logits": old_logits, "keys": keys, "mouse": mouse, "logp": logp_old, "adv": adv, "value": value_old,
          "vtarg": vtarg}
    loss, st = ppo.ppo_loss(Fixed(), mb, cfg.action_range, 0.3, 100.0, 1.0, 0.01, klc)
    loss.backward()
    packed = (keys.long() << torch.arange(4, device="cuda")).sum(1).to(torch.uint8)
    dl, dv = torch.empty_like(logits), torch.empty_like(value)
    partials = torch.zeros(((bsz + 255) // 256, 5), device="cuda")
    klc_dev = torch.tensor(klc, device="cuda")
    env._dev.ppo_loss_grad_dev(bsz, logits.data_ptr(), old_logits.data_ptr(), 10, packed.data_ptr(), mouse.data_ptr(), logp_old.data_ptr(),
                               adv.data_ptr(), value.data_ptr(), value_old.data_ptr(), vtarg.data_ptr(), 0.3, 100.0, 1.0, 0.01,
                               klc_dev.data_ptr(), dl.data_ptr(), dv.data_ptr(), partials.data_ptr())
    torch.cuda.synchronize()
    stats = dict(zip(ppo.STAT_KEYS, (partials.double().sum(0) / bsz).tolist()))
    for k in ppo.STAT_KEYS:
        assert abs(stats[k] - float(st[k])) <= 2e-5 * max(1.0, abs(float(st[k]))), (k, stats[k], float(st[k]))
    scale = float(logits.grad.abs().max())
    assert float((dl - logits.grad).abs().max()) <= 2e-5 * scale, (float((dl - logits.grad).abs().max()), scale)
    assert float((dv - value.grad).abs().max()) <= 1e-6 * float(value.grad.abs().max())
    assert float(logits.grad[:200, 8].abs().max()) == 0.0 and float(dl[:200, 8].abs().max()) == 0.0      # gated by the clamp
    assert float(dl[200:400, 9].abs().max()) == 0.0
    env.close()


def test_fused_loss_learner_matches_torch_loss_learner():
    """PPOLearner(fused_loss=True) (eager and graph-captured) against the torch-loss learner on the same trajectories."""
    import copy
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(3)
    base = P.Q1Policy().cuda()
    cfg, env = make_env(512, time_limit=1.0)
    smp = S.GpuSampler(env, base, horizon=32)
    trajs = []
    for _ in range(2):
        tr = {k: v.clone() for k, v in smp.collect().items()}
        adv, vt = smp.advantages(tr, 0.99, 0.95)
        trajs.append((tr, adv.clone(), vt.clone()))
    res = []
    for fused, graph in ((False, False), (True, False), (True, True)):
        pol = copy.deepcopy(base)
        lr = ppo.PPOLearner(pol, cfg.action_range, lr=1e-3, num_sgd_iter=3, minibatch_size=2048, seed=11, use_graph=graph,
                            fused_loss=fused, env=env)
        stats = [lr.update(*t) for t in trajs]
        res.append(([p.detach().clone() for p in pol.parameters()], stats))
    env.close()
    (p0, s0) = res[0]
    for pk, sk in res[1:]:
        for a, b in zip(p0, pk):              # 18 Adam steps of lr 1e-3: float32-level gradient differences move a weight by << lr
            assert float((a - b).abs().max()) < 5e-4 and float((a - b).abs().mean()) < 1e-5, float((a - b).abs().max())
        for a, b in zip(s0, sk):
            assert a["kl_coeff"] == b["kl_coeff"] and a["sgd_steps"] == b["sgd_steps"]
            for k in ppo.STAT_KEYS:
                assert abs(a[k] - b[k]) <= 1e-3 * max(1.0, abs(a[k])), (k, a[k], b[k])


# ------------------------------------------------------------------------------------------------ discrete mouse (yaw_mode 2)
DISCRETE = dict(discrete_yaw_steps=5, time_delta=0.013888888888888888)      # the Config of tests/golden/g4_discrete_yaw5.npz


def test_policy_sample_discrete_mouse_matches_restatement():
    """Config.discrete_yaw_steps = 5 (reference env.py:216-219: the mouse child is Discrete(11), a Categorical under RLlib's
    ModelCatalog, action_dist.py:221-222): q1env_policy_sample's categorical branch against oracle/dist_oracle.py on the same
    Philox draws, the torch Categorical's logp on the kernel's own samples, sample frequencies, and deterministic mode."""
    import torch
    from q1physrl_amd import policy as P
    n, seed, counter, base = 40_000, 11, 5, 77_000_000_000
    cfg, env = make_env(n, seed, base, **DISCRETE)
    width = P.policy_row_width(4, 5)
    rng = np.random.default_rng(3)
    logits = rng.normal(0, 1.5, (n, width)).astype(np.float32)
    logits[:2000, 8:] = np.float32(0.25)                     # exact ties: the arg-max must be the FIRST maximum
    lt = torch.from_numpy(logits).cuda()
    keys = torch.empty(n, dtype=torch.uint8, device="cuda")
    mouse = torch.empty(n, dtype=torch.float32, device="cuda")
    logp = torch.empty(n, dtype=torch.float32, device="cuda")
    env._dev.policy_sample_dev(lt.data_ptr(), width, seed, counter, keys.data_ptr(), mouse.data_ptr(), logp.data_ptr())
    torch.cuda.synchronize()
    genv = np.arange(n, dtype=np.uint64) + np.uint64(base)
    k2, m2, lp2, margin = DO.sample_from_philox(cfg, logits, seed, genv, counter)
    sure = margin > 1e-5
    assert sure.mean() > 0.995
    mk = mouse.cpu().numpy()
    assert np.array_equal(keys.cpu().numpy()[sure], k2[sure]) and np.array_equal(mk[sure], m2[sure])
    assert mk.min() >= 0 and mk.max() <= 10 and np.array_equal(mk, np.round(mk))
    lp = logp.cpu().numpy()
    assert np.max(np.abs(lp[sure] - lp2[sure]) / np.maximum(np.abs(lp2[sure]), 1.0)) < 2e-5
    dist = P.Q1PhysActionDist(lt.double(), float(np.float32(cfg.action_range)), 4, 5)
    kbits = ((keys[:, None].long() >> torch.arange(4, device="cuda")) & 1)
    lp_t = dist.logp(kbits, mouse.double().view(-1, 1)).cpu().numpy()
    assert np.max(np.abs(lp - lp_t) / np.maximum(np.abs(lp_t), 1.0)) < 2e-5
    # frequencies on the tied rows: uniform over 11 steps
    freq = np.bincount(mk[:2000].astype(np.int64), minlength=11) / 2000.0
    assert np.max(np.abs(freq - 1 / 11)) < 0.03
    env._dev.policy_sample_dev(lt.data_ptr(), width, seed, counter, keys.data_ptr(), mouse.data_ptr(), logp.data_ptr(), True)
    torch.cuda.synchronize()
    kd, md, lpd, _ = DO.sample_from_philox(cfg, logits, seed, genv, counter, deterministic=True)
    assert np.array_equal(keys.cpu().numpy(), kd) and np.array_equal(mouse.cpu().numpy(), md)
    assert np.all(mouse.cpu().numpy()[:2000] == 0)            # first maximum of a tie
    assert np.max(np.abs(logp.cpu().numpy() - lpd) / np.maximum(np.abs(lpd), 1.0)) < 2e-5
    env.close()


def test_discrete_mouse_sampler_tick_and_oracle_replay():
    """The whole sampler tick with a discrete mouse: fused q1env_sample_step == the three separate kernels bit for bit, through
    the torch policy AND the fused matrix-core forward (19 output rows); the stored actions replayed through the NumPy env oracle
    (discrete decode branch, env.py:238) reproduce every stored reward and done flag of a 64-env slice until its first reset."""
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.sampler import GpuSampler
    n, T = 1000, 60
    res = []
    for fused_tick, fused_policy in ((False, False), (True, False), (True, True)):
        torch.manual_seed(4)
        cfg, env = make_env(n, seed=21, zero_start_prob=1.0, time_limit=0.6, **DISCRETE)
        pol = P.Q1Policy(discrete_yaw_steps=5).cuda()
        with torch.no_grad():
            pol.pi[-1].weight.mul_(60.0)                       # spread the step distribution (normc 0.01 init is near-uniform)
        s = GpuSampler(env, P.FusedPolicyForward(pol, env) if fused_policy else pol, horizon=T, fused_tick=fused_tick)
        tr = {k: v.clone() for k, v in s.collect().items()}
        res.append((tr, s.stats, env.get_state()))
        env.close()
    (ta, sa, ea), (tb, sb, eb), (tc, sc, ec) = res
    assert sa == sb and sa["episodes"] >= n
    for k in ta:
        assert torch.equal(ta[k], tb[k]), k
    for k in ea:
        assert np.array_equal(ea[k], eb[k]), k
    assert ta["logits"].shape == (T, n, 19) and tc["logits"].shape == (T, n, 19)
    assert float((tc["logits"][0] - ta["logits"][0]).abs().max()) < 2e-2      # f16 matrix-core forward vs float32 modules, tick 0
    mouse = ta["mouse"].cpu().numpy()
    assert mouse.min() >= 0 and mouse.max() <= 10 and len(np.unique(mouse)) > 3
    # oracle replay of envs 0..63 (zero starts: known initial state) over the first episode (43 ticks at time_limit 0.6)
    m = 64
    np.random.seed(0)
    ora = O.OracleVectorEnv(O.OracleConfig(**{**cfg.__dict__, "num_envs": m}))
    keys = ta["keys"].cpu().numpy()
    rew, done = ta["reward"].cpu().numpy(), ta["done"].cpu().numpy()
    first_done = int(np.argmax(done[:, 0]))
    assert done[first_done, :m].all() and first_done > 30
    for t in range(first_done + 1):
        a = np.concatenate([((keys[t, :m, None] >> np.arange(4)) & 1).astype(np.float64), mouse[t, :m, None].astype(np.float64)], axis=1)
        o, r, d, _ = ora.vector_step(a)
        assert np.array_equal(r, rew[t, :m]) and np.array_equal(d, done[t, :m].astype(bool)), t
        if t < first_done:
            assert np.array_equal(o.astype(np.float32), ta["obs"][t + 1, :m].cpu().numpy()), t


def test_ppo_loss_grad_kernel_discrete_mouse_matches_autograd_and_oracle():
    import torch
    from oracle import ppo_oracle as PO
    from q1physrl_amd import ppo
    from q1physrl_amd.policy import Q1PhysActionDist
    torch.manual_seed(6)
    bsz, steps, width = 3001, 5, 19
    cfg, env = make_env(64, **DISCRETE)
    g = torch.Generator(device="cuda").manual_seed(3)
    rnd = lambda *sh, **kw: torch.randn(*sh, device="cuda", generator=g, **kw)
    stride = 24                                                  # row stride wider than the row: the padding columns get zero gradient
    old_full, new_full = 1.2 * rnd(bsz, stride), None
    new_full = (old_full + 0.3 * rnd(bsz, stride)).contiguous()
    old_logits = old_full[:, :width]
    logits = new_full[:, :width].clone().requires_grad_(True)
    with torch.no_grad():
        dist_old = Q1PhysActionDist(old_logits, cfg.action_range, 4, steps)
        keys, mouse = dist_old.sample(generator=g)
        logp_old = dist_old.logp(keys, mouse) + 0.4 * rnd(bsz)
    value_old = 50.0 * rnd(bsz)
    value = (value_old + 60.0 * rnd(bsz) * (torch.rand(bsz, device="cuda", generator=g) < 0.5)).requires_grad_(True)
    vtarg, adv, klc = value_old + 80.0 * rnd(bsz), rnd(bsz), 0.37

    class Fixed(torch.nn.Module):
        def forward(self, obs):
            return logits, value
    mb = {"obs": None, "old_logits": old_logits, "keys": keys, "mouse": mouse, "logp": logp_old, "adv": adv, "value": value_old, "vtarg": vtarg}
    loss, st = ppo.ppo_loss(Fixed(), mb, cfg.action_range, 0.3, 100.0, 1.0, 0.01, klc, discrete_yaw_steps=steps)
    loss.backward()
    packed = (keys.long() << torch.arange(4, device="cuda")).sum(1).to(torch.uint8)
    mouse_f = mouse.reshape(-1).float().contiguous()
    dl = torch.full((bsz, stride), 7.0, device="cuda")
    dv = torch.empty_like(value)
    partials = torch.zeros(((bsz + 255) // 256, 5), device="cuda")
    klc_dev = torch.tensor(klc, device="cuda")
    env._dev.ppo_loss_grad_dev(bsz, new_full.data_ptr(), old_full.contiguous().data_ptr(), stride, packed.data_ptr(), mouse_f.data_ptr(),
                               logp_old.data_ptr(), adv.data_ptr(), value.data_ptr(), value_old.data_ptr(), vtarg.data_ptr(), 0.3, 100.0,
                               1.0, 0.01, klc_dev.data_ptr(), dl.data_ptr(), dv.data_ptr(), partials.data_ptr())
    torch.cuda.synchronize()
    stats = dict(zip(ppo.STAT_KEYS, (partials.double().sum(0) / bsz).tolist()))
    for k in ppo.STAT_KEYS:
        assert abs(stats[k] - float(st[k])) <= 3e-5 * max(1.0, abs(float(st[k]))), (k, stats[k], float(st[k]))
    scale = float(logits.grad.abs().max())
    assert float((dl[:, :width] - logits.grad).abs().max()) <= 3e-5 * scale
    assert float(dl[:, width:].abs().max()) == 0.0
    assert float((dv - value.grad).abs().max()) <= 1e-6 * float(value.grad.abs().max())
    b = {"keys": keys.cpu().numpy(), "mouse": mouse.cpu().numpy(), "logp": logp_old.cpu().numpy().astype(np.float64),
         "adv": adv.cpu().numpy().astype(np.float64), "value": value_old.cpu().numpy().astype(np.float64),
         "vtarg": vtarg.cpu().numpy().astype(np.float64), "old_logits": old_logits.cpu().numpy()}
    dlo, dvo, sto = PO.ppo_loss_grad_discrete(logits.detach().cpu().numpy(), value.detach().cpu().numpy(), b, steps, 0.3, 100.0, 1.0, 0.01, klc)
    assert np.max(np.abs(dl[:, :width].cpu().numpy() - dlo)) <= 3e-5 * scale
    for k in ppo.STAT_KEYS:
        assert abs(stats[k] - sto[k]) <= 3e-5 * max(1.0, abs(sto[k])), k
    env.close()


def test_no_mouse_policy_row_and_rejected_widths():
    """allow_yaw = False: the policy row is the key pairs only (8 columns) and the sampler tick runs; a row stride smaller than
    the Config's row is refused loudly."""
    import torch
    from q1physrl_amd import _lib, policy as P
    from q1physrl_amd.sampler import GpuSampler
    cfg, env = make_env(512, seed=3, allow_yaw=False, time_limit=0.5, zero_start_prob=1.0)    # 37-tick episodes
    pol = P.Q1Policy(allow_yaw=False).cuda()
    s = GpuSampler(env, pol, horizon=40)
    tr = s.collect()
    torch.cuda.synchronize()
    assert tr["logits"].shape == (40, 512, 8) and torch.isfinite(tr["logp"]).all() and s.stats["episodes"] >= 512
    env.close()
    cfg, env = make_env(64, **DISCRETE)
    lt = torch.zeros((64, 19), device="cuda")
    k, m = torch.empty(64, dtype=torch.uint8, device="cuda"), torch.empty(64, device="cuda")
    with pytest.raises(_lib.Q1EnvError, match="row_stride"):
        env._dev.policy_sample_dev(lt.data_ptr(), 18, 1, 0, k.data_ptr(), m.data_ptr())
    env.close()


def test_graph_recapture_keeps_adam_state():
    """A change of the minibatch size re-captures the SGD-step graph in the middle of training.  The capture's warm-up steps are
    rolled back by restoring parameters AND optimiser state (moments, step counts) - not by zeroing the latter: after a second
    update with another minibatch size the graph learner still agrees with the eager one."""
    import copy
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(5)
    base = P.Q1Policy().cuda()
    cfg, env = make_env(512, time_limit=1.0)
    smp = S.GpuSampler(env, base, horizon=32)
    trajs = []
    for _ in range(2):
        tr = {k: v.clone() for k, v in smp.collect().items()}
        adv, vt = smp.advantages(tr, 0.99, 0.95)
        trajs.append((tr, adv.clone(), vt.clone()))
    env.close()
    res = []
    for use_graph in (False, True):
        pol = copy.deepcopy(base)
        lr = ppo.PPOLearner(pol, cfg.action_range, lr=1e-3, num_sgd_iter=3, minibatch_size=2048, seed=11, use_graph=use_graph)
        lr.update(*trajs[0])
        lr.minibatch_size = 4096                          # -> re-capture on the next update (graph learner)
        lr.update(*trajs[1])
        steps = [int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"]) for st in lr.opt.state.values()]
        res.append(([p.detach().clone() for p in pol.parameters()], steps))
    (pe, se), (pg, sg) = res
    assert se == sg and se[0] == 3 * (512 * 32 // 2048) + 3 * (512 * 32 // 4096)       # Adam's step count survived the re-capture
    for a, b in zip(pe, pg):
        assert torch.allclose(a, b, rtol=0, atol=5e-5), (a - b).abs().max().item()


def test_policy_kernels_reproduce_hand_derived_known_answers():
    """The policy-side HIP kernels against the hand-derived known answers of the reference's action distribution
    (tests/golden/dist_known_answers.json <- oracle/gen_dist_known_answers.py: closed forms of action_dist.py:91-96, 153-196 evaluated by
    hand, independent of oracle/dist_oracle.py and of TensorFlow): (a) q1env_policy_sample in deterministic mode returns arg-max keys, the
    squashed CLIPPED mean and that action's log-probability; (b) q1env_ppo_loss_grad: with logp_old = the known log-probability and
    advantage 1 the probability ratio is 1 (policy_loss = -1), and its entropy / KL statistics are the known sums over the five
    children.  float32 kernels: 2e-5 absolute on O(1..20) quantities."""
    import json
    import torch
    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dist_known_answers.json")))
    assert k["low"] == -10.0 and k["high"] == 10.0
    # ---- (a) deterministic sampling
    det = [c for c in k["cases"] if c["kind"] == "tuple_deterministic"][0]
    rows, want_x = [det["row"]], [det["expect_x"]]
    for c in k["cases"]:
        if c["kind"] == "deterministic":
            rows.append([0.0, 1.0] * 4 + c["self_in"]); want_x.append(c["expect"])
        if c["kind"] == "squash" and abs(c["raw"]) <= 3.0:                       # (beyond 3 the mean clip acts first)
            rows.append([0.0, 1.0] * 4 + [c["raw"], -0.3]); want_x.append(c["expect"])
    n = len(rows)
    cfg, env = make_env(n, seed=1, action_range=10.0)
    lt = torch.tensor(rows, dtype=torch.float32, device="cuda")
    keys = torch.empty(n, dtype=torch.uint8, device="cuda")
    mouse = torch.empty(n, dtype=torch.float32, device="cuda")
    logp = torch.empty(n, dtype=torch.float32, device="cuda")
    env._dev.policy_sample_dev(lt.data_ptr(), 10, 5, 0, keys.data_ptr(), mouse.data_ptr(), logp.data_ptr(), True)
    torch.cuda.synchronize()
    assert keys.cpu().tolist() == [15] * n
    assert np.max(np.abs(mouse.cpu().numpy().astype(np.float64) - np.array(want_x))) < 2e-5
    assert abs(float(logp[0]) - det["expect_logp"]) < 2e-5
    env.close()
    # ---- (b) loss kernel statistics on the known minibatch
    b = k["batch"]
    bsz = len(b)
    cfg, env = make_env(bsz, seed=1, action_range=10.0)
    new = torch.tensor([r["row"] for r in b], dtype=torch.float32, device="cuda")
    old = torch.tensor([r["old_row"] for r in b], dtype=torch.float32, device="cuda")
    packed = torch.tensor([sum(bit << j for j, bit in enumerate(r["keys"])) for r in b], dtype=torch.uint8, device="cuda")
    x = torch.tensor([r["x"] for r in b], dtype=torch.float32, device="cuda")
    logp_old = torch.tensor([r["expect_logp"] for r in b], dtype=torch.float32, device="cuda")
    adv = torch.ones(bsz, device="cuda")
    value = torch.zeros(bsz, device="cuda")
    dl, dv = torch.empty_like(new), torch.empty_like(value)
    partials = torch.zeros(((bsz + 255) // 256, 5), device="cuda")
    klc = torch.tensor(0.2, device="cuda")
    env._dev.ppo_loss_grad_dev(bsz, new.data_ptr(), old.data_ptr(), 10, packed.data_ptr(), x.data_ptr(), logp_old.data_ptr(), adv.data_ptr(),
                               value.data_ptr(), value.data_ptr(), value.data_ptr(), 0.3, 100.0, 1.0, 0.01, klc.data_ptr(), dl.data_ptr(),
                               dv.data_ptr(), partials.data_ptr())
    torch.cuda.synchronize()
    from q1physrl_amd import ppo
    stats = dict(zip(ppo.STAT_KEYS, (partials.double().sum(0) / bsz).tolist()))
    want_h = float(np.mean([r["expect_entropy"] for r in b]))
    want_kl = float(np.mean([r["expect_kl_old_new"] for r in b]))
    # the float32 mouse action x carries ~1e-6 of rounding, which ndtri amplifies in the tails (z = 2): 1e-4 on the ratio
    assert abs(stats["policy_loss"] + 1.0) < 1e-4, stats
    assert abs(stats["entropy"] - want_h) < 2e-5 * max(1.0, abs(want_h)), (stats["entropy"], want_h)
    assert abs(stats["kl"] - want_kl) < 2e-5 * max(1.0, abs(want_kl)), (stats["kl"], want_kl)
    assert abs(stats["vf_loss"]) == 0.0
    env.close()
