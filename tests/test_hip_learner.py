"""GPU: the native PPO learner step (q1env_learner_*: q1physrl_amd/csrc/q1learner.hpp) - forward bit-identical to the sampler's fused
forward, gradients against torch autograd of the float32 modules (relative 1e-3), the whole step against the
torch learner, eager == graph-captured, a ragged minibatch, and the discrete-mouse head."""
import copy

import numpy as np
import pytest

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def make_env(n, seed=5, **over):
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    cfg = O.OracleConfig.get_default(num_envs=n, **over)
    return cfg, TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=seed)


def _policy(seed=0, scale=1.0, **kw):
    import torch
    from q1physrl_amd import policy as P
    torch.manual_seed(seed)
    pol = P.Q1Policy(**kw).cuda()
    with torch.no_grad():
        for p in pol.parameters():
            p.mul_(scale)
    return pol


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("mb", [4096, 1000, 128])   # 128 = RLlib's sgd_minibatch_size, the reference's
def test_native_forward_is_the_samplers_forward_and_gradients_match_autograd(mb):
    """(a) NativeStep.forward on gathered rows == FusedPolicyForward on the same rows, bit for bit (same mlp_tile).  (b) backward of a
    random linear functional of (logits, value): every parameter gradient against torch autograd through the float32 modules -
    float16 operands with float32 accumulation: relative (Frobenius) error <= 3e-3 per tensor, 1.5e-3 over all of them."""
    import torch
    from q1physrl_amd import policy as P, ppo
    total = 3 * mb + 17
    cfg, env = make_env(mb)
    pol = _policy(1, 1.5)
    g = torch.Generator(device="cuda").manual_seed(2)
    obs = torch.randn((total, 6), device="cuda", generator=g) * torch.tensor([0.5, 3.0, 0.3, 1.5, 1.5, 1.0], device="cuda")
    idx = torch.randperm(total, device="cuda", generator=g)[:mb].contiguous()
    nat = ppo.NativeStep(pol, env, mb, splits=16)
    logits, value = nat.forward(obs, idx)
    fused = P.FusedPolicyForward(pol, env)
    rows = obs[idx].contiguous()
    l2, v2 = fused(rows)
    torch.cuda.synchronize()
    assert torch.equal(logits, l2) and torch.equal(value, v2)
    # (b)
    # per-sample (un-averaged) gradients, as q1env_learner_step passes them: O(1) for the logits, O(10..1000) for the value
    scale = float(mb)
    dl_s = torch.randn((mb, 10), device="cuda", generator=g)
    dv_s = torch.randn((mb,), device="cuda", generator=g) * 300.0
    dl, dv = dl_s / scale, dv_s / scale
    nat.backward(obs, idx, dl_s, dv_s, scale)
    torch.cuda.synchronize()
    got = [p.grad.detach().clone() for p in pol.parameters()]
    ref = copy.deepcopy(pol)
    for p in ref.parameters():
        p.grad = None
    lg, vv = ref(rows)
    ((lg * dl).sum() + (vv * dv).sum()).backward()
    want = [p.grad for p in ref.parameters()]
    names = [n for n, _ in pol.named_parameters()]
    worst = 0.0
    for n_, a, b in zip(names, got, want):
        r = _rel(a, b)
        worst = max(worst, r)
        assert r < 3e-3, (n_, r)
    allg, allw = torch.cat([a.reshape(-1) for a in got]), torch.cat([b.reshape(-1) for b in want])
    assert _rel(allg, allw) < 1.5e-3, _rel(allg, allw)
    cos = float(torch.dot(allg, allw) / (allg.norm() * allw.norm()))
    assert cos > 0.999995, cos
    env.close()


def test_native_backward_no_gather_and_full_rows():
    """idx = None (rows 0..B-1) gives the same gradients as the identity index vector."""
    import torch
    from q1physrl_amd import ppo
    mb = 2048
    cfg, env = make_env(mb)
    pol = _policy(4)
    g = torch.Generator(device="cuda").manual_seed(5)
    obs = torch.randn((mb, 6), device="cuda", generator=g)
    dl = torch.randn((mb, 10), device="cuda", generator=g)
    dv = torch.randn((mb,), device="cuda", generator=g) * 100.0
    nat = ppo.NativeStep(pol, env, mb, splits=8)
    nat.forward(obs, None)
    nat.backward(obs, None, dl, dv, float(mb))
    torch.cuda.synchronize()
    a = [p.grad.clone() for p in pol.parameters()]
    ident = torch.arange(mb, device="cuda")
    nat.forward(obs, ident)
    nat.backward(obs, ident, dl, dv, float(mb))
    torch.cuda.synchronize()
    for x, y in zip(a, pol.parameters()):
        assert torch.equal(x, y.grad)
    env.close()


def test_native_learner_matches_the_torch_learner():
    """PPOLearner(native=True), eager and graph-captured, against the fused-loss torch learner on the same trajectories: same
    statistics, same KL-coefficient schedule, parameters after 18 Adam steps equal to within what float16 operands allow."""
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(3)
    base = P.Q1Policy().cuda()
    cfg, env = make_env(512, time_limit=1.0)
    smp = S.GpuSampler(env, P.FusedPolicyForward(base, env), horizon=32)
    trajs = []
    for _ in range(2):
        tr = {k: v.clone() for k, v in smp.collect().items()}
        adv, vt = smp.advantages(tr, 0.99, 0.95)
        trajs.append((tr, adv.clone(), vt.clone()))
    res = []
    # (static float16 loss scales - the default: eager == graph-captured bit for bit; with dynamic_loss_scale=True a change of scale re-captures
    # the graph in the second update, and its warm-up / roll-back is equal only to ~1e-5)
    for native, graph, own_adam in ((False, False, False), (True, False, True), (True, True, True), (True, False, False)):
        pol = copy.deepcopy(base)
        lr = ppo.PPOLearner(pol, cfg.action_range, lr=1e-3, num_sgd_iter=3, minibatch_size=2048, seed=11, use_graph=graph,
                            fused_loss=True, env=env, native=native, native_splits=8, fused_adam=True, native_adam=own_adam)
        stats = [lr.update(*t) for t in trajs]
        res.append(([p.detach().clone() for p in pol.parameters()], stats))
    env.close()
    (p0, s0) = res[0]
    for pk, sk in res[1:]:
        for a, b in zip(p0, pk):              # 18 Adam steps of lr 1e-3 (each moves a weight by <= lr): sign flips of tiny gradients aside, equal
            assert float((a - b).abs().mean()) < 2e-4, float((a - b).abs().mean())
        for a, b in zip(s0, sk):
            assert a["sgd_steps"] == b["sgd_steps"]
            for k in ("kl", "entropy", "policy_loss", "vf_loss"):
                assert abs(a[k] - b[k]) <= 2e-2 * max(1.0, abs(a[k])), (k, a[k], b[k])
    # eager native == graph-captured native (same kernels, same order); the library's Adam == torch's on the same gradients
    for a, b in zip(res[1][0], res[2][0]):
        assert float((a - b).abs().max()) < 1e-6, float((a - b).abs().max())
    for a, b in zip(res[1][0], res[3][0]):          # (18 steps of lr 1e-3: agreement to a few percent of ONE step's movement)
        assert float((a - b).abs().max()) < 1e-4 and float((a - b).abs().mean()) < 1e-5, float((a - b).abs().max())


def test_gradient_saturation_is_counted_not_hidden():
    """ADVICE r3 / VERDICT r3 weak 6: the backward pass carries per-sample gradients as float16 and clamps at 65504.  An ordinary update
    reports zero clamped elements and a finite maximum; value targets of 1e7 (per-sample value gradients far beyond 65504) must show up
    in grad_saturated_vf / grad_max_abs_vf instead of passing silently - and must still leave every weight finite.  The optimizer state of
    the native Adam survives a change of minibatch size (a re-built NativeStep) and a state_dict round trip."""
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(5)
    pol = P.Q1Policy().cuda()
    cfg, env = make_env(256, time_limit=1.0)
    smp = S.GpuSampler(env, P.FusedPolicyForward(pol, env), horizon=16)
    tr = {k: v.clone() for k, v in smp.collect().items()}
    adv, vt = smp.advantages(tr, 0.99, 0.95)
    lr = ppo.PPOLearner(pol, cfg.action_range, lr=1e-4, num_sgd_iter=2, minibatch_size=1024, seed=1, fused_loss=True, env=env, native=True,
                        native_splits=8)
    st = lr.update(tr, adv, vt)
    assert st["grad_saturated_pi"] == 0 and st["grad_saturated_vf"] == 0
    assert 0.0 < st["grad_max_abs_pi"] < 65504.0 and 0.0 < st["grad_max_abs_vf"] < 65504.0
    step_before = int(lr._adam_state[:8].view(torch.int64)[0])
    assert step_before == st["sgd_steps"] > 0
    sd = lr.state_dict()
    assert sd["native_adam"] is not None and int(sd["native_adam"][:8].view(torch.int64)[0]) == step_before
    # another minibatch size: NativeStep is re-built, Adam's moments and step count must carry over
    lr.minibatch_size = 512
    moments = lr._adam_state[256:256 + 4096].clone()
    st2 = lr.update(tr, adv, vt * 1e7)                                # absurd value targets: the value network's gradients saturate
    assert int(lr._adam_state[:8].view(torch.int64)[0]) == step_before + st2["sgd_steps"]
    assert not torch.equal(moments, lr._adam_state[256:256 + 4096])   # ... and kept evolving from where they were
    assert st2["grad_saturated_vf"] > 0 and st2["grad_max_abs_vf"] > 65504.0 and st2["grad_saturated_pi"] == 0
    assert all(bool(torch.isfinite(p).all()) for p in pol.parameters())
    lr2 = ppo.PPOLearner(pol, cfg.action_range, lr=1e-4, num_sgd_iter=1, minibatch_size=512, seed=1, fused_loss=True, env=env, native=True,
                         native_splits=8)
    lr2.load_state_dict(lr.state_dict())
    assert torch.equal(lr2._adam_state.cpu(), lr._adam_state.cpu()) and lr2.kl_coeff == lr.kl_coeff
    env.close()


def test_native_learner_first_step_sees_ratio_one():
    """The sampler's logits come from the same float16 forward the learner runs: on the first minibatch of an update the new policy IS
    the behaviour policy, so the probability ratio is 1 and the KL 0 to float32 rounding - the property that made float16 (not bfloat16)
    operands necessary for the sampler carries over to the learner."""
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(7)
    pol = P.Q1Policy().cuda()
    cfg, env = make_env(1024, time_limit=1.0)
    smp = S.GpuSampler(env, P.FusedPolicyForward(pol, env), horizon=16)
    tr = smp.collect()
    adv, vt = smp.advantages(tr, 0.99, 0.95)
    lr = ppo.PPOLearner(pol, cfg.action_range, lr=0.0, num_sgd_iter=1, minibatch_size=4096, seed=1, fused_loss=True, env=env, native=True,
                        native_splits=8)
    st = lr.update(tr, adv, vt)
    # (exp(log_std) exp(-log_std) is 1 only to float32 rounding; advantages are standardised, so mean(-adv * ratio) = 0 at ratio 1)
    assert abs(st["kl"]) < 1e-6 and abs(st["policy_loss"]) < 1e-5, st
    env.close()


def test_native_learner_discrete_mouse_head():
    """19 policy outputs (4 keys + 11-way categorical): the backward kernel's second K-step over the output index and the wider dW3."""
    import torch
    from q1physrl_amd import policy as P, ppo
    mb = 2048
    cfg, env = make_env(mb, discrete_yaw_steps=5)
    pol = _policy(9, 1.2, discrete_yaw_steps=5)
    width = P.policy_row_width(4, 5)
    assert width == 19
    g = torch.Generator(device="cuda").manual_seed(6)
    obs = torch.randn((mb, 6), device="cuda", generator=g)
    dl_s = torch.randn((mb, width), device="cuda", generator=g)
    dv_s = torch.randn((mb,), device="cuda", generator=g) * 50.0
    dl, dv = dl_s / mb, dv_s / mb
    nat = ppo.NativeStep(pol, env, mb, splits=4)
    logits, value = nat.forward(obs, None)
    nat.backward(obs, None, dl_s, dv_s, float(mb))
    torch.cuda.synchronize()
    got = [p.grad.detach().clone() for p in pol.parameters()]
    ref = copy.deepcopy(pol)
    for p in ref.parameters():
        p.grad = None
    lg, vv = ref(obs)
    assert float((lg.detach() - logits).abs().max()) < 5e-3
    ((lg * dl).sum() + (vv * dv).sum()).backward()
    for (n_, _), a, b in zip(pol.named_parameters(), got, [p.grad for p in ref.parameters()]):
        assert _rel(a, b) < 3e-3, (n_, _rel(a, b))
    env.close()


def test_native_step_never_produces_non_finite_gradients():
    """float16 operands saturate instead of overflowing: a training run once turned value errors of a few thousand into inf in dZ1,
    NaN in the transposition (inf x 0) and NaN weights.  Value targets of 1e3 (the real range) keep the gradients accurate; absurd
    ones (1e7) must still leave every gradient finite."""
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(2)
    pol = _policy(3, 3.0)                                   # large weights: large data gradients
    cfg, env = make_env(1024, time_limit=1.0)
    smp = S.GpuSampler(env, P.FusedPolicyForward(pol, env), horizon=8)
    tr = smp.collect()
    adv, vt = smp.advantages(tr, 0.99, 0.95)
    t, n = tr["reward"].shape
    full = {"obs": tr["obs"][:t].reshape(t * n, 6).contiguous(), "old_logits": tr["logits"].reshape(t * n, -1).contiguous(),
            "keys_packed": tr["keys"].reshape(-1), "mouse": tr["mouse"].reshape(-1), "logp": tr["logp"].reshape(-1),
            "adv": ((adv - adv.mean()) / adv.std()).reshape(-1).contiguous(), "value": tr["value"][:t].reshape(-1).contiguous(), "vtarg": None}
    klc = torch.tensor(0.2, device="cuda")
    nat = ppo.NativeStep(pol, env, t * n, splits=8)
    ref = None
    for big in (1.0e3, 1.0e7):
        full["vtarg"] = (vt.reshape(-1) + big).contiguous()
        nat.step(full, None, 0.3, 1.0e9, 1.0, 0.01, klc)
        torch.cuda.synchronize()
        for name, p in pol.named_parameters():
            assert bool(torch.isfinite(p.grad).all()), (big, name)
        if big == 1.0e3:        # still accurate: against autograd of the float32 modules with the same loss
            ref = copy.deepcopy(pol)
            for p in ref.parameters():
                p.grad = None
            b = {"obs": full["obs"], "old_logits": full["old_logits"], "keys": ((full["keys_packed"].reshape(-1, 1).long() >> torch.arange(4, device="cuda")) & 1),
                 "mouse": full["mouse"].reshape(-1, 1), "logp": full["logp"], "adv": full["adv"], "value": full["value"], "vtarg": full["vtarg"]}
            loss, _ = ppo.ppo_loss(ref, b, cfg.action_range, 0.3, 1.0e9, 1.0, 0.01, klc)
            loss.backward()
            for (name, p), q in zip(pol.named_parameters(), ref.parameters()):
                if name.startswith("vf."):
                    assert _rel(p.grad, q.grad) < 5e-3, (name, _rel(p.grad, q.grad))
    env.close()


def test_minibatch_cursor_selects_the_same_rows_as_an_index_copy():
    """q1env_learner_batch.idx_cursor_dev: with a whole permutation in idx_dev and a device-resident cursor, step k's gradients equal
    those of a step on a copy of perm[k mb : (k + 1) mb] bit for bit, and q1env_learner_adam advances the cursor by the minibatch."""
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(4)
    pol = _policy(5, 2.0)
    cfg, env = make_env(512, time_limit=1.0)
    smp = S.GpuSampler(env, P.FusedPolicyForward(pol, env), horizon=8)
    tr = smp.collect()
    adv, vt = smp.advantages(tr, 0.99, 0.95)
    t, n = tr["reward"].shape
    total, mb = t * n, 1024
    full = {"obs": tr["obs"][:t].reshape(total, 6).contiguous(), "old_logits": tr["logits"].reshape(total, -1).contiguous(),
            "keys_packed": tr["keys"].reshape(-1), "mouse": tr["mouse"].reshape(-1), "logp": tr["logp"].reshape(-1),
            "adv": ((adv - adv.mean()) / adv.std()).reshape(-1).contiguous(), "value": tr["value"][:t].reshape(-1).contiguous(),
            "vtarg": vt.reshape(-1).contiguous()}
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    nat = ppo.NativeStep(pol, env, mb, splits=8)
    for k in (0, 2, total // mb - 1):
        nat.step(full, perm[k * mb:(k + 1) * mb].contiguous(), 0.3, 10.0, 1.0, 0.01, klc)
        torch.cuda.synchronize()
        want = [p.grad.detach().clone() for p in pol.parameters()]
        nat.cursor.fill_(k * mb)
        nat.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, use_cursor=True)
        torch.cuda.synchronize()
        for a, b in zip(want, [p.grad for p in pol.parameters()]):
            assert torch.equal(a, b), k
    nat.cursor.zero_()
    w0 = [p.detach().clone() for p in pol.parameters()]
    nat.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True)
    nat.adam(1e-4)
    torch.cuda.synchronize()
    assert int(nat.cursor.item()) == mb and any(not torch.equal(a, p) for a, p in zip(w0, pol.parameters()))
    env.close()


@pytest.mark.parametrize("mb,n_envs,over", [(1024, 512, {}), (1000, 500, {}), (128, 64, {}), (1024, 512, dict(allow_jump=False))])
def test_sgd_step_equals_step_plus_adam(mb, n_envs, over):
    """q1env_learner_sgd_step (round 4: the backward kernel computes the PPO loss gradient of its own samples, the optimizer's
    bookkeeping rides in the backward and Adam kernels - four launches) against q1env_learner_step(skip_reduce) + q1env_learner_adam
    (six launches): masters, gradients, moments, step count and cursor bit for bit over several steps; the running statistics to float32
    rounding (summed per workgroup instead of per block of 256 samples).  Ragged minibatches and a three-key Config (which takes the
    two-call path inside) included."""
    import copy
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(4)
    cfg, env = make_env(n_envs, time_limit=1.0, **over)
    pol_a = _policy(5, 2.0, **({"num_keys": 3} if over else {}))
    pol_b = copy.deepcopy(pol_a)
    smp = S.GpuSampler(env, P.FusedPolicyForward(pol_a, env), horizon=8)
    tr = smp.collect()
    adv, vt = smp.advantages(tr, 0.99, 0.95)
    t, n = tr["reward"].shape
    total = t * n
    full = {"obs": tr["obs"][:t].reshape(total, 6).contiguous(), "old_logits": tr["logits"].reshape(total, -1).contiguous(),
            "keys_packed": tr["keys"].reshape(-1), "mouse": tr["mouse"].reshape(-1), "logp": tr["logp"].reshape(-1),
            "adv": ((adv - adv.mean()) / adv.std()).reshape(-1).contiguous(), "value": tr["value"][:t].reshape(-1).contiguous(),
            "vtarg": vt.reshape(-1).contiguous()}
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    a, b = ppo.NativeStep(pol_a, env, mb, splits=8), ppo.NativeStep(pol_b, env, mb, splits=8)
    hp = (3e-4, (0.9, 0.999), 1e-8)
    for k in range(total // mb):
        a.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True)
        a.adam(*hp)
        b.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
        torch.cuda.synchronize()
        for (name, p), q in zip(pol_a.named_parameters(), pol_b.parameters()):
            assert torch.equal(p, q) and torch.equal(p.grad, q.grad), (k, name)
        for lo, hi in ((0, 16), (72, 80), (256, None)):                         # count + bias corrections, cursor, moments
            assert torch.equal(a.adam_state[lo:hi], b.adam_state[lo:hi]), (k, lo)
        assert torch.equal(a.ws[:a.ws.numel() // 2], b.ws[:b.ws.numel() // 2])                                             # the weight images lead the workspace
        sa, sb = a.stats_acc.cpu().numpy(), b.stats_acc.cpu().numpy()
        assert np.allclose(sa, sb, rtol=2e-5, atol=1e-6), (k, sa, sb)
        assert torch.equal(a.saturation, b.saturation)
    assert int(b.cursor.item()) == (total // mb) * mb and int(b.adam_state[:8].view(torch.int64)[0]) == total // mb
    env.close()


def test_native_training_learns_strafe_jumping_in_seconds():
    """End-to-end regression of the whole GPU-resident stack (resident sampler + native learner, round 2's hyper-parameters): 200
    iterations = 0.42 G env-steps in ~17 s must take the zero-start reward from ~1 700 (plain running) past 4 200 - strafe-jumping
    discovered - with finite statistics all the way.  (The full 1 300-iteration runs, 5 643-5 686 in 106-108 s for five seeds, are in
    profiles/r3_train_ppo_native_s*.json.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "train_ppo.py"), "--iters", "200", "--envs", "16384", "--horizon", "128", "--lr", "3e-5",
                        "--epochs", "8", "--minibatch", "32768", "--entropy", "0.01", "--kl-target", "0.0036", "--fused-policy", "--resident",
                        "--fused-loss", "--native", "--seed", "0"], capture_output=True, text=True, timeout=600, cwd=root,
                       env=dict(os.environ, Q1_TUNABLEOP="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    its = [x for x in rows if "iter" in x]
    assert its and all(np.isfinite([x["kl"], x["entropy"], x["vf_loss"]]).all() for x in its)
    zs = [x["zero_start_total_reward_mean"] for x in its if np.isfinite(x["zero_start_total_reward_mean"])]     # (no zero-start episode ends in iteration 0)
    assert zs[0] < 2500.0 < 4200.0 < zs[-1], (zs[0], zs[-1])
    assert its[-1]["iter_s"] < 0.2                                  # <= 0.2 s per iteration (measured 0.08)


# ------------------------------------------------------------------------------------------------ persistent learner (round 5)
def _train_batch(n_envs, horizon, pol, seed=4):
    """A real train batch of n_envs x horizon samples from the sampler (the arrays q1env_learner_batch points at)."""
    import torch
    from q1physrl_amd import policy as P, sampler as S
    torch.manual_seed(seed)
    cfg, env = make_env(n_envs, time_limit=1.0)
    smp = S.GpuSampler(env, P.FusedPolicyForward(pol, env), horizon=horizon)
    tr = smp.collect()
    adv, vt = smp.advantages(tr, 0.99, 0.95)
    t, n = tr["reward"].shape
    total = t * n
    full = {"obs": tr["obs"][:t].reshape(total, 6).contiguous(), "old_logits": tr["logits"].reshape(total, -1).contiguous(),
            "keys_packed": tr["keys"].reshape(-1), "mouse": tr["mouse"].reshape(-1), "logp": tr["logp"].reshape(-1),
            "adv": ((adv - adv.mean()) / adv.std()).reshape(-1).contiguous(), "value": tr["value"][:t].reshape(-1).contiguous(),
            "vtarg": vt.reshape(-1).contiguous()}
    return env, full, total


def test_persistent_learner_one_step_against_the_four_launch_step_and_autograd():
    """q1env_learner_sgd_epochs with steps = 1 against ONE q1env_learner_sgd_step on the same 128 rows from the same state: every
    parameter gradient within float16-operand rounding of the four-launch path's (relative Frobenius error <= 3e-3 per tensor - the bound
    the four-launch path itself is held to against autograd) and, directly, of torch autograd through the float32 modules and
    ppo.ppo_loss; masters, moments and step count after the Adam update accordingly."""
    import copy
    import torch
    from q1physrl_amd import ppo
    pol_a = _policy(5, 2.0)
    pol_b, pol_c = copy.deepcopy(pol_a), copy.deepcopy(pol_a)
    env, full, total = _train_batch(64, 8, pol_a)
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    a, b = ppo.NativeStep(pol_a, env, 128, splits=8), ppo.NativeStep(pol_b, env, 128, splits=8)
    assert b.persistent_ok()
    hp = (3e-4, (0.9, 0.999), 1e-8)
    w0 = [p.detach().clone() for p in pol_a.parameters()]
    a.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
    n = b.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp, steps=1)
    torch.cuda.synchronize()
    assert n == 1 and b.persistent_status()[0] == 0
    # autograd reference of the same minibatch
    idx = perm[:128]
    mbatch = {"obs": full["obs"][idx], "old_logits": full["old_logits"][idx], "mouse": full["mouse"][idx].reshape(-1, 1), "logp": full["logp"][idx],
              "adv": full["adv"][idx], "value": full["value"][idx], "vtarg": full["vtarg"][idx],
              "keys": ((full["keys_packed"][idx].reshape(-1, 1).long() >> torch.arange(4, device="cuda")) & 1)}
    for p in pol_c.parameters():
        p.grad = None
    loss, _st = ppo.ppo_loss(pol_c, mbatch, float(env.config.action_range), 0.3, 10.0, 1.0, 0.01, klc)
    loss.backward()
    for (name, pa), pb, pc in zip(pol_a.named_parameters(), pol_b.parameters(), pol_c.parameters()):
        assert _rel(pb.grad, pa.grad) < 3e-3, (name, "vs four-launch", _rel(pb.grad, pa.grad))
        assert _rel(pb.grad, pc.grad) < 4e-3, (name, "vs autograd", _rel(pb.grad, pc.grad))
    for (name, pa), pb, w in zip(pol_a.named_parameters(), pol_b.parameters(), w0):
        da, db = pa.detach() - w, pb.detach() - w
        assert float(da.abs().max()) > 0 and _rel(db, da) < 2e-2, (name, _rel(db, da))      # first Adam step: +- lr per element, sign(g) - a flipped sign of a ~0 gradient moves 2 lr
    assert torch.equal(a.adam_state[:8], b.adam_state[:8])                                     # step count
    ma, mb_ = a.adam_state[256:].view(torch.float32), b.adam_state[256:].view(torch.float32)
    assert _rel(mb_, ma) < 3e-3
    sa, sb = a.stats_acc.cpu().numpy(), b.stats_acc.cpu().numpy()
    for k in (0, 1, 2, 4):
        assert abs(sa[k] - sb[k]) <= 2e-3 * max(1.0, abs(sa[k])), (k, sa, sb)
    env.close()


def test_persistent_learner_epoch_of_391_steps_against_391_four_launch_steps():
    """VERDICT r4 item 3's shape: one epoch of the reference's train batch - 391 minibatches of 128 out of 50 048 samples - as ONE
    dispatch against 391 calls of q1env_learner_sgd_step (same permutation, same initial state, lr 5e-6 as in data/params.yml).  Not bit
    for bit (the summation orders differ): the accumulated parameter CHANGE agrees to a few per cent per tensor, the running statistics to
    1e-3, step counts exactly; then a second launch continues from the first one's state (moments, step count) like two more calls would."""
    import copy
    import torch
    from q1physrl_amd import ppo
    pol_a = _policy(7, 1.0)
    pol_b = copy.deepcopy(pol_a)
    env, full, total = _train_batch(128, 391, pol_a)
    assert total == 50048
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    a, b = ppo.NativeStep(pol_a, env, 128, splits=8), ppo.NativeStep(pol_b, env, 128, splits=8)
    hp = (5e-6, (0.9, 0.999), 1e-8)
    w0 = [p.detach().clone() for p in pol_a.parameters()]
    for _ in range(391):
        a.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
    n = b.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp)
    torch.cuda.synchronize()
    assert n == 391 and b.persistent_status()[0] == 0
    assert int(a.adam_state[:8].view(torch.int64)[0]) == 391 == int(b.adam_state[:8].view(torch.int64)[0])
    for (name, pa), pb, w in zip(pol_a.named_parameters(), pol_b.parameters(), w0):
        da, db = pa.detach() - w, pb.detach() - w
        assert torch.isfinite(pb).all() and _rel(db, da) < 5e-2, (name, _rel(db, da))
    sa, sb = a.stats_acc.cpu().numpy(), b.stats_acc.cpu().numpy()
    for k in (0, 1, 2, 4):
        assert abs(sa[k] - sb[k]) <= 1e-3 * max(391.0, abs(sa[k])), (k, sa, sb)
    # the images of the four-launch path were refreshed: both paths continue from their own state and stay together
    a.cursor.zero_()
    for _ in range(5):
        a.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
    b.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp, steps=5)
    torch.cuda.synchronize()
    assert int(b.adam_state[:8].view(torch.int64)[0]) == 396
    for (name, pa), pb, w in zip(pol_a.named_parameters(), pol_b.parameters(), w0):
        assert _rel(pb.detach() - w, pa.detach() - w) < 5e-2, name
    env.close()


def test_persistent_learner_consecutive_launches_stay_synchronised():
    """Regression (round 5): every launch must re-zero BOTH groups' barrier counters - with one left at its previous final value the
    second launch's waits all pass at once, its workgroups run unsynchronised and training diverges within a few updates (iteration 0 of
    a run is unaffected, which is why a single-launch test cannot see it).  Six launches of 60 steps each on the same schedule against
    360 four-launch steps: the parameter change agrees to a few per cent after EVERY launch."""
    import copy
    import torch
    from q1physrl_amd import ppo
    pol_a = _policy(11, 1.0)
    pol_b = copy.deepcopy(pol_a)
    env, full, total = _train_batch(128, 61, pol_a)            # 7 808 rows: NOT the size the workspace was first made for in other tests
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    a, b = ppo.NativeStep(pol_a, env, 128, splits=8), ppo.NativeStep(pol_b, env, 128, splits=8)
    hp = (5e-5, (0.9, 0.999), 1e-8)
    w0 = [p.detach().clone() for p in pol_a.parameters()]
    p1 = perm.reshape(1, -1).contiguous()
    for launch in range(6):
        a.cursor.zero_()
        for _ in range(60):
            a.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
        b.epochs(full, p1, 0.3, 10.0, 1.0, 0.01, klc, hp, steps=60)
        torch.cuda.synchronize()
        assert b.persistent_status()[0] == 0
        assert int(b.adam_state[:8].view(torch.int64)[0]) == 60 * (launch + 1)
        for (name, pa), pb, w in zip(pol_a.named_parameters(), pol_b.parameters(), w0):
            r = _rel(pb.detach() - w, pa.detach() - w)
            assert torch.isfinite(pb).all() and r < 5e-2, (launch, name, r)
    env.close()


def test_ppo_learner_update_persistent_equals_per_step_loop():
    """PPOLearner.update with the persistent learner (one dispatch per update) against the same learner driving q1env_learner_sgd_step
    per minibatch: same permutations (same generator), same statistics to 1e-3, same adaptive-KL decision."""
    import copy
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    pol_a = _policy(9, 1.0)
    pol_b = copy.deepcopy(pol_a)
    cfg, env = make_env(128, time_limit=1.0)
    smp = S.GpuSampler(env, P.FusedPolicyForward(pol_a, env), horizon=16)
    tr = smp.collect()
    adv, vt = smp.advantages(tr, 0.99, 0.95)
    outs = []
    for pol, persistent in ((pol_a, False), (pol_b, True)):
        lr_ = ppo.PPOLearner(pol, cfg.action_range, lr=5e-6, num_sgd_iter=3, minibatch_size=128, env=env, native=True, persistent=persistent, seed=3)
        outs.append([lr_.update(tr, adv, vt) for _ in range(4)])          # four updates: the later ones start from the earlier ones' state
    torch.cuda.synchronize()
    for oa, ob in zip(*outs):
        assert oa["sgd_steps"] == ob["sgd_steps"] == 3 * 16 and oa["kl_coeff"] == ob["kl_coeff"]
        for k in ("entropy", "kl", "policy_loss", "total_loss", "vf_loss"):
            assert abs(oa[k] - ob[k]) <= 2e-3 * max(1.0, abs(oa[k])), (k, oa[k], ob[k])
    env.close()


def test_loss_scale_setting_changes_rounding_only_and_insists_on_powers_of_two():
    """q1env_learner_set_loss_scale (ABI v5): the same step under pi_upscale 256 / 16 / 2048 and value_downscale 1 / 8 gives the same
    gradients to float16 rounding (the scale is divided out in float32, exactly); anything but a power of two is refused; the persistent
    learner honours the setting too."""
    import copy
    import torch
    from q1physrl_amd import ppo, _lib
    pol0 = _policy(5, 2.0)
    env, full, total = _train_batch(64, 8, pol0)
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    hp = (3e-4, (0.9, 0.999), 1e-8)
    grads = []
    for up, down, persistent in ((0.0, 0.0, False), (16.0, 8.0, False), (2048.0, 1.0, False), (16.0, 8.0, True)):
        pol = copy.deepcopy(pol0)
        env._dev.learner_set_loss_scale(up, down)
        nat = ppo.NativeStep(pol, env, 128, splits=8)
        if persistent:
            nat.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp, steps=1)
        else:
            nat.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
        torch.cuda.synchronize()
        grads.append([p.grad.detach().clone() for p in pol.parameters()])
    for other in grads[1:]:
        for a, b in zip(grads[0], other):
            assert _rel(b, a) < 3e-3
    with pytest.raises(_lib.Q1EnvError):
        env._dev.learner_set_loss_scale(100.0, 0.0)
    env._dev.learner_set_loss_scale(0.0, 0.0)
    env.close()
