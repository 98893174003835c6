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


@pytest.mark.parametrize("mb,n_envs,horizon", [(4096, 2048, 8), (1000, 500, 8), (128, 64, 8), (2048 + 32 * 5 + 7, 1024, 8), (32768, 8192, 8)])
def test_fused_forward_backward_step_against_the_four_launch_step(mb, n_envs, horizon):
    """q1env_learner_sgd_step's kernel sequences (round 6, q1env_learner_set_step_mode; csrc/q1learner_fused.hpp): forward + loss gradient + data
    gradients as ONE kernel.  "fused" must reproduce the four-launch step BIT FOR BIT - masters, gradients, moments, weight images,
    saturation counters - over several steps (the same arithmetic, only the activations' trip through memory is gone); "fused_dw1" (dZ1 and
    tanh(H2) replaced by their per-tile products with [x | 1] and dY) the same bits for W2, b2, b3 of both networks; W1, b1, W3 differ by float32
    summation order (per-tile products added up in tile order instead of one accumulation chain).
    Sizes: whole workgroups, a ragged last tile, fewer tiles than one workgroup has waves, a tile count that is no multiple of eight, and the
    large-minibatch configuration's 32 768."""
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    torch.manual_seed(4)
    cfg, env = make_env(n_envs, time_limit=1.0)
    pols = [_policy(5, 2.0)]
    pols += [copy.deepcopy(pols[0]), copy.deepcopy(pols[0]), copy.deepcopy(pols[0])]
    smp = S.GpuSampler(env, P.FusedPolicyForward(pols[0], env), horizon=horizon)
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
    modes = ["four_launch", "fused", "fused_dw1", "fused_dw1_q"]
    nats = [ppo.NativeStep(p, env, mb, splits=8) for p in pols]
    hp = (3e-4, (0.9, 0.999), 1e-8)
    n_steps = min(total // mb, 4)
    assert n_steps >= 1
    for k in range(n_steps):
        for mode, nat in zip(modes, nats):
            env._dev.learner_set_step_mode(mode)
            nat.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
        torch.cuda.synchronize()
        a, b, c, d = nats
        # "fused_dw1" runs the shared-operand weight-gradient kernel; the column-quarter form (measurement only) runs the same accumulation chains: equal
        for (name, p), q in zip(pols[2].named_parameters(), pols[3].parameters()):
            assert torch.equal(p, q) and torch.equal(p.grad, q.grad), (k, name)
        assert torch.equal(c.adam_state[256:], d.adam_state[256:]) and torch.equal(c.stats_acc, d.stats_acc), k
        for (name, p), q in zip(pols[0].named_parameters(), pols[1].parameters()):
            assert torch.equal(p, q) and torch.equal(p.grad, q.grad), (k, name)
        assert torch.equal(a.adam_state[:16], b.adam_state[:16]) and torch.equal(a.adam_state[72:80], b.adam_state[72:80])
        assert torch.equal(a.adam_state[256:], b.adam_state[256:]), k
        assert torch.equal(a.saturation, b.saturation) and torch.equal(a.saturation, c.saturation)
        sa, sb, sc = (x.stats_acc.cpu().numpy() for x in nats[:3])
        assert np.allclose(sa, sb, rtol=2e-5, atol=1e-6) and np.allclose(sb, sc, rtol=1e-4, atol=1e-6), (k, sa, sb, sc)
        if k == 0:
            assert np.array_equal(sb, sc)              # the same forward, the same loss, the same workgroups: the same sums
        if k == 0:
            # the first step starts from identical weights: every gradient but dW1 / db1 / dW3 is the same bits, those agree to summation order
            for (name, p), q in zip(pols[0].named_parameters(), pols[2].parameters()):
                by_products = name.split(".")[1] == "0" or name.endswith("4.weight")
                if by_products:
                    assert _rel(q.grad, p.grad) < 2e-6, (name, _rel(q.grad, p.grad))
                    assert not torch.isnan(q.grad).any()
                else:
                    assert torch.equal(p.grad, q.grad), name
        else:
            for (name, p), q in zip(pols[0].named_parameters(), pols[2].parameters()):
                # (Adam's first steps move every element by ~lr whatever its gradient's size: an element of dW1 whose last bits differ around
                # zero may move the other way - bounded by a few lr per element)
                assert _rel(q, p) < 1e-3 and _rel(q.grad, p.grad) < 1e-2, (k, name, _rel(q, p), _rel(q.grad, p.grad))
    env._dev.learner_set_step_mode("auto")
    assert int(nats[1].cursor.item()) == n_steps * mb and int(nats[2].adam_state[:8].view(torch.int64)[0]) == n_steps
    with pytest.raises(Exception):
        env._dev.learner_set_step_mode(7)
    env.close()


@pytest.mark.parametrize("mb,n_envs", [(4096, 1024), (32768, 8192)])
def test_every_step_mode_is_as_close_to_float64_autograd(mb, n_envs):
    """The gradients q1env_learner_sgd_step leaves, per kernel sequence, against torch autograd through the float64 modules and ppo.ppo_loss on the
    same minibatch: every tensor within the float16-operand bound the step has always been held to (relative Frobenius error <= 3e-3; VERDICT r5
    item 7), and the fused kernel with per-tile dW1 / dW3 products no further from float64 than the four-launch step is (its three re-ordered tensors
    W1, b1, W3 within 1.25 x of the four-launch step's own error) - the products change the summation order, not the accuracy."""
    import torch
    from q1physrl_amd import ppo
    pols = [_policy(5, 2.0)]
    pols += [copy.deepcopy(pols[0]), copy.deepcopy(pols[0])]
    env, full, total = _train_batch(n_envs, 8, pols[0])
    assert total >= mb
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    ref = copy.deepcopy(pols[0]).double()
    idx = perm[:mb]
    mbatch = {"obs": full["obs"][idx].double(), "old_logits": full["old_logits"][idx].double(), "mouse": full["mouse"][idx].reshape(-1, 1).double(),
              "logp": full["logp"][idx].double(), "adv": full["adv"][idx].double(), "value": full["value"][idx].double(), "vtarg": full["vtarg"][idx].double(),
              "keys": ((full["keys_packed"][idx].reshape(-1, 1).long() >> torch.arange(4, device="cuda")) & 1)}
    loss, _st = ppo.ppo_loss(ref, mbatch, float(env.config.action_range), 0.3, 10.0, 1.0, 0.01, klc.double())
    loss.backward()
    err = {}
    for mode, pol in zip(["four_launch", "fused", "fused_dw1"], pols):
        env._dev.learner_set_step_mode(mode)
        nat = ppo.NativeStep(pol, env, mb, splits=32)
        nat.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=(0.0, (0.9, 0.999), 1e-8))      # lr 0: gradients only
        torch.cuda.synchronize()
        err[mode] = {name: _rel(p.grad.double(), q.grad) for (name, p), q in zip(pol.named_parameters(), ref.parameters())}
        for name, e in err[mode].items():
            assert e < 3e-3, (mode, name, e)
    env._dev.learner_set_step_mode("auto")
    assert err["fused"] == err["four_launch"]                                       # bit-identical gradients
    for name in err["four_launch"]:
        assert err["fused_dw1"][name] <= 1.25 * err["four_launch"][name] + 1e-7, (name, err["fused_dw1"][name], err["four_launch"][name])
    env.close()


def test_fused_step_with_products_is_deterministic():
    """Two runs of 48 steps of the default large-minibatch sequence (fused kernel + per-tile products + shared-operand weight gradients) from the same state on
    the same rows end on the same bits: no atomics on the data path, fixed summation orders - and no race between the waves that share LDS stages (a race
    would show up as a run-to-run difference long before it showed up in a tolerance)."""
    import torch
    from q1physrl_amd import ppo
    pols = [_policy(5, 2.0)]
    pols.append(copy.deepcopy(pols[0]))
    env, full, total = _train_batch(4096, 8, pols[0])
    mb = 8192 + 32 * 3 + 5
    klc = torch.tensor(0.2, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(7)
    perms = [torch.randperm(total, device="cuda", generator=g) for _ in range(16)]
    hp = (3e-4, (0.9, 0.999), 1e-8)
    env._dev.learner_set_step_mode("auto")
    for pol in pols:
        nat = ppo.NativeStep(pol, env, mb, splits=32)
        for perm in perms:
            nat.cursor.zero_()
            for _ in range(total // mb):
                nat.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
        torch.cuda.synchronize()
        pol._moments = nat.adam_state.clone()
    for (name, p), q in zip(pols[0].named_parameters(), pols[1].parameters()):
        assert torch.equal(p, q) and torch.equal(p.grad, q.grad), name
    assert torch.equal(pols[0]._moments, pols[1]._moments)
    assert all(torch.isfinite(p).all() for p in pols[0].parameters())
    env.close()


def test_step_modes_beyond_the_fused_kernels_largest_minibatch():
    """The fused kernel leaves one statistics row per workgroup of eight tiles: 2 048 rows = 262 144 samples per minibatch.  One tile more and automatic
    mode keeps the four launches (same bits as mode "four_launch"), the fused modes are refused - an error code, not a write past the rows."""
    import torch
    from q1physrl_amd import _lib, ppo
    mb = 262144 + 32
    pols = [_policy(5, 2.0)]
    pols.append(copy.deepcopy(pols[0]))
    env, full, total = _train_batch(2048, 8, pols[0])
    klc = torch.tensor(0.2, device="cuda")
    idx = torch.randint(0, total, (mb,), device="cuda")                      # (rows repeat: the minibatch is larger than the train batch)
    hp = (3e-4, (0.9, 0.999), 1e-8)
    for mode, pol in zip(["auto", "four_launch"], pols):
        env._dev.learner_set_step_mode(mode)
        nat = ppo.NativeStep(pol, env, mb, splits=32)
        nat.step(full, idx, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, adam=hp)
        torch.cuda.synchronize()
    for (name, p), q in zip(pols[0].named_parameters(), pols[1].parameters()):
        assert torch.equal(p, q) and torch.equal(p.grad, q.grad), name
    for mode in ("fused", "fused_dw1"):
        env._dev.learner_set_step_mode(mode)
        with pytest.raises(_lib.Q1EnvError):
            nat.step(full, idx, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, adam=hp)
    env._dev.learner_set_step_mode("auto")
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


EXCHANGE_MODES = ["auto", "agent", "census_fail"]


def _set_mode(env, mode):
    env._dev.learner_set_exchange_mode(mode)


def _assert_mode(nat, mode):
    """status[2], [3] = the exchange mode the policy / value group actually ran in (0 agent scope, else 1 + the shared XCD): what was asked
    for is what ran.  "auto" on an unpartitioned MI355X means L2-local (the dispatcher deals workgroups to the XCDs round-robin; a device on
    which the census fails would run - correctly - in agent scope, and this assertion is how the suite would learn of it)."""
    st = nat.persistent_status()
    assert st[0] == 0, st
    if mode == "auto":
        assert 1 <= st[2] <= 8 and 1 <= st[3] <= 8, st
    else:
        assert st[2] == 0 and st[3] == 0, st


@pytest.mark.parametrize("mode", EXCHANGE_MODES)
def test_persistent_learner_one_step_against_the_four_launch_step_and_autograd(mode):
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
    _set_mode(env, mode)
    hp = (3e-4, (0.9, 0.999), 1e-8)
    w0 = [p.detach().clone() for p in pol_a.parameters()]
    a.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
    n = b.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp, steps=1)
    torch.cuda.synchronize()
    assert n == 1
    _assert_mode(b, mode)
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


@pytest.mark.parametrize("mode", EXCHANGE_MODES)
def test_persistent_learner_epoch_of_391_steps_against_391_four_launch_steps(mode):
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
    _set_mode(env, mode)
    hp = (5e-6, (0.9, 0.999), 1e-8)
    w0 = [p.detach().clone() for p in pol_a.parameters()]
    for _ in range(391):
        a.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
    n = b.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp)
    torch.cuda.synchronize()
    assert n == 391
    _assert_mode(b, mode)
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


@pytest.mark.parametrize("mode", EXCHANGE_MODES)
def test_persistent_learner_consecutive_launches_stay_synchronised(mode):
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
    _set_mode(env, mode)
    hp = (5e-5, (0.9, 0.999), 1e-8)
    w0 = [p.detach().clone() for p in pol_a.parameters()]
    p1 = perm.reshape(1, -1).contiguous()
    for launch in range(6):
        a.cursor.zero_()
        for _ in range(60):
            a.step(full, perm, 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
        b.epochs(full, p1, 0.3, 10.0, 1.0, 0.01, klc, hp, steps=60)
        torch.cuda.synchronize()
        _assert_mode(b, mode)
        assert int(b.adam_state[:8].view(torch.int64)[0]) == 60 * (launch + 1)
        for (name, pa), pb, w in zip(pol_a.named_parameters(), pol_b.parameters(), w0):
            r = _rel(pb.detach() - w, pa.detach() - w)
            assert torch.isfinite(pb).all() and r < 5e-2, (launch, name, r)
    env.close()


@pytest.mark.parametrize("mode", EXCHANGE_MODES)
def test_ppo_learner_update_persistent_equals_per_step_loop(mode):
    """PPOLearner.update with the persistent learner (one dispatch per update) against the same learner driving q1env_learner_sgd_step
    per minibatch: same permutations (same generator), same statistics to 1e-3, same adaptive-KL decision."""
    import copy
    import torch
    from q1physrl_amd import policy as P, ppo, sampler as S
    pol_a = _policy(9, 1.0)
    pol_b = copy.deepcopy(pol_a)
    cfg, env = make_env(128, time_limit=1.0)
    _set_mode(env, mode)
    smp = S.GpuSampler(env, P.FusedPolicyForward(pol_a, env), horizon=16)
    tr = smp.collect()
    adv, vt = smp.advantages(tr, 0.99, 0.95)
    outs = []
    for pol, persistent in ((pol_a, False), (pol_b, True)):
        lr_ = ppo.PPOLearner(pol, cfg.action_range, lr=5e-6, num_sgd_iter=3, minibatch_size=128, env=env, native=True, persistent=persistent, seed=3)
        outs.append([lr_.update(tr, adv, vt) for _ in range(4)])          # four updates: the later ones start from the earlier ones' state
        if persistent:
            _assert_mode(lr_._native, mode)
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


def _gather_probe_policy():
    """A policy through which a step's gathered rows can be read back EXACTLY from the exchange buffers it leaves behind: hidden unit u of
    layer 1 sees observation feature u % 6 alone (weight 1, no bias), layer 2 is zero (H2 = 0, outputs = b3 = 0), W3 routes ONE output's
    loss gradient to every unit: dZ2[b][u] = dY[b][1] (policy: logit 1 of key 0) / dY[b][0] (value)."""
    import torch
    from q1physrl_amd import policy as P
    pol = P.Q1Policy().cuda()
    with torch.no_grad():
        for p in pol.parameters():
            p.zero_()
        for seq, out in ((pol.pi, 1), (pol.vf, 0)):
            seq[0].weight[torch.arange(256), torch.arange(256) % 6] = 1.0
            seq[4].weight[out, :] = 1.0
    return pol


def _fragment_rows(buf_u8, torch, f32=False):
    """An exchange array in operand-fragment order -> float32 [128 samples][256 units].  float16 kernel (csrc/q1learner_persist.hpp Net):
    [4 tiles w][16 K-steps s][64 lanes (h, c)][8], element (w, s, h, c, e) = sample 32 w + c, unit 16 s + 8 h + e; float32 kernel
    (csrc/q1learner_persist32.hpp): [4 w][32 s][64 lanes][4], unit 8 s + 4 h + e."""
    if f32:
        return buf_u8.view(torch.float32).reshape(4, 32, 2, 32, 4).permute(0, 3, 1, 2, 4).reshape(128, 256).clone()
    t = buf_u8.view(torch.float16).reshape(4, 16, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(128, 256)
    return t.float()


@pytest.mark.parametrize("kernel", ["f16", "f32"])
@pytest.mark.parametrize("mode", ["auto", "agent"])
def test_persistent_learner_gathers_exactly_the_scheduled_rows(mode, kernel):
    """tolerance tests cannot see one wrong row in 128, so here every row of the train batch ENCODES ITS OWN INDEX -
    the observation in base 9 over its six features (read back through H1 = tanh(x), which the last step leaves in the exchange buffer),
    adv and vtarg as index mod 1024 (read back through dZ2 = the loss gradient of one output, exact in float16) - and after launches
    of k steps (k = 1 .. 9 with 8 minibatches per epoch: every window of an epoch as the last one, then the wrap into the second epoch's
    permutation; then 40 and 391 x 2 + 5 steps of a 50 048-row batch) the rows the last step consumed must be the scheduled ones, all 128,
    for the observation request AND for the loss inputs of BOTH networks.  lr = 0: the probe weights stay what they are."""
    import torch
    from q1physrl_amd import ppo
    pol = _gather_probe_policy()
    cfg, env = make_env(128, time_limit=1.0)
    _set_mode(env, mode)
    env._dev.learner_set_loss_scale(256.0, 1.0)
    levels = torch.tanh(torch.arange(-4, 5, device="cuda", dtype=torch.float32) * 0.25)
    klc = torch.tensor(0.0, device="cuda")
    hp = (0.0, (0.9, 0.999), 1e-8)
    for total, step_counts in ((1024, list(range(1, 10)) + [16]), (50048, [40, 391 * 2 + 5])):
        r = torch.arange(total, device="cuda")
        digits = torch.stack([(r // 9 ** i) % 9 for i in range(6)], dim=1)
        code = (r % 1024).float()
        full = {"obs": ((digits - 4).float() * 0.25).contiguous(), "old_logits": torch.zeros((total, 10), device="cuda"),
                "keys_packed": torch.zeros((total,), dtype=torch.uint8, device="cuda"), "mouse": torch.zeros((total,), device="cuda"),
                "logp": torch.zeros((total,), device="cuda"), "adv": (code / 128.0).contiguous(), "value": torch.zeros((total,), device="cuda"),
                "vtarg": (code * 0.5).contiguous()}
        # logp_old = the log-probability of (keys 0, mouse 0) under all-zero outputs, so that ratio = 1: computed by the learner's own forward + loss
        # is not needed to the last bit - dY = 128 adv ratio is rounded to float16, which absorbs a relative 1e-6
        from q1physrl_amd.policy import Q1PhysActionDist
        dist = Q1PhysActionDist(torch.zeros((1, 10), device="cuda"), float(cfg.action_range), 4, -1, True)
        full["logp"] += float(dist.logp(torch.zeros((1, 4), dtype=torch.long, device="cuda"), torch.zeros((1, 1), device="cuda"))[0])
        g = torch.Generator(device="cuda").manual_seed(total)
        perms = torch.stack([torch.randperm(total, device="cuda", generator=g) for _ in range(3)]).contiguous()
        spe = total // 128
        nat = ppo.NativeStep(pol, env, 128, splits=8)
        f32 = kernel == "f32"
        act = 131072 if f32 else 65536                                  # bytes of one [128][256] exchange array
        for k in step_counts:
            nat.epochs(full, perms, 0.3, 1e9, 1.0, 0.0, klc, hp, steps=k, refresh_images=False, f32=f32)
            torch.cuda.synchronize()
            _assert_mode(nat, mode)
            n = k - 1
            want = perms[n // spe, (n % spe) * 128:(n % spe) * 128 + 128]
            for net in (0, 1):
                lay = env._dev.learner_persistent_layout(total, net)
                pws = nat._pws
                h1 = _fragment_rows(pws[lay["h1x"] + (n & 1) * act: lay["h1x"] + (n & 1) * act + act], torch, f32)
                dz2 = _fragment_rows(pws[lay["dz2x"]: lay["dz2x"] + act], torch, f32)
                # observation rows: unit i (< 6) of H1 is tanh of feature i -> the digit -> the row
                dig = (h1[:, :6, None] - levels[None, None, :]).abs().argmin(dim=2)
                assert float((h1[:, :6] - levels[dig]).abs().max()) < 2e-3
                got_obs = (dig * (9 ** torch.arange(6, device="cuda"))[None, :]).sum(dim=1)
                assert torch.equal(got_obs, want), (total, k, net, "observation rows", (got_obs != want).nonzero().flatten().tolist()[:8])
                # loss rows: policy dY[.][1] = 256 (-adv ratio) (a - p) = 128 adv = code; value dY = 2 (0 - vtarg) = -code: integers below 1 024, exact in float16
                col = dz2[:, 0]
                assert float((dz2 - col[:, None]).abs().max()) == 0.0            # every unit carries the same value (W3 row of ones, H2 = 0)
                sign = (256.0 if f32 else 1.0) if net == 0 else -1.0        # (the float32 kernel carries no loss scale: dY = adv / 2)
                got = (col * sign).round().long()
                assert float((col * sign - got).abs().max()) < 1e-2
                assert torch.equal(got, want % 1024), (total, k, net, "loss rows", (got != want % 1024).nonzero().flatten().tolist()[:8])
        # the weights did not move (lr = 0) and the step count did
        assert int(nat.adam_state[:8].view(torch.int64)[0]) == sum(step_counts)
    env._dev.learner_set_loss_scale(0.0, 0.0)
    env.close()


def test_persistent_learner_refuses_a_schedule_that_runs_past_the_index_list():
    """steps beyond what the permutations hold used to read int64 indices past the end of idx_dev.  NativeStep.epochs raises, and
    the C ABI itself (idx_rows, ABI v6) refuses with Q1ENV_ERR_INVALID_ARG before anything is launched."""
    import torch
    from q1physrl_amd import ppo, _lib
    pol = _policy(3, 1.0)
    env, full, total = _train_batch(64, 8, pol)                    # 512 rows: 4 minibatches per epoch
    klc = torch.tensor(0.2, device="cuda")
    nat = ppo.NativeStep(pol, env, 128, splits=8)
    perms = torch.randperm(total, device="cuda").reshape(1, -1).contiguous()
    hp = (5e-6, (0.9, 0.999), 1e-8)
    with pytest.raises(ValueError, match="steps"):
        nat.epochs(full, perms, 0.3, 10.0, 1.0, 0.01, klc, hp, steps=5)
    assert nat.epochs(full, perms, 0.3, 10.0, 1.0, 0.01, klc, hp, steps=4) == 4
    torch.cuda.synchronize()
    assert nat.persistent_status()[0] == 0
    L = _lib
    ol = full["old_logits"]
    b = L.Q1LearnerBatch(128, perms.data_ptr(), None, full["obs"].data_ptr(), ol.data_ptr(), ol.shape[1], full["keys_packed"].data_ptr(),
                         full["mouse"].data_ptr(), full["logp"].data_ptr(), full["adv"].data_ptr(), full["value"].data_ptr(), full["vtarg"].data_ptr(),
                         0.3, 10.0, 1.0, 0.01, klc.data_ptr(), None, 0, nat.saturation.data_ptr())
    for idx_rows, steps, stride in ((total, 5, total), (total - 1, 4, total), (2 * total - 1, 8, total), (0, 1, total)):
        with pytest.raises(L.Q1EnvError, match="schedule|idx_rows"):
            env._dev.learner_sgd_epochs_dev(nat.pi, nat.vf, nat._pws.data_ptr(), b, total, idx_rows, steps, 4, stride, 5e-6, 0.9, 0.999, 1e-8, nat.adam_state.data_ptr())
    env.close()


def test_persistent_learner_whole_update_under_the_assertion_build():
    """VERDICT r5 items 1b / missing 4: the -DQ1_CHECK build of the persistent learner (libq1env_check.so) compares every exchange offset
    with the group's workspace, every gathered row index with the number of rows, every schedule position with the index list and every
    barrier reading with its range - a failure is a status word, not a memory fault.  tools/soak_plearner_check.py runs a whole update of
    the reference's shape (30 epochs x 391 steps = 11 730 steps) in both exchange modes - and 10 epochs of the float32 kernel in both - with zero
    failures, then plants a row index beyond
    the batch and expects status 0x102 (and a process that is still alive).  Subprocess: the assertion library is chosen before the
    binding loads; the product library reports "not an assertion build"."""
    import os
    import subprocess
    import sys
    from q1physrl_amd import build
    cfg, env = make_env(128)
    assert env._dev.learner_debug_counters() == (0, 0, 0, 0, 0)
    env.close()
    so = build.build_lib(check=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_plearner_check.py")], capture_output=True, text=True, timeout=900, cwd=root,
                       env=dict(os.environ, Q1ENV_LIB_PATH=so))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("soak_plearner_check ok: 31280 steps") and " 0 failures" in last and "planted index -> status 0x102" in last, last


# ---- the float32-arithmetic persistent learner (q1env_learner_sgd_epochs_f32, csrc/q1learner_persist32.hpp): RLlib's own arithmetic


def _torch_f32_steps(pol, full, perm, n_steps, action_range, klc, hp, torch, ppo):
    """n_steps PPO SGD steps of 128-sample minibatches in plain float32 torch (autograd through the modules and ppo.ppo_loss, torch.optim.Adam):
    what RLlib's TF learner computes, and the reference the float32 kernel is held to."""
    opt = torch.optim.Adam(pol.parameters(), lr=hp[0], betas=hp[1], eps=hp[2])
    stats = []
    for s in range(n_steps):
        idx = perm[s * 128:(s + 1) * 128]
        mb = {"obs": full["obs"][idx], "old_logits": full["old_logits"][idx], "mouse": full["mouse"][idx].reshape(-1, 1), "logp": full["logp"][idx],
              "adv": full["adv"][idx], "value": full["value"][idx], "vtarg": full["vtarg"][idx],
              "keys": ((full["keys_packed"][idx].reshape(-1, 1).long() >> torch.arange(4, device="cuda")) & 1)}
        opt.zero_grad(set_to_none=True)
        loss, st = ppo.ppo_loss(pol, mb, action_range, 0.3, 10.0, 1.0, 0.01, klc)
        loss.backward()
        opt.step()
        stats.append(st)
    return stats


@pytest.mark.parametrize("mode", ["auto", "agent"])
def test_f32_persistent_learner_one_step_is_as_close_to_float64_autograd_as_float32_torch(mode):
    """One step of q1env_learner_sgd_epochs_f32 on 128 rows against torch autograd of the same minibatch in FLOAT64 (the truth) and in float32
    (what RLlib's TF learner computes): the value network's gradients within 1e-6 of the truth (relative Frobenius norm per tensor; measured
    1 - 2e-7, like float32 torch), the policy network's within 6e-4 and no further from it than 1.5 x float32 torch's own distance (measured
    2 - 4e-4 against torch's 3.5 - 6e-4: the float32 conditioning of the loss - the squashed-Gaussian pre-image of the mouse action - not
    of the matrix products; the float16 kernel sits at 3 - 9e-4).  The Adam step, moments and statistics accordingly; no loss scale,
    nothing saturates."""
    import copy
    import torch
    from q1physrl_amd import ppo
    pol_b = _policy(5, 2.0)
    pol_c = copy.deepcopy(pol_b)
    env, full, total = _train_batch(64, 8, pol_b)
    _set_mode(env, mode)
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    idx = perm[:128]

    def autograd(dtype):
        p = copy.deepcopy(pol_b).to(dtype)
        mb = {k: full[k][idx].to(dtype) for k in ("obs", "old_logits", "logp", "adv", "value", "vtarg")}
        mb["mouse"] = full["mouse"][idx].reshape(-1, 1).to(dtype)
        mb["keys"] = ((full["keys_packed"][idx].reshape(-1, 1).long() >> torch.arange(4, device="cuda")) & 1)
        loss, _ = ppo.ppo_loss(p, mb, float(env.config.action_range), 0.3, 10.0, 1.0, 0.01, klc.to(dtype))
        loss.backward()
        return [q.grad.double() for q in p.parameters()]

    g64, g32 = autograd(torch.float64), autograd(torch.float32)
    b = ppo.NativeStep(pol_b, env, 128, splits=8)
    hp = (3e-4, (0.9, 0.999), 1e-8)
    w0 = [p.detach().clone() for p in pol_b.parameters()]
    n = b.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp, steps=1, f32=True)
    torch.cuda.synchronize()
    assert n == 1
    _assert_mode(b, mode)
    for (name, pb), t64, t32 in zip(pol_b.named_parameters(), g64, g32):
        mine, torchs = _rel(pb.grad.double(), t64), _rel(t32, t64)
        if name.startswith("vf."):
            assert mine < 1e-6, (name, mine, torchs)
        else:
            assert mine < 6e-4 and mine <= 1.5 * torchs + 1e-6, (name, mine, torchs)
    st = _torch_f32_steps(pol_c, full, perm, 1, float(env.config.action_range), klc, hp, torch, ppo)[0]
    for (name, pb), pc, w in zip(pol_b.named_parameters(), pol_c.parameters(), w0):
        db, dc = pb.detach() - w, pc.detach() - w
        assert float(dc.abs().max()) > 0 and _rel(db, dc) < 2e-2, (name, _rel(db, dc))          # first Adam step: lr sign(g) - a ~0 gradient's sign may differ
    assert int(b.adam_state[:8].view(torch.int64)[0]) == 1
    sb = b.stats_acc.cpu().numpy()
    for k, key in ((0, "entropy"), (1, "kl"), (2, "policy_loss"), (4, "vf_loss")):
        assert abs(sb[k] - float(st[key])) <= 2e-5 * max(1.0, abs(float(st[key]))), (key, sb[k], float(st[key]))
    assert int(b.saturation.abs().sum()) == 0
    env.close()


@pytest.mark.parametrize("mode", ["auto", "agent", "census_fail"])
def test_f32_persistent_learner_epoch_against_a_torch_float32_loop(mode):
    """An epoch of the reference's train batch (391 minibatches of 128, lr 5e-6) as one dispatch of the float32 kernel against 391 steps of
    float32 torch (autograd + torch.optim.Adam): the accumulated parameter change agrees to 2 % per tensor (the summation orders differ; Adam
    turns a ~0 gradient's sign into a full step), the statistics to 1e-4; a second launch continues from the first one's state."""
    import copy
    import torch
    from q1physrl_amd import ppo
    pol_b = _policy(7, 1.0)
    pol_c = copy.deepcopy(pol_b)
    env, full, total = _train_batch(128, 391, pol_b)
    _set_mode(env, mode)
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    b = ppo.NativeStep(pol_b, env, 128, splits=8)
    hp = (5e-6, (0.9, 0.999), 1e-8)
    w0 = [p.detach().clone() for p in pol_b.parameters()]
    n = b.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp, f32=True)
    torch.cuda.synchronize()
    assert n == 391
    _assert_mode(b, mode)
    stats = _torch_f32_steps(pol_c, full, perm, 391, float(env.config.action_range), klc, hp, torch, ppo)
    for (name, pb), pc, w in zip(pol_b.named_parameters(), pol_c.parameters(), w0):
        r = _rel(pb.detach() - w, pc.detach() - w)
        assert torch.isfinite(pb).all() and r < 2e-2, (name, r)
    sb = b.stats_acc.cpu().numpy()
    for k, key in ((0, "entropy"), (1, "kl"), (2, "policy_loss"), (4, "vf_loss")):
        want = float(sum(float(s_[key]) for s_ in stats))
        assert abs(sb[k] - want) <= 1e-4 * max(391.0, abs(want)), (key, sb[k], want)
    b.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp, steps=5, f32=True)
    torch.cuda.synchronize()
    assert int(b.adam_state[:8].view(torch.int64)[0]) == 396 and b.persistent_status()[0] == 0
    env.close()


def test_ppo_learner_precision_f32_runs_the_float32_kernel():
    """PPOLearner(precision="f32"): update() dispatches q1env_learner_sgd_epochs_f32 - same statistics as the float16 persistent learner to
    float16 rounding, no saturation report - and refuses configurations the persistent learner does not serve."""
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
    assert ppo.PPOLearner(pol_a, cfg.action_range, env=env, native=True).precision == "f16"       # the default (STATE.md "fp32 control": why it stays)
    for pol, prec in ((pol_a, "f16"), (pol_b, "f32")):
        lr_ = ppo.PPOLearner(pol, cfg.action_range, lr=5e-6, num_sgd_iter=3, minibatch_size=128, env=env, native=True, persistent=True, seed=3, precision=prec)
        outs.append([lr_.update(tr, adv, vt) for _ in range(3)])
    torch.cuda.synchronize()
    for oa, ob in zip(*outs):
        assert oa["sgd_steps"] == ob["sgd_steps"] == 3 * 16 and oa["kl_coeff"] == ob["kl_coeff"]
        for k in ("entropy", "kl", "policy_loss", "total_loss", "vf_loss"):
            assert abs(oa[k] - ob[k]) <= 3e-3 * max(1.0, abs(oa[k])), (k, oa[k], ob[k])
        assert ob["grad_saturated_pi"] == 0 and ob["grad_saturated_vf"] == 0
    with pytest.raises(ValueError, match="precision"):
        ppo.PPOLearner(pol_a, cfg.action_range, env=env, native=True, precision="bf16")
    with pytest.raises(ValueError, match="persistent"):
        ppo.PPOLearner(pol_b, cfg.action_range, lr=5e-6, num_sgd_iter=1, minibatch_size=256, env=env, native=True, precision="f32").update(tr, adv, vt)
    env.close()
