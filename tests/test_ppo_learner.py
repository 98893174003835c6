"""CPU: PPO loss (torch) vs the NumPy oracle on a synthetic batch, and the multi-GPU learner model - gradient
all-reduce on one flat bucket - with world_size-2 gloo: both ranks must end with identical parameters, equal to a
single process training on the concatenated batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ppo_oracle as PO
from q1physrl_amd import policy as P, ppo

AR = 10.0


def synth(rng, t, n):
    obs = rng.normal(0, 1, (t + 1, n, 6)).astype(np.float32)
    traj = {"obs": torch.from_numpy(obs), "keys": torch.from_numpy(rng.integers(0, 16, (t, n), dtype=np.uint8)),
            "mouse": torch.from_numpy(rng.uniform(-9.5, 9.5, (t, n)).astype(np.float32)),
            "logp": torch.from_numpy(rng.normal(-5, 1, (t, n)).astype(np.float32)),
            "value": torch.from_numpy(rng.normal(0, 1, (t + 1, n)).astype(np.float32)),
            "reward": torch.from_numpy(rng.normal(0, 1, (t, n)).astype(np.float32)),
            "done": torch.from_numpy((rng.random((t, n)) < 0.05).astype(np.uint8))}
    adv, vtarg = PO.gae(traj["reward"].numpy(), traj["value"].numpy(), traj["done"].numpy(), 0.99, 0.95)
    return traj, torch.from_numpy(adv), torch.from_numpy(vtarg)


def test_ppo_loss_matches_numpy_oracle():
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    pol = P.Q1Policy().double()
    b = 256
    batch = {"obs": torch.from_numpy(rng.normal(0, 1, (b, 6))), "keys": torch.from_numpy(rng.integers(0, 2, (b, 4))),
             "mouse": torch.from_numpy(rng.uniform(-9.5, 9.5, (b, 1))), "logp": torch.from_numpy(rng.normal(-5, 1, b)),
             "value": torch.from_numpy(rng.normal(0, 1, b)), "adv": torch.from_numpy(rng.normal(0, 1, b)),
             "vtarg": torch.from_numpy(rng.normal(0, 1, b)), "old_logits": torch.from_numpy(rng.normal(0, 0.5, (b, 10)))}
    loss, st = ppo.ppo_loss(pol, batch, AR, 0.3, 100.0, 1.0, 0.01, 0.2)
    with torch.no_grad():
        logits, value = pol(batch["obs"])
    nb = {k: v.numpy() for k, v in batch.items()}
    want, kl, ent = PO.ppo_loss(logits.numpy(), value.numpy(), nb, AR, 0.3, 100.0, 1.0, 0.01, 0.2)
    assert abs(float(loss.detach()) - want) < 1e-9 * max(1, abs(want))
    assert abs(float(st["kl"]) - kl) < 1e-9 and abs(float(st["entropy"]) - ent) < 1e-9
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in pol.parameters())


def test_learner_update_reduces_loss_and_adapts_kl():
    rng = np.random.default_rng(1)
    torch.manual_seed(1)
    pol = P.Q1Policy()
    traj, adv, vtarg = synth(rng, 16, 64)
    with torch.no_grad():        # make logp consistent with the current policy so ratio starts at 1
        logits, value = pol(traj["obs"][:16].reshape(-1, 6))
        d = P.Q1PhysActionDist(logits, AR)
        keys = ((traj["keys"].reshape(-1, 1).long() >> torch.arange(4)) & 1)
        traj["logp"] = d.logp(keys, traj["mouse"].reshape(-1, 1)).reshape(16, 64)
    lrn = ppo.PPOLearner(pol, AR, lr=1e-3, num_sgd_iter=4, minibatch_size=256, kl_target=0.0036)
    s1 = lrn.update(traj, adv, vtarg)
    s2 = lrn.update(traj, adv, vtarg)
    assert s2["total_loss"] < s1["total_loss"] and s1["sgd_steps"] == 16
    assert lrn.kl_coeff != 0.2                                   # adapted one way or the other


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    traj, adv, vtarg = synth(rng, 8, 32)                         # the GLOBAL batch, identical on both ranks
    torch.manual_seed(3)
    pol = P.Q1Policy().double()
    sl = slice(rank * 16, (rank + 1) * 16)                       # this rank's env shard
    mine = {k: (v[:, sl].double() if v.dtype == torch.float32 else v[:, sl]) for k, v in traj.items()}
    lrn = ppo.PPOLearner(pol, AR, lr=1e-3, num_sgd_iter=3, minibatch_size=10 ** 9, seed=0)   # full-batch SGD: deterministic
    lrn.update(mine, adv[:, sl].double(), vtarg[:, sl].double())
    flat = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.manual_seed(3)
        ref = P.Q1Policy().double()
        dist_world = lrn.world
        single = ppo.PPOLearner(ref, AR, lr=1e-3, num_sgd_iter=3, minibatch_size=10 ** 9, seed=0)
        single.world = 1
        full = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in traj.items()}
        single.update(full, adv.double(), vtarg.double())
        ref_flat = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
        np.save(out, np.array([float(torch.equal(gathered[0], gathered[1])), float((gathered[0] - ref_flat).abs().max()), dist_world]))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_process(tmp_path):
    out = str(tmp_path / "r.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    same, err, world = np.load(out)
    assert same == 1.0 and world == 2 and err < 1e-10


def test_fused_loss_needs_an_env_handle():
    """fused_loss=True runs q1env_ppo_loss_grad on an env handle's stream: constructing the learner without one fails loudly."""
    import pytest
    import torch
    from q1physrl_amd import policy as P, ppo
    with pytest.raises(ValueError, match="fused_loss=True / native=True need env"):
        ppo.PPOLearner(P.Q1Policy(), 10.0, fused_loss=True)


def test_closed_form_loss_gradient_matches_autograd():
    """oracle/ppo_oracle.ppo_loss_grad (the float64 restatement of what the q1env_ppo_loss_grad kernel computes) against float64
    torch autograd of ppo.ppo_loss: gradients with respect to the policy outputs and the five statistics, on a batch that
    exercises the clamps, ratios outside the clip range and clipped value errors."""
    import torch
    from q1physrl_amd import ppo
    rng = np.random.default_rng(4)
    bsz = 3000
    old = rng.normal(0, 1, (bsz, 10)) * np.array([1, 1, 1, 1, 1, 1, 1, 1, 1.5, 0.7])
    new = old + 0.3 * rng.normal(0, 1, (bsz, 10))
    new[:150, 8] = 3.4; new[150:300, 8] = -3.7; new[300:400, 9] = 2.5; new[400:500, 9] = -21.0
    keys = rng.integers(0, 2, (bsz, 4))
    mouse = rng.uniform(-9.9, 9.9, (bsz, 1))
    lp_old = PO.dist_terms(old, keys, mouse, AR)[0] + 0.4 * rng.normal(0, 1, bsz)
    v_old = 50 * rng.normal(0, 1, bsz)
    v_new = v_old + 60 * rng.normal(0, 1, bsz) * (rng.random(bsz) < 0.5)
    b = {"keys": keys, "mouse": mouse, "logp": lp_old, "adv": rng.normal(0, 1, bsz), "value": v_old, "vtarg": v_old + 80 * rng.normal(0, 1, bsz),
         "old_logits": old}
    dl, dv, st = PO.ppo_loss_grad(new, v_new, b, AR, 0.3, 100.0, 1.0, 0.01, 0.37)
    lt, vt = torch.tensor(new, requires_grad=True), torch.tensor(v_new, requires_grad=True)

    class Fixed(torch.nn.Module):
        def forward(self, obs):
            return lt, vt
    loss, tst = ppo.ppo_loss(Fixed(), {k: (None if v is None else torch.as_tensor(v)) for k, v in {**b, "obs": None}.items()}, AR, 0.3, 100.0,
                             1.0, 0.01, 0.37)
    loss.backward()
    # (relative: the rows with log_std clamped at -20 have std = 2e-9 and astronomically large - but equal - KL gradients)
    rel = lambda a, r: np.max(np.abs(a - r) / np.maximum(np.abs(r), 1e-9))
    assert rel(dl, lt.grad.numpy()) <= 1e-9 and rel(dv, vt.grad.numpy()) <= 1e-12
    assert np.abs(lt.grad.numpy()[:300, 8]).max() == 0.0 and np.abs(dl[300:500, 9]).max() == 0.0      # gated by the clamps
    for k in ppo.STAT_KEYS:
        assert abs(st[k] - float(tst[k])) <= 1e-10 * max(1.0, abs(float(tst[k]))), k
    assert abs(st["total_loss"] - PO.ppo_loss(new, v_new, b, AR, 0.3, 100.0, 1.0, 0.01, 0.37)[0]) <= 1e-10


def test_discrete_mouse_closed_form_gradient_matches_autograd():
    """Discrete mouse (Config.discrete_yaw_steps = 5: the Tuple's last child is Discrete(11), reference env.py:216-219, taken
    through ModelCatalog's Categorical, action_dist.py:221-222): oracle/ppo_oracle.ppo_loss_grad_discrete (float64 restatement of
    the kernel's yaw_mode == 2 branch) against float64 torch autograd of ppo.ppo_loss over the torch Categorical."""
    import torch
    from q1physrl_amd import ppo
    from q1physrl_amd.policy import Q1PhysActionDist, policy_row_width
    rng = np.random.default_rng(8)
    bsz, steps = 2000, 5
    width = policy_row_width(4, steps)
    assert width == 19
    old = rng.normal(0, 1.2, (bsz, width))
    new = old + 0.3 * rng.normal(0, 1, (bsz, width))
    keys = rng.integers(0, 2, (bsz, 4))
    mouse = rng.integers(0, 2 * steps + 1, (bsz, 1)).astype(np.float64)
    d_old = Q1PhysActionDist(torch.tensor(old), AR, 4, steps)
    lp_old = d_old.logp(torch.tensor(keys), torch.tensor(mouse)).numpy() + 0.4 * rng.normal(0, 1, bsz)
    v_old = 50 * rng.normal(0, 1, bsz)
    v_new = v_old + 60 * rng.normal(0, 1, bsz) * (rng.random(bsz) < 0.5)
    b = {"keys": keys, "mouse": mouse, "logp": lp_old, "adv": rng.normal(0, 1, bsz), "value": v_old, "vtarg": v_old + 80 * rng.normal(0, 1, bsz),
         "old_logits": old}
    dl, dv, st = PO.ppo_loss_grad_discrete(new, v_new, b, steps, 0.3, 100.0, 1.0, 0.01, 0.37)
    lt, vt = torch.tensor(new, requires_grad=True), torch.tensor(v_new, requires_grad=True)

    class Fixed(torch.nn.Module):
        def forward(self, obs):
            return lt, vt
    loss, tst = ppo.ppo_loss(Fixed(), {k: (None if v is None else torch.as_tensor(v)) for k, v in {**b, "obs": None}.items()}, AR, 0.3, 100.0,
                             1.0, 0.01, 0.37, discrete_yaw_steps=steps)
    loss.backward()
    assert np.max(np.abs(dl - lt.grad.numpy())) <= 1e-12 * max(1.0, np.abs(dl).max()) + 1e-15
    assert np.max(np.abs(dv - vt.grad.numpy())) <= 1e-12 * np.abs(dv).max()
    for k in ppo.STAT_KEYS:
        assert abs(st[k] - float(tst[k])) <= 1e-10 * max(1.0, abs(float(tst[k]))), k
    # the sampled-step bookkeeping of the torch distribution: inverse-CDF sampling reproduces softmax frequencies
    g = torch.Generator().manual_seed(0)
    big = Q1PhysActionDist(torch.tensor(np.tile(new[:1], (200000, 1))), AR, 4, steps)
    _, m = big.sample(generator=g)
    freq = np.bincount(m.reshape(-1).long().numpy(), minlength=11) / 200000
    p = np.exp(new[0, 8:] - new[0, 8:].max()); p /= p.sum()
    assert np.max(np.abs(freq - p)) < 5e-3
    kd, md = big.deterministic_sample()
    assert int(md[0, 0]) == int(np.argmax(new[0, 8:]))


def test_dynamic_loss_scale_choice_is_a_power_of_two_with_headroom():
    """the next update's float16 loss scales from this update's largest travelled element - exact powers of two,
    the largest element lands within (1/16, 1/8] of float16's largest finite value, bounded ranges, no change without information."""
    import math
    lr = ppo.PPOLearner(P.Q1Policy(), 10.0)                    # (CPU: only the arithmetic of the choice is exercised)
    lr.native = True
    assert (lr.pi_upscale, lr.value_downscale) == (256.0, 1.0)
    for max_pi in (3.0, 94.0 * 256, 12000.0, 3.55e6, 4e7, 1e-3):
        for max_vf in (237.0, 7100.0, 3.0e5, 0.5):
            up, down = lr._next_loss_scales(max_pi, max_vf)
            assert math.frexp(up)[0] == 0.5 and math.frexp(down)[0] == 0.5               # exact powers of two
            raw_pi, raw_vf = max_pi / lr.pi_upscale, max_vf * lr.value_downscale
            if lr.PI_UPSCALE_RANGE[0] < up < lr.PI_UPSCALE_RANGE[1]:
                assert 65504.0 / 16 < raw_pi * up <= 65504.0 / 8
            if lr.VALUE_DOWNSCALE_RANGE[0] < down < lr.VALUE_DOWNSCALE_RANGE[1]:
                assert 65504.0 / 16 < raw_vf / down <= 65504.0 / 8
            assert lr.PI_UPSCALE_RANGE[0] <= up <= lr.PI_UPSCALE_RANGE[1] and lr.VALUE_DOWNSCALE_RANGE[0] <= down <= lr.VALUE_DOWNSCALE_RANGE[1]
    assert lr._next_loss_scales(0.0, float("nan")) == (256.0, 1.0)
    # the reference-configuration run of round 4 ended at 3.55 M (x 256) with 1 832 saturated elements: the choice for such an update
    assert lr._next_loss_scales(3.55e6, 7100.0)[0] == 0.5
