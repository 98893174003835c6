"""CPU: the torch restatement of the reference's action distribution (q1physrl_amd/policy.py) against the
NumPy/SciPy oracle (oracle/dist_oracle.py) and against first principles.  float64 tensors: tolerance 1e-9."""
import warnings

import numpy as np
import pytest
import torch
from scipy import integrate

from oracle import dist_oracle as DO
from q1physrl_amd import policy as P

LOW, HIGH = -10.08, 10.08


def rand_inputs(rng, b):
    x = rng.normal(0, 1.5, (b, 10))
    x[:, 8] = rng.uniform(-4, 4, b)        # mean, partly outside the +-3 clip
    x[:, 9] = rng.uniform(-3, 2.5, b)      # log_std, partly above the clip at 2
    return x


def test_mouse_logp_entropy_kl_match_oracle():
    rng = np.random.default_rng(0)
    x = rand_inputs(rng, 500)
    d = P.GaussianSquashedGaussian(torch.from_numpy(x[:, 8:10]), LOW, HIGH)
    a = rng.uniform(LOW * 0.999, HIGH * 0.999, (500, 1))
    assert np.allclose(d.logp(torch.from_numpy(a)).numpy(), DO.mouse_logp(a[:, 0], x[:, 8], x[:, 9], LOW, HIGH), rtol=1e-9, atol=1e-9)
    assert np.allclose(d.entropy().numpy(), DO.mouse_entropy(x[:, 8], x[:, 9], LOW, HIGH), rtol=1e-12)
    y = rand_inputs(rng, 500)
    d2 = P.GaussianSquashedGaussian(torch.from_numpy(y[:, 8:10]), LOW, HIGH)
    assert np.allclose(d.kl(d2).numpy(), DO.mouse_kl(x[:, 8], x[:, 9], y[:, 8], y[:, 9]), rtol=1e-12)
    assert np.all(d.kl(d).numpy() == 0)


def test_mouse_density_integrates_to_one_and_entropy_is_consistent():
    for mean, log_std in ((0.0, 0.0), (1.2, -1.0), (-1.0, -0.3)):      # mass beyond the 1e-6 clip is negligible here
        inp = torch.tensor([[mean, log_std]], dtype=torch.float64)
        d = P.GaussianSquashedGaussian(inp, LOW, HIGH)
        f = lambda a: float(torch.exp(d.logp(torch.tensor([[a]], dtype=torch.float64))))   # noqa: E731
        warnings.simplefilter("ignore")
        total, _ = integrate.quad(f, LOW, HIGH, limit=400, points=[0.0])
        assert abs(total - 1.0) < 1e-4
        h, _ = integrate.quad(lambda a: -f(a) * np.log(max(f(a), 1e-300)), LOW, HIGH, limit=400, points=[0.0])
        assert abs(h - float(d.entropy())) < 1e-3          # closed form (action_dist.py:167-178) = differential entropy


def test_squash_roundtrip_and_bounds():
    d = P.GaussianSquashedGaussian(torch.zeros((1, 2), dtype=torch.float64), LOW, HIGH)
    raw = torch.linspace(-3, 3, 101, dtype=torch.float64).view(-1, 1)
    assert torch.allclose(d._unsquash(d._squash(raw)), raw, atol=1e-9)
    big = d._squash(torch.tensor([[50.0], [-50.0]], dtype=torch.float64))
    assert LOW < float(big[1]) < float(big[0]) < HIGH        # never exactly low / high (SMALL_NUMBER clip)
    g = torch.Generator().manual_seed(0)
    dd = P.GaussianSquashedGaussian(torch.tensor([[0.5, 0.3]], dtype=torch.float64).repeat(20000, 1), LOW, HIGH)
    s = dd.sample(g)
    assert s.shape == (20000, 1) and float(s.min()) > LOW and float(s.max()) < HIGH
    mc_entropy = float(-dd.logp(s).mean())
    assert abs(mc_entropy - float(dd.entropy()[0])) < 0.03


def test_tuple_distribution_matches_oracle_and_policy_shapes():
    rng = np.random.default_rng(3)
    x = rand_inputs(rng, 64)
    dist = P.Q1PhysActionDist(torch.from_numpy(x), 10.08, num_keys=4)
    keys = torch.from_numpy(rng.integers(0, 2, (64, 4)))
    mouse = torch.from_numpy(rng.uniform(-10, 10, (64, 1)))
    want = DO.mouse_logp(mouse[:, 0].numpy(), x[:, 8], x[:, 9], -10.08, 10.08)
    for k in range(4):
        lp0, lp1 = DO.key_logprobs(x[:, 2 * k], x[:, 2 * k + 1])
        want = want + np.where(keys[:, k].numpy() == 1, lp1, lp0)
    assert np.allclose(dist.logp(keys, mouse).numpy(), want, rtol=1e-9, atol=1e-9)
    ent = dist.entropy().numpy()
    assert ent.shape == (64,) and np.all(np.isfinite(ent))
    assert np.allclose(dist.kl(dist).numpy(), 0, atol=1e-12)
    k, m = dist.sample(torch.Generator().manual_seed(1))
    assert k.shape == (64, 4) and m.shape == (64, 1) and set(np.unique(k.numpy())) <= {0, 1}
    assert P.pack_keys(torch.tensor([[1, 0, 1, 1], [0, 0, 0, 0], [1, 1, 1, 1]])).tolist() == [13, 0, 15]
    net = P.Q1Policy()
    assert net.num_parameters() == 137995                     # the WR checkpoint's parameter count
    logits, value = net(torch.zeros(5, 6))
    assert logits.shape == (5, 10) and value.shape == (5,)
    assert P.Q1PhysActionDist.required_model_output_shape(4) == 10


def test_wr_policy_scores_published_reward_in_the_oracle_env():
    """CPU known-answer for the ORACLE: the reference's published world-record policy (weights fixture exported from
    data/checkpoints/wr) reached zero_start_total_reward_mean ~ 5700 in the reference env (README.md:54).  Driving the
    NumPy oracle with that policy (deterministic head: arg-max keys, squashed mean) must give a run of that quality -
    a check of the oracle against the reference's published artefacts that is independent of the generated vectors."""
    import json
    import os
    from oracle import np_oracle as O
    w = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wr_policy.npz")))
    ec = json.loads(str(w["env_config_json"]))
    ec["initial_yaw_range"] = tuple(ec["initial_yaw_range"])
    pol = P.load_rllib_fcnet_weights(P.Q1Policy(), w)
    np.random.seed(0)
    env = O.OracleVectorEnv(O.OracleConfig(**{**ec, "num_envs": 1, "zero_start_prob": 1.0}))
    obs = env.observation()
    total, done, ticks = 0.0, False, 0
    with torch.no_grad():
        while not done:
            logits, _ = pol(torch.from_numpy(obs.astype(np.float32)))
            keys, mouse = P.Q1PhysActionDist(logits, float(ec["action_range"])).deterministic_sample()
            a = np.concatenate([keys.numpy().astype(np.float64), mouse.numpy().astype(np.float64)], axis=1)
            obs, r, d, _ = env.vector_step(a)
            total += float(r[0]); done = bool(d[0]); ticks += 1
    assert ticks == 720 and 5600.0 < total < 5900.0, (ticks, total)


# ---- the pin TensorFlow's absence left open: hand-derived known answers (oracle/gen_dist_known_answers.py) --------------------------
def _known():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dist_known_answers.json")))


def test_known_answers_fixture_is_what_its_generator_writes(tmp_path):
    """The fixture is reproducible from its generator (pure `math`, no repository imports) - nobody edits expected values by hand."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "oracle", "gen_dist_known_answers.py")).read()
    assert "import numpy" not in src and "from oracle" not in src and "q1physrl_amd" not in src.split('"""')[2]
    patched = tmp_path / "gen.py"
    patched.write_text(src.replace('os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dist_known_answers.json")',
                                   repr(str(tmp_path / "out.json"))))
    subprocess.run([sys.executable, str(patched)], check=True, capture_output=True)
    import json
    assert json.load(open(tmp_path / "out.json")) == _known()


def test_oracle_and_torch_distribution_reproduce_the_hand_derived_known_answers():
    """oracle/dist_oracle.py (the checker of the HIP kernels) and q1physrl_amd/policy.py (the learner's distribution) against closed
    forms of action_dist.py:91-96, 153-196 evaluated by hand at points where they collapse to ln 2, ln 3, ln 20, Phi(1), Phi(2):
    with TensorFlow absent, THIS is what pins the restatements to the reference's formulas.  float64: 1e-12 (1e-9 through ndtri)."""
    k = _known()
    low, high = k["low"], k["high"]
    f64 = lambda v: torch.tensor([v], dtype=torch.float64)          # noqa: E731
    seen = set()
    for c in k["cases"]:
        kind = c["kind"]
        seen.add(kind)
        if kind == "kl":
            (m, ls), (m2, ls2) = c["self_in"], c["other_in"]
            assert abs(float(DO.mouse_kl(m, ls, m2, ls2)) - c["expect"]) < 1e-12, c["name"]
            d1, d2 = P.GaussianSquashedGaussian(f64([m, ls]), low, high), P.GaussianSquashedGaussian(f64([m2, ls2]), low, high)
            assert abs(float(d1.kl(d2)) - c["expect"]) < 1e-12, c["name"]
        elif kind == "entropy":
            m, ls = c["self_in"]
            assert abs(float(DO.mouse_entropy(m, ls, low, high)) - c["expect"]) < 1e-12, c["name"]
            assert abs(float(P.GaussianSquashedGaussian(f64([m, ls]), low, high).entropy()) - c["expect"]) < 1e-12, c["name"]
        elif kind == "logp":
            m, ls = c["self_in"]
            assert abs(float(DO.mouse_logp(np.array([c["x"]]), m, ls, low, high)[0]) - c["expect"]) < 1e-9, c["name"]
            assert abs(float(P.GaussianSquashedGaussian(f64([m, ls]), low, high).logp(f64([c["x"]]))) - c["expect"]) < 1e-9, c["name"]
        elif kind == "squash":
            assert abs(float(DO.squash(c["raw"], low, high)) - c["expect"]) < 1e-12, c["name"]
            d0 = P.GaussianSquashedGaussian(f64([0.0, 0.0]), low, high)
            assert abs(float(d0._squash(f64([c["raw"]]))) - c["expect"]) < 1e-12, c["name"]
        elif kind == "deterministic":
            m, ls = c["self_in"]
            assert abs(float(P.GaussianSquashedGaussian(f64([m, ls]), low, high).deterministic_sample()) - c["expect"]) < 1e-12, c["name"]
            assert abs(float(DO.squash(DO.clip_params(m, ls)[0], low, high)) - c["expect"]) < 1e-12, c["name"]
        elif kind == "key":
            l0, l1 = c["logits"]
            lp0, lp1 = DO.key_logprobs(np.array([l0]), np.array([l1]))
            assert abs(lp0[0] - c["expect_logp"][0]) < 1e-12 and abs(lp1[0] - c["expect_logp"][1]) < 1e-12, c["name"]
            cat = P.Categorical2(f64([l0, l1]))
            assert abs(float(cat.logp(torch.tensor([0]))) - c["expect_logp"][0]) < 1e-12 and abs(float(cat.logp(torch.tensor([1]))) - c["expect_logp"][1]) < 1e-12
            assert abs(float(cat.entropy()) - c["expect_entropy"]) < 1e-12, c["name"]
            _, ent, _ = DO.categorical_terms(np.array([[l0, l1]]))
            assert abs(ent[0] - c["expect_entropy"]) < 1e-12
        elif kind == "key_kl":
            a, b = P.Categorical2(f64(c["logits"])), P.Categorical2(f64(c["other"]))
            assert abs(float(a.kl(b)) - c["expect"]) < 1e-12 and abs(float(b.kl(a)) - c["expect_reverse"]) < 1e-12, c["name"]
            _, _, klv = DO.categorical_terms(np.array([c["other"]]), np.array([c["logits"]]))          # KL(old = logits || new = other)
            assert abs(klv[0] - c["expect"]) < 1e-12
        elif kind == "tuple":
            dist = P.Q1PhysActionDist(f64(c["row"]), high, num_keys=4)
            assert abs(float(dist.logp(torch.tensor([c["keys"]]), f64([c["x"]]))) - c["expect_logp"]) < 1e-9, c["name"]
            assert abs(float(dist.entropy()) - c["expect_entropy"]) < 1e-12, c["name"]
        elif kind == "tuple_deterministic":
            dist = P.Q1PhysActionDist(f64(c["row"]), high, num_keys=4)
            kk, mm = dist.deterministic_sample()
            assert kk[0].tolist() == c["expect_keys"] and abs(float(mm) - c["expect_x"]) < 1e-12
            assert abs(float(dist.logp(kk, mm)) - c["expect_logp"]) < 1e-9, c["name"]
    assert seen == {"kl", "entropy", "logp", "squash", "deterministic", "key", "key_kl", "tuple", "tuple_deterministic"}
    # the minibatch rows (sums over the five children): the torch distribution and the loss oracle
    from oracle import ppo_oracle as PO
    rows = k["batch"]
    new = torch.tensor([r["row"] for r in rows], dtype=torch.float64)
    old = torch.tensor([r["old_row"] for r in rows], dtype=torch.float64)
    keys = torch.tensor([r["keys"] for r in rows])
    x = torch.tensor([[r["x"]] for r in rows], dtype=torch.float64)
    dn, do = P.Q1PhysActionDist(new, high, num_keys=4), P.Q1PhysActionDist(old, high, num_keys=4)
    assert np.allclose(dn.logp(keys, x).numpy(), [r["expect_logp"] for r in rows], rtol=0, atol=1e-9)
    assert np.allclose(dn.entropy().numpy(), [r["expect_entropy"] for r in rows], rtol=0, atol=1e-11)
    assert np.allclose(do.kl(dn).numpy(), [r["expect_kl_old_new"] for r in rows], rtol=1e-12, atol=1e-11)
    assert hasattr(PO, "ppo_loss_grad")
