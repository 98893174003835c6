"""Hand-derived KNOWN ANSWERS for the reference's action distribution (q1physrl/action_dist.py:46-243) - the pin of
oracle/dist_oracle.py, q1physrl_amd/policy.py and the policy-side HIP kernels that TensorFlow's absence leaves open.

TEST INFRASTRUCTURE.  This script imports NOTHING from this repository and nothing numerical but Python's `math`: every expected
value below is the closed form of action_dist.py evaluated BY HAND at parameter points chosen so that the expression collapses to a
few named constants.  Run it to (re)write tests/golden/dist_known_answers.json:

    python oracle/gen_dist_known_answers.py

Notation (action_dist.py line numbers): S = _SCALE = 0.5 * 1.8137 (:151); the distribution of the raw value is N(m, s) with
m = clip(mean, -3, 3) (:72), log s = clip(log_std, -20, 2) (:68-69, RLlib's MIN/MAX_LOG_NN_OUTPUT); R = high - low;
Phi = standard normal CDF; squash(u) = clip(Phi(u / S), 1e-6, 1 - 1e-6) * R + low (:186-192); unsquash(x) = S * Phi^-1((x - low) / R)
(:194-196); logp(x) = log N(m, s)(u) - [log N(0, S)(u) + log R], u = unsquash(x) (:91-96, :180-184);
kl(self || other) = log s_o - log s + (s^2 + (m - m_o)^2) / (2 s_o^2) - 1/2 (:153-165);
entropy = log R - (log S - log s + (s^2 + m^2) / (2 S^2) - 1/2) (:167-178).
Keys: each Discrete(2) child is RLlib's Categorical over two logits (action_dist.py:221-222 -> ModelCatalog): p1 = softmax(l)[1].
The tuple distribution sums its children (MultiActionDistribution).
"""
import json
import math
import os

S = 0.5 * 1.8137
LN2, LN3, LN20 = math.log(2.0), math.log(3.0), math.log(20.0)
PHI_1 = 0.8413447460685429      # Phi(1), standard tables (A&S 26.2): 0.84134 47460 68543
PHI_2 = 0.9772498680518208      # Phi(2)
LOW, HIGH = -10.0, 10.0         # R = 20: params.yml's action_range = 10 (data/params.yml:17)
R = HIGH - LOW

cases = []


def case(name, derivation, **kw):
    cases.append({"name": name, "derivation": " ".join(derivation.split()), **kw})


# ---------------------------------------------------------------------------------------------------- KL (action_dist.py:153-165)
case("kl_identical", "self = other: log s_o - log s = 0, (s^2 + 0) / (2 s^2) = 1/2, minus 1/2 -> 0 exactly.",
     kind="kl", self_in=[0.7, -0.4], other_in=[0.7, -0.4], expect=0.0)
case("kl_unit_shift", "m = 1, m_o = 0, s = s_o = 1 (log_std 0): 0 + (1 + 1) / 2 - 1/2 = 1/2.",
     kind="kl", self_in=[1.0, 0.0], other_in=[0.0, 0.0], expect=0.5)
case("kl_wider_other", "m = m_o, s = 1, s_o = 2 (log_std ln 2): ln 2 + 1 / (2 * 4) - 1/2 = ln 2 - 3/8.",
     kind="kl", self_in=[-0.5, 0.0], other_in=[-0.5, LN2], expect=LN2 - 0.375)
case("kl_wider_self", "m = -2, s = 3 (log_std ln 3); m_o = 1, s_o = 1: -ln 3 + (9 + 9) / 2 - 1/2 = 17/2 - ln 3.",
     kind="kl", self_in=[-2.0, LN3], other_in=[1.0, 0.0], expect=8.5 - LN3)
case("kl_mean_clip", "means 5 and -7 are clipped to 3 and -3 (:72); s = s_o = 1: (1 + 36) / 2 - 1/2 = 18.",
     kind="kl", self_in=[5.0, 0.0], other_in=[-7.0, 0.0], expect=18.0)
case("kl_log_std_clip", "log_std 4 -> 2 and 3 -> 2 (MAX_LOG_NN_OUTPUT), equal means: the clipped distributions coincide -> 0; "
     "unclipped it would be 3 - 4 + e^8 / (2 e^6) - 1/2 = e^2 / 2 - 3/2 = 2.19.",
     kind="kl", self_in=[0.3, 4.0], other_in=[0.3, 3.0], expect=0.0)

# ---------------------------------------------------------------------------------------------------- entropy (:167-178)
case("entropy_uniform", "m = 0, s = S: squashing N(0, S) with its own CDF gives the uniform law on (low, high): "
     "log R - (log S - log S + (S^2 + 0) / (2 S^2) - 1/2) = log R = ln 20.",
     kind="entropy", self_in=[0.0, math.log(S)], expect=LN20)
case("entropy_shifted", "m = S, s = S: log R - ((S^2 + S^2) / (2 S^2) - 1/2) = ln 20 - 1/2.",
     kind="entropy", self_in=[S, math.log(S)], expect=LN20 - 0.5)
case("entropy_narrow", "m = 0, s = S / 2 (log_std = log S - ln 2): log R - (ln 2 + (S^2 / 4) / (2 S^2) - 1/2) = ln 20 - ln 2 + 3/8 = ln 10 + 3/8.",
     kind="entropy", self_in=[0.0, math.log(S) - LN2], expect=math.log(10.0) + 0.375)
case("entropy_mean_clip", "mean 9 is clipped to 3, s = S: ln 20 - ((S^2 + 9) / (2 S^2) - 1/2) = ln 20 - 9 / (2 S^2).",
     kind="entropy", self_in=[9.0, math.log(S)], expect=LN20 - 9.0 / (2.0 * S * S))

# ---------------------------------------------------------------------------------------------------- logp (:91-96, :180-184)
case("logp_uniform_mid", "m = 0, s = S, x = midpoint: u = S Phi^-1(1/2) = 0; the two Gaussian terms cancel for every x: logp = -log R = -ln 20.",
     kind="logp", self_in=[0.0, math.log(S)], x=0.0, expect=-LN20)
case("logp_uniform_phi1", "same law at x = low + R Phi(1): u = S, still -ln 20 (the density of a uniform law is flat).",
     kind="logp", self_in=[0.0, math.log(S)], x=LOW + R * PHI_1, expect=-LN20)
case("logp_mid_general", "x = midpoint -> u = 0: log N(m, s)(0) - log N(0, S)(0) - log R = -m^2 / (2 s^2) - log s + log S - log R; "
     "m = 1, s = 1/2: -2 + ln 2 + log S - ln 20.",
     kind="logp", self_in=[1.0, -LN2], x=0.0, expect=-2.0 + LN2 + math.log(S) - LN20)
case("logp_at_mode", "m = S, s = S / 2, x = low + R Phi(1) -> u = S = m: log N(m, s)(m) = -log(S / 2) - c, log N(0, S)(S) = -1/2 - log S - c "
     "(c = ln sqrt(2 pi)): logp = ln 2 + 1/2 - ln 20.",
     kind="logp", self_in=[S, math.log(S) - LN2], x=LOW + R * PHI_1, expect=LN2 + 0.5 - LN20)
case("logp_two_sigma", "m = 0, s = 2 S, x = low + R Phi(2) -> u = 2 S: log N(0, 2S)(2S) = -1/2 - log(2S) - c, log N(0, S)(2S) = -2 - log S - c: "
     "logp = 3/2 - ln 2 - ln 20.",
     kind="logp", self_in=[0.0, math.log(S) + LN2], x=LOW + R * PHI_2, expect=1.5 - LN2 - LN20)

# ---------------------------------------------------------------------------------------------------- squash / deterministic sample
case("squash_zero", "Phi(0) = 1/2: the midpoint.", kind="squash", raw=0.0, expect=0.0)
case("squash_one_scale", "raw = S: Phi(1) R + low.", kind="squash", raw=S, expect=LOW + R * PHI_1)
case("squash_minus_two_scales", "raw = -2 S: (1 - Phi(2)) R + low.", kind="squash", raw=-2.0 * S, expect=LOW + R * (1.0 - PHI_2))
case("squash_clip_high", "raw = 100: Phi = 1 to double precision, clipped to 1 - 1e-6 (:190-191): high - 1e-6 R.",
     kind="squash", raw=100.0, expect=HIGH - 1e-6 * R)
case("squash_clip_low", "raw = -100: clipped to 1e-6: low + 1e-6 R.", kind="squash", raw=-100.0, expect=LOW + 1e-6 * R)
case("deterministic_is_squashed_clipped_mean", "deterministic_sample = squash(distr.mean()) (:84-89) and the mean is the CLIPPED one: "
     "mean 7 -> 3; Phi(3 / S) R + low with 3 / S = 3.308154...; Phi(3.308154) = 0.99953045 (erfc) - evaluated with math.erfc.",
     kind="deterministic", self_in=[7.0, 0.1], expect=LOW + R * (1.0 - 0.5 * math.erfc((3.0 / S) / math.sqrt(2.0))))

# ---------------------------------------------------------------------------------------------------- keys: Categorical over 2 logits
case("key_even", "logits (0, 0): p = (1/2, 1/2): logp = -ln 2 for either action, entropy ln 2.",
     kind="key", logits=[0.0, 0.0], expect_logp=[-LN2, -LN2], expect_entropy=LN2)
case("key_three_to_one", "logits (0, ln 3): p1 = 3/4: logp(1) = ln 3 - 2 ln 2, logp(0) = -2 ln 2; entropy = 2 ln 2 - (3/4) ln 3.",
     kind="key", logits=[0.0, LN3], expect_logp=[-2.0 * LN2, LN3 - 2.0 * LN2], expect_entropy=2.0 * LN2 - 0.75 * LN3)
case("key_shift_invariance", "logits (5, 5 + ln 3): a common shift changes nothing.",
     kind="key", logits=[5.0, 5.0 + LN3], expect_logp=[-2.0 * LN2, LN3 - 2.0 * LN2], expect_entropy=2.0 * LN2 - 0.75 * LN3)
case("key_kl", "KL((1/4, 3/4) || (1/2, 1/2)) = (3/4) ln(3/2) + (1/4) ln(1/2) = (3/4) ln 3 - ln 2; "
     "the reverse KL((1/2, 1/2) || (1/4, 3/4)) = (1/2) ln 2 + (1/2) ln(2/3) = ln 2 - (1/2) ln 3.",
     kind="key_kl", logits=[0.0, LN3], other=[0.0, 0.0], expect=0.75 * LN3 - LN2, expect_reverse=LN2 - 0.5 * LN3)

# ---------------------------------------------------------------------------------------------------- the tuple distribution (sum of children)
case("tuple_sum", "4 keys with logits (0, ln 3) and actions (1, 0, 1, 1) + the mouse child of 'logp_at_mode': "
     "logp = 3 (ln 3 - 2 ln 2) + (-2 ln 2) + ln 2 + 1/2 - ln 20; entropy = 4 (2 ln 2 - (3/4) ln 3) + entropy of (m = S, s = S/2) "
     "= ln 20 - (ln 2 + (S^2/4 + S^2) / (2 S^2) - 1/2) = ln 20 - ln 2 - 1/8.",
     kind="tuple", row=[0.0, LN3, 0.0, LN3, 0.0, LN3, 0.0, LN3, S, math.log(S) - LN2], keys=[1, 0, 1, 1], x=LOW + R * PHI_1,
     expect_logp=3.0 * (LN3 - 2.0 * LN2) - 2.0 * LN2 + LN2 + 0.5 - LN20,
     expect_entropy=4.0 * (2.0 * LN2 - 0.75 * LN3) + LN20 - LN2 - 0.125)

case("tuple_deterministic", "deterministic_sample of the row above: every key's arg-max is 1 (ln 3 > 0), the mouse is squash(m) = low + R Phi(1); "
     "its log-probability: 4 (ln 3 - 2 ln 2) + ln 2 + 1/2 - ln 20.",
     kind="tuple_deterministic", row=[0.0, LN3, 0.0, LN3, 0.0, LN3, 0.0, LN3, S, math.log(S) - LN2], expect_keys=[1, 1, 1, 1],
     expect_x=LOW + R * PHI_1, expect_logp=4.0 * (LN3 - 2.0 * LN2) + LN2 + 0.5 - LN20)


# ---------------------------------------------------------------------------------------------------- a batch for the loss kernels
# Rows of a minibatch for q1env_ppo_loss_grad / ppo.ppo_loss: (new row, old row, action) -> logp(new), entropy(new), KL(old || new), each
# the SUM over the five children, evaluated here from the closed forms with `math` only.  The action's mouse value is given through
# z = u / S (x = low + R Phi(z), Phi by math.erfc), so that u is known exactly: logp_mouse = -(S z - m)^2 / (2 s^2) - log s + z^2 / 2 + log S - log R.
def phi(z):
    return 0.5 * math.erfc(-z / math.sqrt(2.0))


def clip(v, lo, hi):
    return min(max(v, lo), hi)


def key_terms(l0, l1, a, o0, o1):
    d, do = l1 - l0, o1 - o0
    p1, q1 = 1.0 / (1.0 + math.exp(-d)), 1.0 / (1.0 + math.exp(-do))
    lp1, lp0 = -math.log1p(math.exp(-d)), -math.log1p(math.exp(d))
    lq1, lq0 = -math.log1p(math.exp(-do)), -math.log1p(math.exp(do))
    return (lp1 if a else lp0), -(p1 * lp1 + (1 - p1) * lp0), q1 * (lq1 - lp1) + (1 - q1) * (lq0 - lp0)


def mouse_terms(mean, log_std, z, o_mean, o_log_std):
    m, ls = clip(mean, -3.0, 3.0), clip(log_std, -20.0, 2.0)
    mo, lso = clip(o_mean, -3.0, 3.0), clip(o_log_std, -20.0, 2.0)
    s, so = math.exp(ls), math.exp(lso)
    logp = -(S * z - m) ** 2 / (2 * s * s) - ls + z * z / 2 + math.log(S) - math.log(R)
    ent = math.log(R) - (math.log(S) - ls + (s * s + m * m) / (2 * S * S) - 0.5)
    kl = ls - lso + (so * so + (mo - m) ** 2) / (2 * s * s) - 0.5                  # KL(old || new): self = old, other = new
    return logp, ent, kl


batch = []
specs = [   # (new key logit pairs, new mean, new log_std, old key logit pairs, old mean, old log_std, key actions, z)
    ([(0, LN3)] * 4, S, math.log(S) - LN2, [(0, LN3)] * 4, S, math.log(S) - LN2, [1, 0, 1, 1], 1.0),
    ([(0, 0), (0, LN3), (LN3, 0), (1.0, -1.0)], 0.0, math.log(S), [(0, 0)] * 4, 0.0, math.log(S), [0, 1, 1, 0], 0.0),
    ([(0.5, -0.25), (2.0, 2.0), (-1.0, 0.5), (0.0, 3.0)], 1.0, -LN2, [(0.0, 0.0), (1.0, 2.0), (-1.0, 0.5), (0.5, 2.0)], 0.5, 0.0, [1, 1, 0, 1], -0.5),
    ([(0, LN3)] * 4, -2.0, LN3, [(0, 0)] * 4, 1.0, 0.0, [0, 0, 0, 0], 2.0),
    ([(0.3, 0.1), (0.1, 0.3), (0.0, 0.0), (-2.0, 2.0)], 5.0, 0.0, [(0.3, 0.1), (0.1, 0.3), (0.0, 0.0), (-2.0, 2.0)], -7.0, 0.0, [1, 0, 1, 1], 1.5),
    ([(1.0, 0.0)] * 4, 0.3, 4.0, [(0.0, 1.0)] * 4, 0.3, 3.0, [1, 1, 0, 0], -1.25),
    ([(0, LN3), (LN3, 0), (0, 0), (0, LN3)], 0.0, math.log(S) + LN2, [(0, LN3), (LN3, 0), (0, 0), (0, LN3)], 0.25, math.log(S), [1, 0, 0, 1], 2.0),
    ([(0.0, 0.0)] * 4, -0.75, -1.5, [(0.2, -0.2)] * 4, -0.5, -1.0, [0, 1, 0, 1], -0.75 / S),
]
for new_keys, mean, ls, old_keys, omean, ols, acts, z in specs:
    logp = ent = kl = 0.0
    for (l0, l1), (o0, o1), a in zip(new_keys, old_keys, acts):
        a_lp, a_h, a_kl = key_terms(l0, l1, a, o0, o1)
        logp += a_lp; ent += a_h; kl += a_kl
    m_lp, m_h, m_kl = mouse_terms(mean, ls, z, omean, ols)
    batch.append({"row": [v for pr in new_keys for v in pr] + [mean, ls], "old_row": [v for pr in old_keys for v in pr] + [omean, ols],
                  "keys": acts, "z": z, "x": LOW + R * phi(z), "expect_logp": logp + m_lp, "expect_entropy": ent + m_h, "expect_kl_old_new": kl + m_kl})

out = {"low": LOW, "high": HIGH, "scale_S": S, "batch": batch, "constants": {"Phi(1)": PHI_1, "Phi(2)": PHI_2},
       "generated_by": "oracle/gen_dist_known_answers.py (closed forms of q1physrl/action_dist.py evaluated by hand; no TensorFlow, no repository code)",
       "cases": cases}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dist_known_answers.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(f"wrote {len(cases)} cases to {path}")
