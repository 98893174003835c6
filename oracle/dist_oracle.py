"""NumPy / SciPy ORACLE for the policy-side action distribution (reference q1physrl/action_dist.py:46-243).
TEST INFRASTRUCTURE.  float64 restatement of the formulas with scipy.special.ndtr / ndtri.

PIN.  action_dist.py is TensorFlow 2.1 + tensorflow-probability 0.9 + RLlib 0.8.4 code, none of them installable here (no network),
and the reference's own tests do not cover it, so no golden vector can be produced BY RUNNING it.  The pin is instead
tests/golden/dist_known_answers.json: the closed forms of action_dist.py:91-96 (logp), :153-165 (kl), :167-178 (entropy), :186-196
(squash / unsquash) and of the key Categoricals, evaluated BY HAND at parameter points where they collapse to ln 2, ln 3, ln 20, Phi(1),
Phi(2) - written by oracle/gen_dist_known_answers.py, which imports nothing from this repository and nothing numerical but `math`.
This module, q1physrl_amd/policy.py and the HIP kernels (q1env_policy_sample, q1env_ppo_loss_grad) are all checked against it
(tests/test_policy_dist.py, tests/test_hip_policy.py), next to the first-principles checks (density integrates to 1, closed-form entropy
= numerical differential entropy, finite-difference Jacobian) and the published WR checkpoint scoring ~5700.
"""
import numpy as np
from scipy import special

SMALL_NUMBER, MIN_LOG, MAX_LOG = 1e-6, -20.0, 2.0
S = 0.5 * 1.8137
HALF_LOG_2PI = 0.5 * np.log(2 * np.pi)


def clip_params(mean, log_std):
    return np.clip(mean, -3, 3), np.clip(log_std, MIN_LOG, MAX_LOG)


def squash(raw, low, high):
    return np.clip(special.ndtr(raw / S), SMALL_NUMBER, 1 - SMALL_NUMBER) * (high - low) + low


def unsquash(x, low, high):
    return S * special.ndtri((x - low) / (high - low))


def normal_logpdf(x, mean, log_std):
    z = (x - mean) * np.exp(-log_std)
    return -0.5 * z * z - log_std - HALF_LOG_2PI


def mouse_logp(x, mean, log_std, low, high):
    mean, log_std = clip_params(mean, log_std)
    raw = unsquash(x, low, high)
    return normal_logpdf(raw, mean, log_std) - (normal_logpdf(raw, 0.0, np.log(S)) + np.log(high - low))


def mouse_entropy(mean, log_std, low, high):
    mean, log_std = clip_params(mean, log_std)
    std = np.exp(log_std)
    return np.log(high - low) - (np.log(S) - log_std + (std ** 2 + mean ** 2) / (2 * S ** 2) - 0.5)


def mouse_kl(mean, log_std, mean2, log_std2):
    mean, log_std = clip_params(mean, log_std)
    mean2, log_std2 = clip_params(mean2, log_std2)
    return log_std2 - log_std + (np.exp(log_std) ** 2 + (mean - mean2) ** 2) / (2 * np.exp(log_std2) ** 2) - 0.5


def key_logprobs(l0, l1):
    m = np.maximum(l0, l1)
    lse = m + np.log(np.exp(l0 - m) + np.exp(l1 - m))
    return l0 - lse, l1 - lse


def categorical_lse(lg):
    m = lg.max(axis=1)
    return m + np.log(np.exp(lg - m[:, None]).sum(axis=1))


def categorical_terms(lg, lg_old=None):
    """log-probabilities, entropy and (with lg_old) KL(old || new) of RLlib's Categorical over the rows of lg."""
    lp = lg - categorical_lse(lg)[:, None]
    ent = -(np.exp(lp) * lp).sum(axis=1)
    if lg_old is None:
        return lp, ent, None
    lpo = lg_old - categorical_lse(lg_old)[:, None]
    return lp, ent, (np.exp(lpo) * (lpo - lp)).sum(axis=1)


def sample_from_philox(cfg, logits, seed, genv, counter, deterministic=False):
    """What q1env_policy_sample produces for these logits (float64 restatement of policy_sample_kernel)."""
    from . import philox as PH
    k = cfg.num_keys
    a = PH.draw(seed, genv, counter, 3, 0)
    b = PH.draw(seed, genv, counter, 3, 1)
    ku = [a[0], a[1], b[0], b[1]]
    n = logits.shape[0]
    keys = np.zeros(n, dtype=np.uint8)
    logp = np.zeros(n)
    margin = np.full(n, np.inf)                   # |u - p1|: how close a key draw is to its threshold
    for j in range(k):
        l0, l1 = logits[:, 2 * j].astype(np.float64), logits[:, 2 * j + 1].astype(np.float64)
        p1 = 1.0 / (1.0 + np.exp(-(l1 - l0)))
        u = (ku[j] >> np.uint64(8)).astype(np.float64) / 16777216.0
        bit = (l1 > l0) if deterministic else (u < p1)
        margin = np.minimum(margin, np.abs(u - p1))
        keys |= bit.astype(np.uint8) << j
        lp0, lp1 = key_logprobs(l0, l1)
        logp += np.where(bit, lp1, lp0)
    if not cfg.allow_yaw:
        return keys, np.zeros(n), logp, margin
    if cfg.discrete_yaw_steps != -1:              # Categorical over 2S+1 steps: inverse CDF on the uniform of word a[2]
        m = 2 * cfg.discrete_yaw_steps + 1
        lg = logits[:, 2 * k:2 * k + m].astype(np.float64)
        lse = categorical_lse(lg)
        prob = np.exp(lg - lse[:, None])
        cdf = np.cumsum(prob, axis=1)
        if deterministic:
            choice = np.argmax(lg, axis=1)
        else:
            u = (a[2] >> np.uint64(8)).astype(np.float64) / 16777216.0
            choice = np.minimum((cdf <= u[:, None]).sum(axis=1), m - 1)
            edges = np.concatenate([np.zeros((n, 1)), cdf[:, :-1]], axis=1)
            margin = np.minimum(margin, np.min(np.abs(edges[:, 1:] - u[:, None]), axis=1))
        logp += lg[np.arange(n), choice] - lse
        return keys, choice.astype(np.float64), logp, margin
    low, high = -float(np.float32(cfg.action_range)), float(np.float32(cfg.action_range))
    mean, log_std = clip_params(logits[:, 2 * k].astype(np.float64), logits[:, 2 * k + 1].astype(np.float64))
    if deterministic:
        eps = 0.0
    else:
        u1 = ((a[2] >> np.uint64(8)).astype(np.float64) + 1.0) / 16777216.0
        u2 = (a[3] >> np.uint64(8)).astype(np.float64) / 16777216.0
        eps = np.sqrt(-2 * np.log(u1)) * np.cos(2 * np.pi * u2)
    raw = mean + np.exp(log_std) * eps
    mouse = squash(raw, low, high)
    logp += mouse_logp(mouse, logits[:, 2 * k].astype(np.float64), logits[:, 2 * k + 1].astype(np.float64), low, high)
    return keys, mouse, logp, margin
