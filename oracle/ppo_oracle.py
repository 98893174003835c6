"""NumPy ORACLE for the learner-side glue (TEST INFRASTRUCTURE): generalised advantage estimation and the PPO loss of
RLlib 0.8.4 as the reference configures it (q1physrl/train.py:60-64, data/params.yml:4-13).  PARITY UNPINNED by the
reference (Ray/TF are not installable here); pinned to the published formulas and to dist_oracle.py."""
import numpy as np

from . import dist_oracle as DO


def gae(reward, value, done, gamma, lam):
    """reward (T,N), value (T+1,N), done (T,N) -> adv, vtarg (T,N), float32 arithmetic in the kernel's order."""
    t, n = reward.shape
    adv = np.zeros((t, n), np.float32)
    a = np.zeros(n, np.float32)
    v_next = value[t].astype(np.float32)
    g, l = np.float32(gamma), np.float32(lam)
    for i in range(t - 1, -1, -1):
        nd = np.where(done[i] != 0, np.float32(0), np.float32(1))
        v = value[i].astype(np.float32)
        delta = reward[i] + g * v_next * nd - v
        a = delta + g * l * nd * a
        adv[i] = a
        v_next = v
    return adv, adv + value[:t].astype(np.float32)


def dist_terms(logits, keys, mouse, action_range):
    lp = DO.mouse_logp(mouse[:, 0], logits[:, 8], logits[:, 9], -action_range, action_range)
    ent = DO.mouse_entropy(logits[:, 8], logits[:, 9], -action_range, action_range)
    for k in range(4):
        lp0, lp1 = DO.key_logprobs(logits[:, 2 * k], logits[:, 2 * k + 1])
        lp = lp + np.where(keys[:, k] == 1, lp1, lp0)
        ent = ent - (np.exp(lp0) * lp0 + np.exp(lp1) * lp1)
    return lp, ent


def kl_terms(old, new):
    kl = DO.mouse_kl(old[:, 8], old[:, 9], new[:, 8], new[:, 9])
    for k in range(4):
        a0, a1 = DO.key_logprobs(old[:, 2 * k], old[:, 2 * k + 1])
        b0, b1 = DO.key_logprobs(new[:, 2 * k], new[:, 2 * k + 1])
        kl = kl + np.exp(a0) * (a0 - b0) + np.exp(a1) * (a1 - b1)
    return kl


def ppo_loss(new_logits, new_value, b, action_range, clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff, kl_coeff):
    logp, ent = dist_terms(new_logits, b["keys"], b["mouse"], action_range)
    ratio = np.exp(logp - b["logp"])
    sur = np.minimum(b["adv"] * ratio, b["adv"] * np.clip(ratio, 1 - clip_param, 1 + clip_param))
    vc = b["value"] + np.clip(new_value - b["value"], -vf_clip_param, vf_clip_param)
    vf = np.maximum((new_value - b["vtarg"]) ** 2, (vc - b["vtarg"]) ** 2)
    kl = kl_terms(b["old_logits"], new_logits)
    return float(np.mean(-sur + kl_coeff * kl + vf_loss_coeff * vf - entropy_coeff * ent)), float(kl.mean()), float(ent.mean())
