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


def ppo_loss_grad_discrete(new_logits, new_value, b, steps, clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff, kl_coeff,
                           num_keys=4):
    """ppo_loss_grad for a DISCRETE mouse (Config.discrete_yaw_steps = steps): the row is num_keys x (logit0, logit1) then the
    2*steps+1 logits of RLlib's Categorical; b["mouse"] holds the chosen step index.  Same closed forms as the kernel's
    yaw_mode == 2 branch: d logp / d l_j = [j == a] - p_j, d H / d l_j = -p_j (log p_j + H), d KL / d l_j = p_j - po_j."""
    L, O = new_logits.astype(np.float64), b["old_logits"].astype(np.float64)
    n, k2, m = L.shape[0], 2 * num_keys, 2 * steps + 1
    sig = lambda d: 1.0 / (1.0 + np.exp(-d))
    softplus = lambda z: np.maximum(z, 0.0) + np.log1p(np.exp(-np.abs(z)))
    logp, ent, kl = np.zeros(n), np.zeros(n), np.zeros(n)
    dlp, dh, dk = np.zeros_like(L), np.zeros_like(L), np.zeros_like(L)
    for k in range(num_keys):
        d, d_o = L[:, 2 * k + 1] - L[:, 2 * k], O[:, 2 * k + 1] - O[:, 2 * k]
        pn, po = sig(d), sig(d_o)
        a = b["keys"][:, k].astype(np.float64)
        logp -= np.where(a != 0, softplus(-d), softplus(d))
        ent += softplus(d) - d * pn
        kl += po * (softplus(-d) - softplus(-d_o)) + (1 - po) * (softplus(d) - softplus(d_o))
        for arr, g in ((dlp, a - pn), (dh, -d * pn * (1 - pn)), (dk, pn - po)):
            arr[:, 2 * k + 1], arr[:, 2 * k] = g, -g
    lp, h_c, kl_c = DO.categorical_terms(L[:, k2:k2 + m], O[:, k2:k2 + m])
    lpo = O[:, k2:k2 + m] - DO.categorical_lse(O[:, k2:k2 + m])[:, None]
    act = b["mouse"].reshape(-1).astype(np.int64)
    logp += lp[np.arange(n), act]
    ent += h_c
    kl += kl_c
    onehot = np.zeros((n, m))
    onehot[np.arange(n), act] = 1.0
    dlp[:, k2:k2 + m] = onehot - np.exp(lp)
    dh[:, k2:k2 + m] = -np.exp(lp) * (lp + h_c[:, None])
    dk[:, k2:k2 + m] = np.exp(lp) - np.exp(lpo)
    ratio = np.exp(logp - b["logp"])
    s1, s2 = b["adv"] * ratio, b["adv"] * np.clip(ratio, 1 - clip_param, 1 + clip_param)
    sur = np.minimum(s1, s2)
    c_lp = np.where(s1 <= s2, -s1, 0.0)
    v, vo, vt = new_value.astype(np.float64), b["value"].astype(np.float64), b["vtarg"].astype(np.float64)
    dv = v - vo
    vc = vo + np.clip(dv, -vf_clip_param, vf_clip_param)
    e1, e2 = (v - vt) ** 2, (vc - vt) ** 2
    vf = np.maximum(e1, e2)
    dvf = np.where(e1 >= e2, 2 * (v - vt), np.where(np.abs(dv) <= vf_clip_param, 2 * (vc - vt), 0.0))
    dlogits = (c_lp[:, None] * dlp + kl_coeff * dk - entropy_coeff * dh) / n
    total = -sur + kl_coeff * kl + vf_loss_coeff * vf - entropy_coeff * ent
    stats = {"entropy": ent.mean(), "kl": kl.mean(), "policy_loss": (-sur).mean(), "total_loss": total.mean(), "vf_loss": vf.mean(),
             "logp": logp}
    return dlogits, vf_loss_coeff * dvf / n, stats


def ppo_loss_grad(new_logits, new_value, b, action_range, clip_param, vf_clip_param, vf_loss_coeff, entropy_coeff, kl_coeff):
    """float64 restatement of q1env_ppo_loss_grad (q1physrl_amd/csrc/q1env_policy.hip::ppo_loss_grad_kernel): the closed-form derivatives
    of ppo_loss above with respect to (new_logits, new_value), and the five statistics.  b["keys"] is (B, 4) 0/1.
    Returns dlogits (B, 10), dvalue (B,), stats dict (means of entropy, kl, policy_loss, total_loss, vf_loss)."""
    L, O = new_logits.astype(np.float64), b["old_logits"].astype(np.float64)
    n = L.shape[0]
    low, high = -float(action_range), float(action_range)
    sig = lambda d: 1.0 / (1.0 + np.exp(-d))
    softplus = lambda z: np.maximum(z, 0.0) + np.log1p(np.exp(-np.abs(z)))
    logp, ent, kl = np.zeros(n), np.zeros(n), np.zeros(n)
    dlp, dh, dk = np.zeros((n, 10)), np.zeros((n, 10)), np.zeros((n, 10))        # d logp / d logits, d entropy / ., d kl / .
    for k in range(4):
        d, d_o = L[:, 2 * k + 1] - L[:, 2 * k], O[:, 2 * k + 1] - O[:, 2 * k]
        pn, po = sig(d), sig(d_o)
        a = b["keys"][:, k].astype(np.float64)
        logp -= np.where(a != 0, softplus(-d), softplus(d))
        ent += softplus(d) - d * pn
        kl += po * (softplus(-d) - softplus(-d_o)) + (1 - po) * (softplus(d) - softplus(d_o))
        for arr, g in ((dlp, a - pn), (dh, -d * pn * (1 - pn)), (dk, pn - po)):
            arr[:, 2 * k + 1], arr[:, 2 * k] = g, -g
    m_raw, s_raw = L[:, 8], L[:, 9]
    in_m, in_s = (m_raw >= -3) & (m_raw <= 3), (s_raw >= DO.MIN_LOG) & (s_raw <= DO.MAX_LOG)     # the clamps gate the gradient
    mean, ls = DO.clip_params(m_raw, s_raw)
    mean_o, ls_o = DO.clip_params(O[:, 8], O[:, 9])
    std, std_o = np.exp(ls), np.exp(ls_o)
    u = DO.unsquash(b["mouse"][:, 0].astype(np.float64), low, high)
    z = (u - mean) / std
    logp += DO.normal_logpdf(u, mean, ls) - (DO.normal_logpdf(u, 0.0, np.log(DO.S)) + np.log(high - low))
    ent += np.log(high - low) - (np.log(DO.S) - ls + (std ** 2 + mean ** 2) / (2 * DO.S ** 2) - 0.5)
    q = (std_o ** 2 + (mean_o - mean) ** 2) / std ** 2
    kl += ls - ls_o + 0.5 * q - 0.5
    dlp[:, 8], dlp[:, 9] = in_m * z / std, in_s * (z * z - 1)
    dh[:, 8], dh[:, 9] = in_m * (-mean / DO.S ** 2), in_s * (1 - std ** 2 / DO.S ** 2)
    dk[:, 8], dk[:, 9] = in_m * (-(mean_o - mean) / std ** 2), in_s * (1 - q)
    ratio = np.exp(logp - b["logp"])
    s1, s2 = b["adv"] * ratio, b["adv"] * np.clip(ratio, 1 - clip_param, 1 + clip_param)
    sur = np.minimum(s1, s2)
    c_lp = np.where(s1 <= s2, -s1, 0.0)                                        # d(-surrogate) / d logp
    v, vo, vt = new_value.astype(np.float64), b["value"].astype(np.float64), b["vtarg"].astype(np.float64)
    dv = v - vo
    vc = vo + np.clip(dv, -vf_clip_param, vf_clip_param)
    e1, e2 = (v - vt) ** 2, (vc - vt) ** 2
    vf = np.maximum(e1, e2)
    dvf = np.where(e1 >= e2, 2 * (v - vt), np.where(np.abs(dv) <= vf_clip_param, 2 * (vc - vt), 0.0))
    dlogits = (c_lp[:, None] * dlp + kl_coeff * dk - entropy_coeff * dh) / n
    total = -sur + kl_coeff * kl + vf_loss_coeff * vf - entropy_coeff * ent
    stats = {"entropy": ent.mean(), "kl": kl.mean(), "policy_loss": (-sur).mean(), "total_loss": total.mean(), "vf_loss": vf.mean()}
    return dlogits, vf_loss_coeff * dvf / n, stats
