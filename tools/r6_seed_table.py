#!/usr/bin/env python3
"""The seed x arithmetic table of the reference-configuration training run (2 989 iterations = 149.45 M env-steps, data/params.yml + RLlib 0.8.4
defaults) from the result files under profiles/: float16 persistent learner with the static loss scale (round 5: seeds 0 - 4; round 6: 5 - 7), with the
dynamic scale (round 5: seeds 1, 2), and the float32 learner (round 6: q1env_learner_sgd_epochs_f32).  Markdown on stdout."""
import glob, json, os, re, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def row(path):
    d = json.load(open(path))
    log = d["log"]
    ev = [r for r in log if "eval_det" in r]
    e600 = min(log, key=lambda r: abs(r["iter"] - 600))
    sat = sum(r.get("grad_saturated_pi", 0) for r in log[-30:]) / max(1, len(log[-30:]))
    return {"det": ev[-1]["eval_det"], "sto": ev[-1]["eval_stochastic"], "det10": sum(r["eval_det"] for r in ev[-10:]) / len(ev[-10:]),
            "ent600": e600["entropy"], "ent_end": log[-1]["entropy"], "min": d["final"]["wall_s"] / 60.0, "sat": sat}


def files(kind):
    out = {}
    pats = {"f16 static": ["r5_train_ppo_refcfg_persistent_v3_seed*.json", "r6_train_ppo_refcfg_f16_seed*.json"],
            "f16 dynamic": ["r5_train_ppo_refcfg_persistent_v3_seed*_dynscale.json"], "f32": ["r6_train_ppo_refcfg_f32_seed*.json"]}[kind]
    for pat in pats:
        for f in glob.glob(os.path.join(P, pat)):
            if kind == "f16 static" and "dynscale" in f:
                continue
            out[int(re.search(r"seed(\d+)", f).group(1))] = f
    return out


kinds = ("f16 static", "f16 dynamic", "f32")
data = {k: {s: row(f) for s, f in files(k).items()} for k in kinds}
seeds = sorted(set().union(*[set(v) for v in data.values()]))
print("| seed | " + " | ".join(f"{k}: det / stoch (last-10 det) · entropy @600 → end" for k in kinds) + " |")
print("|---|" + "---|" * len(kinds))
for s in seeds:
    cells = []
    for k in kinds:
        r = data[k].get(s)
        cells.append("—" if r is None else f"{r['det']:.0f} / {r['sto']:.0f} ({r['det10']:.0f}) · {r['ent600']:.2f} → {r['ent_end']:.2f}")
    print(f"| {s} | " + " | ".join(cells) + " |")
cells = []
for k in kinds:
    v = sorted(r["det"] for r in data[k].values())
    if not v:
        cells.append("—"); continue
    cells.append(f"n = {len(v)}: median {statistics.median(v):.0f}, mean {statistics.mean(v):.0f}, min {v[0]:.0f}, max {v[-1]:.0f}; below 5 500: {sum(x < 5500 for x in v)}; "
                 f"{statistics.mean(r['min'] for r in data[k].values()):.1f} min per run")
print("| all | " + " | ".join(cells) + " |")
