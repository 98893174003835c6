#!/usr/bin/env python3
"""Thin the per-iteration logs of tools/train_ppo.py result files under profiles/ (eleven 1.4 MB logs = 20 of the directory's
24 MB, pushed to the GPU box with every lease): keep every Nth iteration's row + every row that carries an evaluation or a checkpoint + the last
one; args / final untouched; the file says what was done to it.  New runs write thinned files themselves (train_ppo.py --out-stride)."""
import json, os, sys

stride = int(os.environ.get("STRIDE", "10"))
for path in sys.argv[1:]:
    d = json.load(open(path))
    log = d.get("log")
    if not isinstance(log, list) or len(log) < 400 or "log_thinned" in d:
        continue
    last = log[-1].get("iter")
    kept = [r for r in log if r.get("iter", 0) % stride == 0 or r.get("iter") == last or "eval_det" in r or "eval_stochastic" in r]
    d["log"] = kept
    d["log_thinned"] = f"every {stride}th iteration + every evaluation row + the last of {len(log)} rows (tools/thin_train_logs.py); the full per-iteration log is in the git history of this file"
    before = os.path.getsize(path)
    json.dump(d, open(path, "w"))
    print(f"{path}: {len(log)} -> {len(kept)} rows, {before} -> {os.path.getsize(path)} bytes")
