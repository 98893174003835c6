"""GPU: tools/train_ppo.py's trainer checkpoints (VERDICT r4 missing 5; reference q1physrl/train.py:110-133: a checkpoint every 100
iterations and whenever the tracked metric exceeds its previous best; params['checkpoint_fname'] resumes).  A short native-learner run
(128-sample minibatches: the persistent learner) writes periodic + best checkpoints; a second process restores one and continues at the
next iteration with the optimizer's step count, moments and the adaptive KL coefficient where the first one left them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train(extra, timeout=600):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_ppo.py"), "--envs", "256", "--horizon", "32", "--minibatch", "128", "--epochs", "2",
           "--lr", "1e-4", "--native", "--fused-policy", "--log-every", "1", "--zero-start-prob", "0.5", "--time-limit", "0.25"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=dict(os.environ, Q1_TUNABLEOP="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    return [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_checkpoint_every_n_best_metric_and_restore(tmp_path):
    import torch
    d = str(tmp_path / "ck")
    rows = _train(["--iters", "5", "--checkpoint-dir", d, "--checkpoint-every", "2"])
    its = [r for r in rows if "iter" in r]
    assert [r["iter"] for r in its] == [0, 1, 2, 3, 4]
    assert all(("checkpoint" in r) == (r["iter"] % 2 == 0 or r["iter"] == 4) for r in its)
    assert os.path.exists(os.path.join(d, "checkpoint_000002.pt")) and os.path.exists(os.path.join(d, "checkpoint_000004.pt"))
    best_rows = [r for r in its if "checkpoint_best" in r]
    assert best_rows and os.path.exists(os.path.join(d, "checkpoint_best.pt"))
    # a new best is saved exactly when the metric exceeds every earlier finite value
    best = float("-inf")
    for r in its:
        z = r["zero_start_total_reward_mean"]
        assert ("checkpoint_best" in r) == (z == z and z > best)
        if z == z and z > best:
            best = z
    ck = torch.load(os.path.join(d, "checkpoint_000002.pt"), map_location="cpu", weights_only=False)
    steps_per_iter = 2 * (256 * 32 // 128)
    assert ck["iter"] == 2 and int(ck["learner"]["native_adam"][:8].view(torch.int64)[0]) == 3 * steps_per_iter
    # resume: continues at iteration 3, the step count keeps counting from the checkpoint's
    rows2 = _train(["--iters", "5", "--restore", os.path.join(d, "checkpoint_000002.pt"), "--checkpoint-dir", str(tmp_path / "ck2"), "--checkpoint-every", "1"])
    assert rows2[0]["restored"].endswith("checkpoint_000002.pt") and rows2[0]["resume_at_iter"] == 3
    its2 = [r for r in rows2 if "iter" in r]
    assert [r["iter"] for r in its2] == [3, 4] and its2[0]["kl_coeff"] > 0
    ck2 = torch.load(os.path.join(str(tmp_path / "ck2"), "checkpoint_000004.pt"), map_location="cpu", weights_only=False)
    assert int(ck2["learner"]["native_adam"][:8].view(torch.int64)[0]) == 5 * steps_per_iter
    assert abs(ck2["best_metric"]) < 1e9 and ck2["best_metric"] >= ck["best_metric"]
