#!/usr/bin/env python3
"""Time the persistent learner (q1env_learner_sgd_epochs) at the reference's shape: 30 epochs x 391 minibatches of 128 out of a 50 048-sample
train batch, one dispatch - next to the four-launch q1env_learner_sgd_step replayed from a hipGraph.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from q1physrl_amd import ppo
    import test_hip_learner as T
    epochs = int(os.environ.get("EPOCHS", "30"))
    pol = T._policy(7, 1.0)
    env, full, total = T._train_batch(128, 391, pol)
    klc = torch.tensor(0.2, device="cuda")
    nat = ppo.NativeStep(pol, env, 128, splits=8)
    # MODE=auto|agent|census_fail: the exchange mode (q1env_learner_set_exchange_mode); PROF=<g + 8 wave>: the stamped wave (q1env_learner_set_profiling)
    env._dev.learner_set_exchange_mode(os.environ.get("MODE", "auto"))
    if os.environ.get("PROF"):
        env._dev.learner_set_profiling(int(os.environ["PROF"]))
    hp = (5e-6, (0.9, 0.999), 1e-8)
    perms = torch.stack([torch.randperm(total, device="cuda") for _ in range(epochs)]).contiguous()
    out = {"total": total, "epochs": epochs, "steps": epochs * (total // 128)}
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = nat.epochs(full, perms, 0.3, 10.0, 1.0, 0.01, klc, hp, refresh_images=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"persistent_rep{rep}_s"] = dt
        out[f"persistent_rep{rep}_us_per_step"] = dt / n * 1e6
    out["status"] = nat.persistent_status()
    if os.environ.get("PROF"):
        ticks = nat._pws[24:24 + 160].view(torch.int64).cpu().tolist()
        names = ["rows+P1", "barrier1", "gather+P2+L3", "barrier2", "loss", "B3+arrive3", "dW2+Adam", "barrier3", "B2+dW1", "-", "loss:ysum", "loss:ppo", "loss:rows", "-", "G2:wait0", "G2:tile0", "G2:zr+stores0", "G2:tile1", "-", "-"]  # (loss = its three parts + the reductions)
        out["prof_us_per_step"] = {nm: t * 0.01 / n for nm, t in zip(names, ticks)}
    nat.images()
    # the four-launch step, eager (no graph): 391 steps
    nat.cursor.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(391):
        nat.step(full, perms[0], 0.3, 10.0, 1.0, 0.01, klc, skip_reduce=True, use_cursor=True, adam=hp)
    torch.cuda.synchronize()
    out["four_launch_eager_us_per_step"] = (time.perf_counter() - t0) / 391 * 1e6
    print(json.dumps(out), flush=True)
    env.close()


if __name__ == "__main__":
    main()
