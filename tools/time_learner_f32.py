#!/usr/bin/env python3
"""Time q1env_learner_sgd_epochs_f32 (the float32-arithmetic persistent learner) at the reference's shape next to the float16 kernel.  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from q1physrl_amd import ppo
    import test_hip_learner as T
    epochs = int(os.environ.get("EPOCHS", "10"))
    pol = T._policy(7, 1.0)
    env, full, total = T._train_batch(128, 391, pol)
    env._dev.learner_set_exchange_mode(os.environ.get("MODE", "auto"))
    if os.environ.get("PROF"):
        env._dev.learner_set_profiling(int(os.environ["PROF"]))
    klc = torch.tensor(0.2, device="cuda")
    nat = ppo.NativeStep(pol, env, 128, splits=8)
    hp = (5e-6, (0.9, 0.999), 1e-8)
    perms = torch.stack([torch.randperm(total, device="cuda") for _ in range(epochs)]).contiguous()
    out = {"epochs": epochs, "steps": epochs * (total // 128)}
    for name, f32 in (("f32", True), ("f16", False)):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = nat.epochs(full, perms, 0.3, 10.0, 1.0, 0.01, klc, hp, refresh_images=False, f32=f32)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out[f"{name}_rep{rep}_us_per_step"] = dt / n * 1e6
        out[f"{name}_status"] = nat.persistent_status()
        if os.environ.get("PROF") and f32:
            ticks = nat._pws[24:24 + 96].view(torch.int64).cpu().tolist()
            names = ["rows+P1", "barrier1", "gather+P2+L3", "barrier2", "loss", "dW3+B3", "arrive3+small", "dW2+Adam", "barrier3", "B2+dW1", "-", "-"]
            out["f32_prof_us_per_step"] = {nm: round(t * 0.01 / n, 3) for nm, t in zip(names, ticks)}
    print(json.dumps(out), flush=True)
    env.close()


if __name__ == "__main__":
    main()
