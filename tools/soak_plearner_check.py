#!/usr/bin/env python3
"""The persistent learner under the ASSERTION build of the library (libq1env_check.so = the same sources with -DQ1_CHECK; select it through
Q1ENV_LIB_PATH before q1physrl_amd is imported - tests/test_hip_learner.py does): a whole update of the reference's shape, 30 epochs x 391
minibatches of 128 = 11 730 SGD steps as one dispatch, in both exchange modes, with every exchange offset, row index, schedule position and
barrier reading checked on the device (csrc/q1learner_persist.hpp "the assertion build"); then a planted out-of-range row index, which must
come back as status 0x102 - not as a memory fault.  Prints one summary line; exit code 0 = all as expected."""
import os
import sys

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
    c0 = env._dev.learner_debug_counters()
    if c0[0] != 1:
        print("soak_plearner_check: the loaded library was not built with -DQ1_CHECK (set Q1ENV_LIB_PATH to libq1env_check.so)")
        return 2
    klc = torch.tensor(0.2, device="cuda")
    nat = ppo.NativeStep(pol, env, 128, splits=8)
    hp = (5e-6, (0.9, 0.999), 1e-8)
    perms = torch.stack([torch.randperm(total, device="cuda") for _ in range(epochs)]).contiguous()
    steps = 0
    # the float16 kernel: the whole update; the float32 kernel (csrc/q1learner_persist32.hpp: the same accessors, the same assertions): a third of one
    for f32, mode in ((False, "auto"), (False, "agent"), (True, "auto"), (True, "agent")):
        env._dev.learner_set_exchange_mode(mode)
        n = nat.epochs(full, perms[:10] if f32 else perms, 0.3, 10.0, 1.0, 0.01, klc, hp, refresh_images=False, f32=f32)
        torch.cuda.synchronize()
        st = nat.persistent_status()
        print(f"{'float32' if f32 else 'float16'} kernel, mode {mode}: {n} steps, status {st}, counters {env._dev.learner_debug_counters()}", flush=True)
        if st[0] != 0:
            print("soak_plearner_check FAILED: status", st)
            return 1
        if (mode == "auto") != (st[2] != 0 and st[3] != 0):
            print("soak_plearner_check FAILED: exchange mode", mode, "ran as", st[2:])
            return 1
        steps += n
    built, nx, nrow, nbar, nfail = env._dev.learner_debug_counters()
    ok = nfail == 0 and nx > 50 * steps and nrow >= 4 * steps and nbar >= 3 * steps and all(torch.isfinite(p).all() for p in pol.parameters())
    # a planted row index beyond the batch: reported, the access skipped, the process alive
    bad = perms.clone()
    bad[0, 5 * 128 + 77] = total + 12345
    nat.epochs(full, bad, 0.3, 10.0, 1.0, 0.01, klc, hp, steps=10, refresh_images=False)
    torch.cuda.synchronize()
    st = nat.persistent_status()
    planted = st[0] == 0x102 and st[1] == total + 12345
    nat.epochs(full, perms, 0.3, 10.0, 1.0, 0.01, klc, hp, steps=10, refresh_images=False)      # ... and the next launch is healthy again
    torch.cuda.synchronize()
    again = nat.persistent_status()[0] == 0
    env.close()
    if not (ok and planted and again):
        print(f"soak_plearner_check FAILED: counters {(built, nx, nrow, nbar, nfail)}, planted status {st}, next launch ok {again}")
        return 1
    print(f"soak_plearner_check ok: {steps} steps, {nx} exchange accesses + {nrow} row indices + {nbar} barrier readings checked, {nfail} failures; "
          f"planted index -> status 0x102")
    return 0


if __name__ == "__main__":
    sys.exit(main())
