#!/usr/bin/env python3
"""Per-tensor gradient error of the float32 persistent learner, the float16 one and float32 torch autograd against FLOAT64 torch autograd of
the same 128-row minibatch (who is closest to the truth).  Prints a table."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from q1physrl_amd import ppo
    import test_hip_learner as T
    pol = T._policy(5, 2.0)
    env, full, total = T._train_batch(64, 8, pol)
    klc = torch.tensor(0.2, device="cuda")
    perm = torch.randperm(total, device="cuda")
    hp = (3e-4, (0.9, 0.999), 1e-8)
    idx = perm[:128]

    def autograd(dtype):
        p = copy.deepcopy(pol).to(dtype)
        mb = {"obs": full["obs"][idx].to(dtype), "old_logits": full["old_logits"][idx].to(dtype), "mouse": full["mouse"][idx].reshape(-1, 1).to(dtype),
              "logp": full["logp"][idx].to(dtype), "adv": full["adv"][idx].to(dtype), "value": full["value"][idx].to(dtype), "vtarg": full["vtarg"][idx].to(dtype),
              "keys": ((full["keys_packed"][idx].reshape(-1, 1).long() >> torch.arange(4, device="cuda")) & 1)}
        loss, _ = ppo.ppo_loss(p, mb, float(env.config.action_range), 0.3, 10.0, 1.0, 0.01, klc.to(dtype))
        loss.backward()
        return [q.grad.double() for q in p.parameters()]

    g64, g32 = autograd(torch.float64), autograd(torch.float32)
    res = {}
    for name, f32 in (("kernel f32", True), ("kernel f16", False)):
        p = copy.deepcopy(pol)
        nat = ppo.NativeStep(p, env, 128, splits=8)
        nat.epochs(full, perm.reshape(1, -1).contiguous(), 0.3, 10.0, 1.0, 0.01, klc, hp, steps=1, f32=f32)
        torch.cuda.synchronize()
        res[name] = [q.grad.double() for q in p.parameters()]
    res["torch f32"] = g32
    names = [n for n, _ in pol.named_parameters()]
    print(f"{'tensor':14s}" + "".join(f"{k:>14s}" for k in res))
    for i, n in enumerate(names):
        print(f"{n:14s}" + "".join(f"{float((res[k][i] - g64[i]).norm() / g64[i].norm()):14.2e}" for k in res))
    env.close()


if __name__ == "__main__":
    main()
