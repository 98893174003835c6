"""Time the native learner's kernels on one MI355X: `--steps` forward + backward + weight-gradient launches of a `--mb`-sample
minibatch (32 768 = tools/train_ppo.py's) on synthetic rows, HIP events around each phase; meant to be run under
`rocprofv3 --kernel-trace --stats` / `--pmc ...` as well (tools/profile_learner.sh).  Prints one JSON line."""
import argparse
import json
import time
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--splits", type=int, default=32)
    ap.add_argument("--phase", default="all", choices=["all", "forward", "backward", "step"])
    ap.add_argument("--step-mode", default="auto", choices=["auto", "four_launch", "fused", "fused_dw1", "fused_dw1_q", "fused_dw1_r4wgrad"],
                    help="kernel sequence of q1env_learner_sgd_step (include/q1env.h q1env_learner_set_step_mode)")
    ap.add_argument("--two-call", action="store_true", help="the step as q1env_learner_step + q1env_learner_adam instead of q1env_learner_sgd_step")
    args = ap.parse_args()
    import torch
    from q1physrl_amd import policy as P, ppo
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    env = TensorVectorEnv(Config(**dict(Config.get_default().__dict__, num_envs=256)), device=0, seed=1)
    env._dev.learner_set_step_mode(args.step_mode)
    torch.manual_seed(0)
    pol = P.Q1Policy().cuda()
    mb = args.mb
    total = 4 * mb
    g = torch.Generator(device="cuda").manual_seed(2)
    obs = torch.randn((total, 6), device="cuda", generator=g) * torch.tensor([0.5, 3.0, 0.3, 1.5, 1.5, 1.0], device="cuda")
    idx = torch.randperm(total, device="cuda", generator=g)[:mb].contiguous()
    nat = ppo.NativeStep(pol, env, mb, splits=args.splits)
    dl = torch.randn((mb, 10), device="cuda", generator=g)
    dv = torch.randn((mb,), device="cuda", generator=g) * 100.0

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host_us.append((time.perf_counter() - t0) * 1e6 / n)      # what the host needed to ENQUEUE one call (the queue does not fill at these counts)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    host_us = []

    out = {"mb": mb, "splits": args.splits, "steps": args.steps, "step_mode": args.step_mode}
    if args.phase in ("all", "forward"):
        out["forward_us"] = timed(lambda: nat.forward(obs, idx), args.steps)
    if args.phase in ("all", "backward"):
        out["backward_wgrad_reduce_us"] = timed(lambda: nat.backward(obs, idx, dl, dv, float(mb)), args.steps)
    if args.phase in ("all", "step"):
        # the whole SGD step as PPOLearner runs it: forward, loss gradient, backward, weight gradients, reduction + Adam + images
        total_rows = total
        full = {"obs": obs, "old_logits": torch.randn((total_rows, 10), device="cuda", generator=g).contiguous(),
                "keys_packed": torch.randint(0, 16, (total_rows,), device="cuda", dtype=torch.uint8),
                "mouse": (torch.rand((total_rows, 1), device="cuda", generator=g) * 20 - 10), "logp": -torch.rand((total_rows,), device="cuda", generator=g) * 5,
                "adv": torch.randn((total_rows,), device="cuda", generator=g), "value": torch.randn((total_rows,), device="cuda", generator=g) * 50,
                "vtarg": torch.randn((total_rows,), device="cuda", generator=g) * 50}
        klc = torch.full((1,), 0.2, device="cuda")

        def one():
            if args.two_call:                            # q1env_learner_step + q1env_learner_adam: six launches
                nat.step(full, idx, 0.1, 7500.0, 1.0, 0.01, klc, skip_reduce=True)
                nat.adam(3e-5)
            else:                                        # q1env_learner_sgd_step: four (round 4; what PPOLearner runs)
                nat.step(full, idx, 0.1, 7500.0, 1.0, 0.01, klc, skip_reduce=True, adam=(3e-5, (0.9, 0.999), 1e-8))
        out["eager_step_us"] = timed(one, args.steps)
        out["eager_step_host_enqueue_us"] = host_us[-1]
        gph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            env.use_current_stream()
            one()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(gph):
            env.use_current_stream()                     # the library's launches must land on the capturing stream
            one()
        env.use_current_stream()
        out["graph_step_us"] = timed(gph.replay, args.steps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
