"""Per-tick time of the resident tick server (q1env_step_persistent_*) fed by the dependent reference producer, next to the per-tick
step kernel replayed from a hipGraph and the fused rollout kernel on the same actions.  HIP events on the server's stream.

    python tools/time_persistent.py [--envs 65536] [--ticks 720] [--reps 5]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[4096, 65536, 131072, 196608, 262144])
    ap.add_argument("--ticks", type=int, default=720)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from q1physrl_amd import _lib, env as E
    from q1physrl_amd.tensor_env import TensorVectorEnv
    rows = []
    for n in args.envs:
        cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
        env = TensorVectorEnv(cfg, device=0, seed=1)
        env.reset()
        T = args.ticks
        keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device="cuda")
        mouse = (torch.rand((T, n), device="cuda") * 20 - 10).contiguous()
        dev = env._dev
        row = {"envs": n, "ticks": T}
        for label, two in (("server_us_per_tick", False), ("server_two_streams_us_per_tick", True)):
            try:
                env.serve_ticks(keys, mouse, two_streams=two)                  # warm-up
            except _lib.Q1EnvError as ex:
                row[label.replace("us_per_tick", "refused")] = str(ex)[-90:]
                continue
            best = 1e30
            for _ in range(args.reps):
                env.reset()
                torch.cuda.synchronize()
                dev.timer_start()
                env.serve_ticks(keys, mouse, sync=False, two_streams=two)
                ms = dev.timer_stop()
                st = env._srv["status"].cpu().numpy()
                assert not st.any(), st
                best = min(best, ms * 1e3 / T)
            row[label] = best
        obs = torch.empty((n, 6), device="cuda"); rew = torch.empty((n,), device="cuda"); done = torch.empty((n,), dtype=torch.uint8, device="cuda")
        for g in (2, 1):
            dev.step_many_dev(T, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), 0, g)
        torch.cuda.synchronize()
        dev.timer_start()
        for _ in range(args.reps):
            dev.step_many_dev(T, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), 0, 1)
        row["step_graph_us_per_tick"] = dev.timer_stop() * 1e3 / (T * args.reps)
        dev.timer_start()
        for _ in range(args.reps):
            dev.rollout_dev(T, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, _lib.OBS_F32, 0, 0, 0, True)
        row["fused_rollout_no_outputs_us_per_tick"] = dev.timer_stop() * 1e3 / (T * args.reps)
        if "server_us_per_tick" in row:
            row["server_frac_of_8TBps_at_204B"] = 204.0 * n / (row["server_us_per_tick"] * 1e-6) / 8e12
        row["step_frac_of_8TBps_at_204B"] = 204.0 * n / (row["step_graph_us_per_tick"] * 1e-6) / 8e12
        rows.append(row)
        print(json.dumps(row), flush=True)
        env.close()


if __name__ == "__main__":
    main()
