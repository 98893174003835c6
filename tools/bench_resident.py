"""BASELINE configs[4]'s per-GPU shard (and neighbours): the sampler horizon as ONE dispatch (GpuSampler(resident=True):
q1env_sample_resident + one batched value forward) next to the two-launch-per-tick sampler (hipGraph-captured), same policy, same
Config (params.yml).  HIP-event times per tick; also the split resident dispatch / value forward.

    python tools/bench_resident.py [--envs 8192 32768 43520 65536] [--horizon 128]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from q1physrl_amd import policy as P
from q1physrl_amd.env import Config
from q1physrl_amd.sampler import GpuSampler
from q1physrl_amd.tensor_env import TensorVectorEnv

PARAMS_YML = dict(action_range=10, allow_jump=True, allow_yaw=True, auto_jump=False, discrete_yaw_steps=-1, fmove_max=800,
                  smove_max=1060, hover=False, initial_yaw_range=(0, 360), key_press_delay=0.3, max_initial_speed=700,
                  smooth_keys=True, speed_reward=False, time_delta=0.013888888888888, time_limit=10, zero_start_prob=0.01)

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, nargs="+", default=[8192, 32768, 65536])
ap.add_argument("--horizon", type=int, default=128)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
T = args.horizon
for n in args.envs:
    row = {"envs": n, "horizon": T}
    for label, kw in (("two_launch_graph", dict(use_graph=True)), ("resident", dict(resident=True))):
        env = TensorVectorEnv(Config(num_envs=n, **PARAMS_YML), seed=1)
        pol = P.Q1Policy().cuda()
        s = GpuSampler(env, P.FusedPolicyForward(pol, env), horizon=T, **kw)
        try:
            s.collect(); s.collect()
        except Exception as ex:   # noqa: BLE001
            row[label] = "refused: " + str(ex)[-70:]
            env.close()
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e30
        for _ in range(args.reps):
            env.use_current_stream()
            e0.record(); s.collect(check_status=False); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / T)
        row[label + "_us_per_tick"] = round(best, 2)
        row[label + "_M_env_steps_per_s"] = round(n / best, 1)
        if label == "resident":
            assert not s.resident_status().any(), s.resident_status()
            # the dispatch alone (without the batched value forward)
            dev = env._dev
            best = 1e30
            for _ in range(args.reps):
                torch.cuda.synchronize(); dev.timer_start()
                pi = s.policy._mlp("pi", s.logits.view(T * n, -1))
                dev.sample_resident_dev(T, pi, env.seed, s.tick.data_ptr(), 0, False, s.keys.data_ptr(), s.mouse.data_ptr(),
                                        s.logp.data_ptr(), s.obs.data_ptr(), s.reward.data_ptr(), s.done.data_ptr(), env.zero_start.data_ptr(),
                                        s.ep_return.data_ptr(), s._stats.data_ptr(), s._status.data_ptr(), 5.0)
                best = min(best, dev.timer_stop() * 1e3 / T)
            row["resident_dispatch_only_us_per_tick"] = round(best, 2)
        env.close()
    if "resident_us_per_tick" in row and "two_launch_graph_us_per_tick" in row:
        row["speedup"] = round(row["two_launch_graph_us_per_tick"] / row["resident_us_per_tick"], 2)
    print(json.dumps(row), flush=True)
