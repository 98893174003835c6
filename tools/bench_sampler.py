"""BASELINE configs[4] on one GPU: the env inside a GPU-resident sampler loop with a torch-ROCm policy forward
(MLP 6->256->256->10 + value net) per tick.  Reports env-steps/s and the env-only share of a tick."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from q1physrl_amd import policy as P
from q1physrl_amd.env import Config
from q1physrl_amd.sampler import GpuSampler
from q1physrl_amd.tensor_env import TensorVectorEnv

PARAMS_YML = dict(action_range=10, allow_jump=True, allow_yaw=True, auto_jump=False, discrete_yaw_steps=-1, fmove_max=800,
                  smove_max=1060, hover=False, initial_yaw_range=(0, 360), key_press_delay=0.3, max_initial_speed=700,
                  smooth_keys=True, speed_reward=False, time_delta=0.013888888888888, time_limit=10, zero_start_prob=0.01)

for n in (32768, 262144):
  for use_graph in (False, True):
    for ac in (None, torch.bfloat16, "fused"):
        env = TensorVectorEnv(Config(num_envs=n, **PARAMS_YML), seed=1)
        pol = P.Q1Policy().cuda()
        T = 64
        s = GpuSampler(env, P.FusedPolicyForward(pol, env) if ac == "fused" else pol, horizon=T,
                       autocast_dtype=None if ac == "fused" else ac, use_graph=use_graph)
        s.collect(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            s.collect()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (reps * T)
        # env-only: the same ticks without the policy forward
        keys, mouse = s.keys, s.mouse
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(reps):
            for t in range(T):
                env.step_tensor((keys[t], mouse[t])); env.reset_done()
        torch.cuda.synchronize()
        de = (time.perf_counter() - t1) / (reps * T)
        print(f"n={n:7d} graph={int(use_graph)} policy={'torch-fp32' if ac is None else ('fused-mfma' if ac == 'fused' else 'torch-bf16-autocast')}: {dt*1e6:8.1f} us/tick = {n/dt/1e6:8.2f} M env-steps/s; "
              f"env step+reset alone {de*1e6:7.1f} us/tick ({100*de/dt:4.1f} % of the tick)")
        env.close()
