"""BASELINE configs[4] under rocprofv3 (kernel trace): which kernels make up a tick of the GPU-resident sampler, at one batch size.

    python tools/profile_sampler.py ENVS [horizon]

params.yml Config (data/params.yml:16-33), random-init policy of the reference's shape.  Runs the two-launch tick (fused matrix-core
policy + value forward, then the fused sample / step / reset kernel; eager launches so that every kernel shows up as its own
dispatch) and the resident sampler (one dispatch per horizon + one batched value forward; above 65 536 envs its workgroups
run as successive sets).  Prints the HIP-event time per tick of both; the per-kernel statistics come from the profiler around it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from q1physrl_amd import policy as P
from q1physrl_amd.env import Config
from q1physrl_amd.sampler import GpuSampler
from q1physrl_amd.tensor_env import TensorVectorEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 128
for label, kw in (("two_launch", dict(use_graph=False)), ("resident", dict(resident=True))):
    env = TensorVectorEnv(Config(num_envs=n, **bench.PARAMS_YML), device=0, seed=1)
    s = GpuSampler(env, P.FusedPolicyForward(P.Q1Policy().cuda(), env), horizon=horizon, **kw)
    s.collect(); s.collect()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(4):
        env.use_current_stream()
        e0.record(); s.collect(check_status=False); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / horizon)
    print(f"n={n} {label}: {best:.2f} us per tick = {n / best / 1e3:.3f} G env-steps/s (horizon {horizon}, HIP events, best of 4)")
    env.close()
    del s, env
    torch.cuda.empty_cache()
