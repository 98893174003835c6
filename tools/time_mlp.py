"""Time the fused policy-forward kernel alone (both networks = one sampler-tick's worth) at a few batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from q1physrl_amd import policy as P
from q1physrl_amd.env import Config
from q1physrl_amd.tensor_env import TensorVectorEnv

sizes = [int(a) for a in sys.argv[1:]] or [32768, 262144, 1048576]
for n in sizes:
    env = TensorVectorEnv(Config(**{**Config.get_default().__dict__, "num_envs": n}), seed=1)
    f = P.FusedPolicyForward(P.Q1Policy().cuda(), env)
    obs = torch.randn((n, 6), device="cuda")
    res = []
    for sep in (True, False):
        for _ in range(20):
            f(obs, separate_launches=sep)
        torch.cuda.synchronize()
        reps = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f(obs, separate_launches=sep)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / reps)
    flops = 2 * n * 2 * (16 * 256 + 256 * 256 + 256 * 32)                # both networks, padded shapes
    print(f"n={n}: policy+value forward {res[0]:.1f} us as two launches, {res[1]:.1f} us as one launch "
          f"({flops / res[1] * 1e-6:.0f} TFLOP/s on the padded shapes)")
    env.close()
