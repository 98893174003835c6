"""Run the fused policy-forward kernel alone (for rocprofv3 kernel-trace / PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from q1physrl_amd import policy as P
from q1physrl_amd.env import Config
from q1physrl_amd.tensor_env import TensorVectorEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
env = TensorVectorEnv(Config(**{**Config.get_default().__dict__, "num_envs": n}), seed=1)
f = P.FusedPolicyForward(P.Q1Policy().cuda(), env)
obs = torch.randn((n, 6), device="cuda")
for _ in range(10):
    f(obs)
torch.cuda.synchronize()
