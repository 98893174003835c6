"""One sampler configuration under rocprofv3 (kernel trace): which kernels make up a tick of the GPU-resident sampler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from q1physrl_amd import policy as P
from q1physrl_amd.env import Config
from q1physrl_amd.sampler import GpuSampler
from q1physrl_amd.tensor_env import TensorVectorEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cfg = Config(**{**Config.get_default().__dict__, "num_envs": n})
env = TensorVectorEnv(cfg, seed=1)
pol = P.Q1Policy().cuda()
fused = len(sys.argv) > 2 and sys.argv[2] == "fused"      # the two-launch tick: fused MFMA policy+value forward + q1env_sample_step
s = GpuSampler(env, P.FusedPolicyForward(pol, env) if fused else pol, horizon=32, use_graph=False)
for _ in range(4):
    s.collect()
torch.cuda.synchronize()
