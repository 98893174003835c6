"""Error of the fused policy forward against the float32 torch modules (random well-scaled weights and the WR policy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from q1physrl_amd import policy as P
from q1physrl_amd.env import Config
from q1physrl_amd.tensor_env import TensorVectorEnv

n = 32768
env = TensorVectorEnv(Config(**{**Config.get_default().__dict__, "num_envs": n}), seed=1)
torch.manual_seed(0)
obs = (torch.randn((n, 6), device="cuda") * torch.tensor([0.5, 3.0, 0.3, 2.0, 2.0, 1.0], device="cuda")).contiguous()
pols = {"random (1.5/sqrt(fan_in))": P.Q1Policy().cuda()}
with torch.no_grad():
    for net in (pols["random (1.5/sqrt(fan_in))"].pi, pols["random (1.5/sqrt(fan_in))"].vf):
        for layer in (net[0], net[2], net[4]):
            layer.weight.copy_(torch.randn_like(layer.weight) * (1.5 / layer.in_features ** 0.5))
            layer.bias.copy_(torch.randn_like(layer.bias) * 0.3)
w = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wr_policy.npz")))
pols["WR checkpoint"] = P.load_rllib_fcnet_weights(P.Q1Policy(), w).cuda()
for name, pol in pols.items():
    f = P.FusedPolicyForward(pol, env)
    lg, v = f(obs)
    with torch.no_grad():
        rl, rv = pol(obs)
    torch.cuda.synchronize()
    print(f"{name}: logits |err| max {float((lg - rl).abs().max()):.2e} mean {float((lg - rl).abs().mean()):.2e} (|logits| mean {float(rl.abs().mean()):.2f}); "
          f"value |err| max {float((v - rv).abs().max()):.2e} mean {float((v - rv).abs().mean()):.2e} (|value| mean {float(rv.abs().mean()):.2f})")
