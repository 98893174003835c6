"""Phase times inside q1pol::mlp_forward_kernel (wave 0 of workgroup 0), from a -DQ1POL_TRACE build of libq1env.so:
    python -c "from q1physrl_amd import build as b; b.build_lib(force=True, extra_flags=['-DQ1POL_TRACE'])"
(rebuild without the flag afterwards: the trace overwrites the first outputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from q1physrl_amd import policy as P
from q1physrl_amd.env import Config
from q1physrl_amd.tensor_env import TensorVectorEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
env = TensorVectorEnv(Config(**{**Config.get_default().__dict__, "num_envs": n}), seed=1)
f = P.FusedPolicyForward(P.Q1Policy().cuda(), env)
obs = torch.randn((n, 6), device="cuda")
for _ in range(5):
    logits, _ = f(obs)
torch.cuda.synchronize()
pro, loop, l3, chunks, real, mem = logits[0, :6].tolist()
tick_ns = real * 10.0 / mem                       # wall_clock64 is 100 MHz; calibrates the s_memtime tick
print(f"s_memtime tick = {tick_ns:.3f} ns ({1e3 / tick_ns:.0f} MHz); tile loop of wave 0 took {real * 10.0:.0f} ns")
print(f"chunks={chunks:.0f}  per 32-env tile: prologue {pro / chunks * tick_ns:.0f} ns, layer-2 loop {loop / chunks * tick_ns:.0f} ns, "
      f"layer-3 phase {l3 / chunks * tick_ns:.0f} ns, total {(pro + loop + l3) / chunks * tick_ns:.0f} ns")
