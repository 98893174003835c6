"""Soak of the resident sampler's LDS hand-offs: H horizons of T ticks at N envs (params.yml-like Config with frequent resets),
every trajectory tensor compared with the two-launch sampler's after every horizon (bit-equality), then the env state and the episode
statistics.  One stale, torn or skipped hand-off anywhere would show.

    python tools/soak_resident.py [--envs 32768] [--horizons 200] [--ticks 128]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from q1physrl_amd import policy as P
from q1physrl_amd.env import Config
from q1physrl_amd.sampler import GpuSampler
from q1physrl_amd.tensor_env import TensorVectorEnv

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=32768)
ap.add_argument("--horizons", type=int, default=200)
ap.add_argument("--ticks", type=int, default=128)
args = ap.parse_args()
cfgd = dict(Config.get_default().__dict__, num_envs=args.envs, time_limit=1.5, zero_start_prob=0.2)
samplers = []
for resident in (False, True):
    torch.manual_seed(0)
    env = TensorVectorEnv(Config(**cfgd), device=0, seed=11)
    pol = P.Q1Policy().cuda()
    with torch.no_grad():
        for p_ in pol.parameters():
            p_.mul_(3.0)
    samplers.append((env, GpuSampler(env, P.FusedPolicyForward(pol, env), horizon=args.ticks, resident=resident, use_graph=not resident)))
t0 = time.time()
for h in range(args.horizons):
    a = samplers[0][1].collect()
    b = samplers[1][1].collect()
    for k in a:
        if not torch.equal(a[k], b[k]):
            raise SystemExit(f"MISMATCH in horizon {h}, tensor {k}: {float((a[k] != b[k]).float().mean()):.3g} of the elements differ")
torch.cuda.synchronize()
assert not samplers[1][1].resident_status().any(), samplers[1][1].resident_status()
sa, sb = samplers[0][0].get_state(), samplers[1][0].get_state()
for k in sa:
    assert np.array_equal(sa[k], sb[k]), k
assert samplers[0][1].stats == samplers[1][1].stats
st = samplers[1][1].stats
print(f"soak ok: {args.envs} envs x {args.horizons} horizons x {args.ticks} ticks = {args.envs * args.horizons * args.ticks / 1e9:.2f} G env-steps, "
      f"{st['episodes']} episodes finished, every trajectory tensor / state / statistic bit-identical to the two-launch sampler "
      f"({time.time() - t0:.1f} s)")
