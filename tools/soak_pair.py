"""Soak of q1env_step_persistent_pair (server wave + driver wave per workgroup, LDS hand-offs, LDS-rotated sub-batch states): L launches
of T ticks at N envs with in-kernel resets, against the per-tick q1env_step_autoreset kernels on the same actions - the final state,
the last tick's outputs and the producer's float64 sums of every reward / first observation column it received, per launch.

    python tools/soak_pair.py [--envs 131072 262144 294912] [--launches 20] [--ticks 720]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from q1physrl_amd.env import Config
from q1physrl_amd.tensor_env import TensorVectorEnv

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, nargs="+", default=[131072, 262144, 294912])
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--ticks", type=int, default=720)
args = ap.parse_args()
for n in args.envs:
    cfgd = dict(Config.get_default().__dict__, num_envs=n, time_limit=2.0, zero_start_prob=0.3)
    a = TensorVectorEnv(Config(**cfgd), device=0, seed=3); b = TensorVectorEnv(Config(**cfgd), device=0, seed=3)
    a.reset(); b.reset()
    T = args.ticks
    g = torch.Generator(device="cuda").manual_seed(8)
    keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device="cuda", generator=g)
    mouse = ((torch.rand((T, n), device="cuda", generator=g) * 2 - 1) * 10.0).contiguous()
    t0 = time.time()
    episodes = 0
    for l in range(args.launches):
        want = torch.zeros((2, n), dtype=torch.float64, device="cuda")
        for t in range(T):
            obs_b, rew_b, done_b = b.step_autoreset((keys[t], mouse[t]))
            episodes += int(done_b.sum())
            if t != T - 1:
                want[0] += rew_b.double(); want[1] += obs_b[:, 0].double()
        res = a.serve_ticks(keys, mouse)
        assert not res["status"].any(), (n, l, res["status"])
        assert torch.equal(res["checksum"], want), (n, l, "checksum")
        assert torch.equal(res["obs"], obs_b) and torch.equal(res["obs_from_granules"], obs_b) and torch.equal(res["reward"], rew_b) and torch.equal(res["done"], done_b), (n, l)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), (n, k)
    print(f"soak ok: {n} envs x {args.launches} launches x {T} ticks = {n * args.launches * T / 1e9:.2f} G env-steps, {episodes} episodes finished: "
          f"state, last outputs and the producer's sums identical to the per-tick kernels ({time.time() - t0:.1f} s)", flush=True)
    a.close(); b.close()
