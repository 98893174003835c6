"""Volume check of the fused rollout kernel against the multi-threaded C oracle (oracle/q1_oracle.c: the reference's arithmetic on the
host's libm): R rounds of 65 536 envs x 720 ticks, each with its own start states (random yaw / speed / time, 30 % zero starts, the
oracle's state injected into the device env) and its own sticky random actions; every tick's reward and done, the observations of
every 60th tick, and the final velocity / z / yaw / position compared for BIT equality.  Reports mismatch counts instead of stopping:
the device's sin / cos differs from libm's by an ulp in ~3 % of calls, and this measures how often that reaches a float32 velocity
(expected: never at this volume - DESIGN.md section 3).  Test infrastructure (uses oracle/), not part of the product path.

    python tools/soak_oracle.py [--rounds 20] [--envs 65536] [--ticks 720]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import np_oracle as O, c_oracle as CO
from q1physrl_amd.tensor_env import TensorVectorEnv
from q1physrl_amd.env import Config
from test_hip_fastpath import inject

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=20)
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--ticks", type=int, default=720)
args = ap.parse_args()
n, ticks = args.envs, args.ticks
bad = {"reward": 0, "done": 0, "obs": 0, "vel": 0, "z": 0, "yaw": 0}
worst_pos = 0.0
t0 = time.time()
bits = np.arange(4)[None, :]
for rnd in range(args.rounds):
    cfg = O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.3, time_limit=1e9)      # no episode end inside a round
    np.random.seed(100 + rnd)
    ora = CO.COracleVectorEnv(cfg, threads=min(64, os.cpu_count() or 1))
    tenv = TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=1)
    inject(ora, tenv)
    rng = np.random.default_rng(1000 + rnd)
    keys = np.empty((ticks, n), np.uint8)
    cur = rng.integers(0, 16, n, dtype=np.uint8)
    for t in range(ticks):
        flip = (rng.random((4, n)) < 0.05).astype(np.uint8)
        cur = cur ^ (flip[0] | (flip[1] << 1) | (flip[2] << 2) | (flip[3] << 3))
        keys[t] = cur
    mouse = rng.uniform(-10.08, 10.08, (ticks, n)).astype(np.float32)
    obs, rew, done = tenv.rollout(ticks, (torch.from_numpy(keys).cuda(), torch.from_numpy(mouse).cuda()), outputs=True)
    torch.cuda.synchronize()
    rew_g, done_g = rew.cpu().numpy(), done.cpu().numpy().astype(bool)
    dist = np.zeros((n, 2))
    for t in range(ticks):
        a = np.concatenate([((keys[t][:, None] >> bits) & 1).astype(np.float64), mouse[t][:, None].astype(np.float64)], axis=1)
        o, r, d, _ = ora.vector_step(a)
        bad["reward"] += int(np.count_nonzero(r.view(np.uint32) != rew_g[t].view(np.uint32)))
        bad["done"] += int(np.count_nonzero(d != done_g[t]))
        if t % 60 == 59 or t == ticks - 1:
            bad["obs"] += int(np.count_nonzero(o.astype(np.float32).view(np.uint32) != obs[t].cpu().numpy().view(np.uint32)))
        dist += cfg.time_delta * ora.st["vel"][:, :2].astype(np.float64)
    st = tenv.get_state()
    bad["vel"] += int(np.count_nonzero(st["vel_x"].view(np.uint32) != ora.st["vel"][:, 0].copy().view(np.uint32)) +
                      np.count_nonzero(st["vel_y"].view(np.uint32) != ora.st["vel"][:, 1].copy().view(np.uint32)))
    bad["z"] += int(np.count_nonzero(st["z_pos"] != ora.st["z_pos"]))
    bad["yaw"] += int(np.count_nonzero(st["yaw"] != ora.yaw))
    worst_pos = max(worst_pos, float(np.abs(st["pos_x"] - dist[:, 0]).max()), float(np.abs(st["pos_y"] - dist[:, 1]).max()))
    tenv.close()
    print(f"round {rnd}: cumulative mismatches {bad}, max |pos - ref| so far {worst_pos:.3e}  ({time.time() - t0:.0f} s)", flush=True)
total = args.rounds * n * ticks
print(f"soak_oracle: {args.rounds} rounds x {n} envs x {ticks} ticks = {total / 1e9:.2f} G env-steps against the C oracle: mismatching elements {bad}; "
      f"max |pos - ref| {worst_pos:.3e}")
sys.exit(0 if not any(bad.values()) else 1)
