#!/usr/bin/env python3
"""bench.py - env-steps/s of the q1physrl hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--mode step|rollout]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one tick (one VectorPhysEnv.vector_step, reference env.py:482-510) of the whole batch.
Workload at N=1 = BASELINE.json configs[1]: 65 536 envs, zero-start 100 m run (zero_start_prob = 1,
get_default Config, dt = 1/72, 720-tick episodes), random actions (keys flip with p = 0.05 per tick,
mouse ~ U(-action_range, action_range) float32).  Inputs (the packed 5 B/env action tensor for a whole
720-tick episode) are resident in HBM before the timed region; every tick writes obs float32 (N,6),
reward float32, done uint8; all envs are reset on device at each episode end (inside the timed region).
Multi-GPU: one process per GPU, the batch is split (65 536 envs per GPU, weak scaling), no collective
on the data path; ranks only meet in the barriers around the timed region.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 204.0            # algorithmic bytes per env-step (SURVEY.md 8d / DESIGN.md)
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
EPISODE_TICKS = 720


def make_actions(n, ticks, action_range, seed):
    """Packed actions for one episode, tick-major: keys uint8 (ticks, n) bit k = Key k, mouse float32 (ticks, n)."""
    rng = np.random.default_rng(seed)
    keys = np.empty((ticks, n), dtype=np.uint8)
    cur = rng.integers(0, 16, size=n, dtype=np.uint8)
    for t in range(ticks):
        flip = np.zeros(n, dtype=np.uint8)
        for k in range(4):
            flip |= (rng.random(n) < 0.05).astype(np.uint8) << k
        cur = cur ^ flip
        keys[t] = cur
    mouse = rng.uniform(-action_range, action_range, size=(ticks, n)).astype(np.float32)
    return keys, mouse


def cpu_baseline(n, action_range, budget_s=12.0):
    """The NumPy oracle (a from-scratch restatement of the reference's NumPy path, bit-pinned to the reference
    by tests/golden) timed on this box's host: 1 process, 1 thread (NumPy elementwise kernels are single-threaded)."""
    from oracle import np_oracle as O
    np.random.seed(0)
    env = O.OracleVectorEnv(O.OracleConfig.get_default(num_envs=n, zero_start_prob=1.0))
    rng = np.random.default_rng(1)
    acts = [np.concatenate([(rng.random((n, 4)) < 0.5).astype(np.float64),
                            rng.uniform(-action_range, action_range, (n, 1)).astype(np.float32).astype(np.float64)], axis=1)
            for _ in range(8)]
    env.vector_step(acts[0])                       # warm-up
    t0 = time.perf_counter()
    ticks = 0
    while time.perf_counter() - t0 < budget_s:
        env.vector_step(acts[ticks % len(acts)])
        ticks += 1
    dt = time.perf_counter() - t0
    return {"value": n * ticks / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{ticks} ticks of {n} envs (ndarray actions, oracle/np_oracle.py) in {dt:.1f} s; host has {os.cpu_count()} logical cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7200)
    ap.add_argument("--warmup", type=int, default=720)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--mode", choices=("step", "rollout"), default="step")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from q1physrl_amd import _lib, env as E
    from q1physrl_amd.device import DeviceEnv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = args.envs
    cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    dev = DeviceEnv(cfg, device=local_rank, env_index_base=rank * n)    # own stream; global env index keys the RNG
    ar = float(cfg.action_range)
    keys_h, mouse_h = make_actions(n, EPISODE_TICKS, ar, seed=1234 + rank)
    d = torch.device("cuda", local_rank)
    keys = torch.from_numpy(keys_h).to(d)
    mouse = torch.from_numpy(mouse_h).to(d)
    obs = torch.empty((n, 6), dtype=torch.float32, device=d)
    reward = torch.empty((n,), dtype=torch.float32, device=d)
    done = torch.empty((n,), dtype=torch.uint8, device=d)
    if args.mode == "rollout":       # tick-major per-tick outputs for a whole episode
        obs = torch.empty((EPISODE_TICKS, n, 6), dtype=torch.float32, device=d)
        reward = torch.empty((EPISODE_TICKS, n), dtype=torch.float32, device=d)
        done = torch.empty((EPISODE_TICKS, n), dtype=torch.uint8, device=d)
    torch.cuda.synchronize()

    def run_ticks(k, tick0):
        """k ticks starting at episode phase tick0 % 720; resets every env on device at each episode end."""
        t = tick0
        left = k
        launches = 0
        while left > 0:
            ph = t % EPISODE_TICKS
            chunk = min(left, EPISODE_TICKS - ph)
            ka = keys.data_ptr() + ph * n
            ma = mouse.data_ptr() + ph * n * 4
            if args.mode == "step":
                dev.step_many_dev(chunk, _lib.ACT_PACKED, ka, ma, _lib.OBS_F32, obs.data_ptr(), reward.data_ptr(),
                                  done.data_ptr(), out_stride_ticks=0, use_graph=not args.no_graph)
                launches += chunk
            else:
                dev.rollout_dev(chunk, _lib.ACT_PACKED, ka, ma, 0, _lib.OBS_F32, obs.data_ptr(), reward.data_ptr(),
                                done.data_ptr(), auto_reset=False)
                launches += 1
            t += chunk
            left -= chunk
            if t % EPISODE_TICKS == 0:
                dev.reset_philox_dev(seed=99, done_only=True)          # zero_start_prob = 1: every env back to the start line
        return launches

    def barrier():
        dev.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    run_ticks(args.warmup, 0)
    barrier()
    t0 = time.perf_counter()
    dev.timer_start()
    launches = run_ticks(args.steps, args.warmup)
    ev_ms = dev.timer_stop()                      # HIP events on the stream the kernels were launched on
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64, device=d)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    total_env_steps = float(n) * args.steps * world
    value = total_env_steps / wall
    units_per_launch = n * (args.steps / launches)
    kern_us = ev_ms * 1e3 / launches
    achieved = B_ALG * units_per_launch / (kern_us * 1e-6) / 1e9
    out = {
        "metric": "env-steps/sec @ 64k envs, 1/2/4/8 MI355X; max |pos - NumPy ref| over 10 s",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 arithmetic, f32 vel/obs/reward storage", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: {n} envs/GPU, zero-start 100 m run, random actions, get_default Config, "
                               f"720-tick episodes, mode={args.mode}" + ("" if args.no_graph or args.mode != "step" else "+hipGraph"),
                   "envs_per_gpu": n, "parallelism": f"batch-split x{world}, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": "step_kernel<float>" if args.mode == "step" else "rollout_kernel<float>",
                     "avg_launch_us": kern_us, "alg_bytes_per_env_step": B_ALG,
                     "note": "achieved = 204 B x env-steps per launch / (HIP-event time of the timed region / launches); "
                             "traffic: see profiles/ (PMC pass is a separate run)"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, ar)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
