"""Where do the microseconds of a 20-tick timed region go?  (The driver runs `bench.py --steps 20 --warmup 5`: ~90 us of GPU
work, so host-side launch and synchronisation latency is a visible fraction of `ms_per_step`.)  Repeats the region many times
and prints the distribution of: host time of the graph launch call, wall from t0 to the end of the synchronisation (per sync
flavour), and the HIP-event time of the same region.

    python tools/timing_probe.py [--envs 65536] [--ticks 20] [--reps 300]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--ticks", type=int, default=20)
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    import torch
    from q1physrl_amd import _lib, env as E
    from q1physrl_amd.device import DeviceEnv
    n, T = args.envs, args.ticks
    d = torch.device("cuda", 0)
    cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    dev = DeviceEnv(cfg, device=0)
    keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device=d)
    mouse = torch.rand((T, n), device=d) * 20 - 10
    obs = torch.empty((n, 6), device=d)
    rew = torch.empty((n,), device=d)
    done = torch.empty((n,), dtype=torch.uint8, device=d)
    torch.cuda.synchronize()

    def go(g):
        dev.step_many_dev(T, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(),
                          done.data_ptr(), out_stride_ticks=0, use_graph=g)

    out = {"envs": n, "ticks": T, "reps": args.reps}
    for label, g, sync in (("graph+torch_sync", 1, torch.cuda.synchronize), ("graph+stream_sync", 1, dev.sync),
                           ("eager+torch_sync", 0, torch.cuda.synchronize)):
        go(2 if g else 0)
        go(g)
        torch.cuda.synchronize()
        launch, wall, ev = [], [], []
        for _ in range(args.reps):
            dev.reset_philox_dev(seed=1, done_only=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.timer_start()
            go(g)
            t1 = time.perf_counter()
            dev.timer_mark()
            sync()
            t2 = time.perf_counter()
            ev.append(dev.timer_elapsed() * 1e3)
            launch.append((t1 - t0) * 1e6)
            wall.append((t2 - t0) * 1e6)
        q = lambda x: {"min": float(np.min(x)), "p50": float(np.median(x)), "p90": float(np.percentile(x, 90))}   # noqa: E731
        out[label] = {"launch_call_us": q(launch), "wall_us": q(wall), "event_us": q(ev),
                      "wall_per_tick_p50": float(np.median(wall)) / T, "event_per_tick_p50": float(np.median(ev)) / T}
    # the resident tick server (one dispatch): where its 20-tick region's wall time goes
    mailbox = torch.zeros((n,), dtype=torch.int64, device=d)
    results = torch.zeros((4, n, 2), dtype=torch.int64, device=d)
    status = torch.zeros((5,), dtype=torch.int32, device=d)
    tag = 0
    for label, flags, sync in (("pair+flags+torch_sync", 1 | _lib.TIMER_START | _lib.TIMER_STOP, torch.cuda.synchronize),
                               ("pair+flags+stream_sync", 1 | _lib.TIMER_START | _lib.TIMER_STOP, dev.sync),
                               ("pair+noevents+stream_sync", 1, dev.sync)):
        launch, wall, ev = [], [], []
        for rep in range(args.reps + 5):
            dev.reset_philox_dev(seed=1, done_only=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.persistent_pair(T, tag, keys.data_ptr(), mouse.data_ptr(), mailbox.data_ptr(), results.data_ptr(), obs.data_ptr(), 1, flags, 0,
                                status.data_ptr(), 2.0)
            t1 = time.perf_counter()
            sync()
            t2 = time.perf_counter()
            tag = (tag + T) % 0xFFFFFF
            if rep < 5:
                continue
            if flags & _lib.TIMER_STOP:
                ev.append(dev.timer_elapsed() * 1e3)
            launch.append((t1 - t0) * 1e6)
            wall.append((t2 - t0) * 1e6)
        assert not status.cpu().numpy().any()
        q = lambda x: {"min": float(np.min(x)), "p50": float(np.median(x)), "p90": float(np.percentile(x, 90))}   # noqa: E731
        out[label] = {"launch_call_us": q(launch), "wall_us": q(wall), "event_us": q(ev) if ev else None,
                      "wall_per_tick_p50": float(np.median(wall)) / T}
    # long region for reference: per-tick event time when launch/sync latency is amortised
    T2 = 720
    keys2 = torch.randint(0, 16, (T2, n), dtype=torch.uint8, device=d)
    mouse2 = torch.rand((T2, n), device=d) * 20 - 10
    for g in (2, 1):
        dev.step_many_dev(T2, _lib.ACT_PACKED, keys2.data_ptr(), mouse2.data_ptr(), _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(),
                          done.data_ptr(), out_stride_ticks=0, use_graph=g)
    torch.cuda.synchronize()
    dev.timer_start()
    for _ in range(5):
        dev.step_many_dev(T2, _lib.ACT_PACKED, keys2.data_ptr(), mouse2.data_ptr(), _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(),
                          done.data_ptr(), out_stride_ticks=0, use_graph=1)
    out["long_region_event_us_per_tick"] = dev.timer_stop() * 1e3 / (5 * T2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
