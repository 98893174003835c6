"""What a pause before a launch costs: a 20-tick `q1env_step_persistent_pair` launch + synchronisation, timed by the host, after the
device has been idle for 0 / 200 us / 2 ms / 20 ms (host spinning) or the host thread slept 2 ms.  Measured on an MI355X: 59 us with
no pause, +5-6 us once the device idled for >= 200 us, +11 us when the host thread slept (its launch call itself takes 13 instead of
6 us); the kernel's HIP-event time is 45 us in every case.  (Explains why the one timed 20-tick region of `bench.py --steps 20` -
wall ~70 us - is slower than the median of a loop over the same launch, 55-59 us.)

    python tools/idle_probe.py
"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from q1physrl_amd import _lib, env as E
from q1physrl_amd.device import DeviceEnv
n, T = 65536, 20
d = torch.device("cuda", 0)
cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
dev = DeviceEnv(cfg, device=0)
keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device=d)
mouse = torch.rand((T, n), device=d) * 20 - 10
obs = torch.empty((n, 6), device=d)
mailbox = torch.zeros((n,), dtype=torch.int64, device=d)
results = torch.zeros((4, n, 2), dtype=torch.int64, device=d)
status = torch.zeros((5,), dtype=torch.int32, device=d)
flags = 1 | _lib.TIMER_START | _lib.TIMER_STOP
tag = 0
def spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us: pass
for label, idle_us, how in (("no idle", 0, None), ("spin 200us", 200, spin), ("spin 2ms", 2000, spin), ("sleep 2ms", 2000, lambda us: time.sleep(us * 1e-6)), ("spin 20ms", 20000, spin), ("no idle again", 0, None)):
    wall, ev, call = [], [], []
    for rep in range(45):
        dev.reset_philox_dev(seed=1, done_only=False)
        torch.cuda.synchronize()
        if how: how(idle_us)
        t0 = time.perf_counter()
        dev.persistent_pair(T, tag, keys.data_ptr(), mouse.data_ptr(), mailbox.data_ptr(), results.data_ptr(), obs.data_ptr(), 1, flags, 0, status.data_ptr(), 2.0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tag = (tag + T) % 0xFFFFFF
        if rep >= 5:
            wall.append((t2 - t0) * 1e6); ev.append(dev.timer_elapsed() * 1e3); call.append((t1 - t0) * 1e6)
    print(f"{label:14s} call p50 {np.median(call):6.1f}  wall p50 {np.median(wall):6.1f}  event p50 {np.median(ev):6.1f}", flush=True)
