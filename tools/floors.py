"""Launch / memory floors on the GPU box: calib copy kernel (85 B in + 85 B out per env, no math) vs step kernel, per size."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from q1physrl_amd import _lib
from q1physrl_amd.tensor_env import TensorVectorEnv
from q1physrl_amd.env import Config
for n in (4096, 65536, 262144, 1048576, 4194304, 16777216):
    cfg = Config(**{**Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    e = TensorVectorEnv(cfg, seed=1)
    T = 120
    keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device="cuda")
    mouse = (torch.rand((T, n), device="cuda") * 20 - 10)
    e._dev.calibrate_traffic(T)
    e._dev.timer_start(); e._dev.calibrate_traffic(T * 3); ms_copy = e._dev.timer_stop()
    def run1():
        e._dev.step_many_dev(T, 2, keys.data_ptr(), mouse.data_ptr(), 1, e.obs.data_ptr(), e.reward.data_ptr(), e.done.data_ptr(), 0, True)
    run1(); torch.cuda.synchronize()
    e._dev.timer_start()
    for _ in range(3): run1()
    ms = e._dev.timer_stop()
    us_copy, us_step = ms_copy * 1e3 / (3 * T), ms * 1e3 / (3 * T)
    print(f"n={n:9d}  copy(170 B/env) {us_copy:9.2f} us = {170*n/us_copy/1e6:7.2f} TB/s   step(204 B/env) {us_step:9.2f} us = {204*n/us_step/1e6:7.2f} TB/s  {n/us_step/1e3:7.2f} G env-steps/s")
    e.close()
