"""Time the fused rollout kernel and the per-tick step kernel (packed actions, f32 obs out) at 65 536 and 1 M envs for one or
more builds of libq1env.so given on the command line (used for A/B comparisons of kernel variants)."""
import sys, os, shutil, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
from q1physrl_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from q1physrl_amd.tensor_env import TensorVectorEnv
from q1physrl_amd.env import Config
for n in (65536, 1048576):
    cfg = Config(**{**Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    e = TensorVectorEnv(cfg, seed=1)
    T = 240
    keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device="cuda")
    mouse = (torch.rand((T, n), device="cuda") * 20 - 10)
    obs = torch.empty((T, n, 6), dtype=torch.float32, device="cuda"); rew = torch.empty((T, n), device="cuda"); done = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    def run():
        e._dev.rollout_dev(T, 2, keys.data_ptr(), mouse.data_ptr(), 0, 1, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), False, 0)
    run(); torch.cuda.synchronize()
    e._dev.timer_start()
    for _ in range(5): run()
    ms = e._dev.timer_stop()
    print(f"  n={n:8d} rollout {ms*1e3/(5*T):8.3f} us/tick", end="")
    # single tick kernel
    def run1():
        e._dev.step_many_dev(T, 2, keys.data_ptr(), mouse.data_ptr(), 1, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), 0, True)
    run1(); torch.cuda.synchronize()
    e._dev.timer_start()
    for _ in range(5): run1()
    ms = e._dev.timer_stop()
    print(f"   step {ms*1e3/(5*T):8.3f} us/tick")
    e.close()
''' % ROOT
for v in sys.argv[1:]:
    print("variant", v)
    subprocess.run([sys.executable, "-c", code, v], check=False)
