import ctypes as C, time, numpy as np, sys
sys.path.insert(0, '/root/repo')
from q1physrl_amd import _lib, env as E
from q1physrl_amd.device import DeviceEnv
for n in (1, 100, 1024, 4096):
    cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n})
    dev = DeviceEnv(cfg, device=0)
    lib = dev._lib
    a = np.zeros((n, 5)); obs = np.empty((n, 6)); rew = np.empty(n, np.float32); done = np.empty(n, np.uint8); zs = np.empty(n, np.uint8)
    args = (dev._h, C.c_int(0), C.c_void_p(a.ctypes.data), None, C.c_int(0), C.c_void_p(obs.ctypes.data), C.c_void_p(rew.ctypes.data), C.c_void_p(done.ctypes.data), C.c_void_p(zs.ctypes.data))
    for _ in range(200): lib.q1env_step_host(*args)
    t0 = time.perf_counter()
    K = 2000
    for _ in range(K): lib.q1env_step_host(*args)
    dt = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K): dev.step_host(a)
    dt2 = (time.perf_counter() - t0) / K
    print(f"n={n:5d}: q1env_step_host through ctypes (prepared args) {dt*1e6:6.2f} us;  DeviceEnv.step_host {dt2*1e6:6.2f} us")
    dev.close()
