"""Time q1env_sample_step alone (the second launch of a sampler tick) on random policy outputs: HIP events, un-profiled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from q1physrl_amd.env import Config
from q1physrl_amd.tensor_env import TensorVectorEnv

PARAMS_YML = dict(action_range=10, allow_jump=True, allow_yaw=True, auto_jump=False, discrete_yaw_steps=-1, fmove_max=800,
                  smove_max=1060, hover=False, initial_yaw_range=(0, 360), key_press_delay=0.3, max_initial_speed=700,
                  smooth_keys=True, speed_reward=False, time_delta=0.013888888888888, time_limit=10, zero_start_prob=0.01)
for n in [int(a) for a in sys.argv[1:]] or [32768, 262144]:
    env = TensorVectorEnv(Config(num_envs=n, **PARAMS_YML), seed=1)
    env.reset()
    d = env.device
    logits = torch.randn((n, 10), device=d)
    keys = torch.empty((n,), dtype=torch.uint8, device=d); mouse = torch.empty((n,), device=d); logp = torch.empty((n,), device=d)
    obs = torch.empty((n, 6), device=d); rew = torch.empty((n,), device=d); done = torch.empty((n,), dtype=torch.uint8, device=d)
    zs = torch.empty((n,), dtype=torch.uint8, device=d)
    ep = torch.zeros((n,), dtype=torch.float64, device=d); part = torch.zeros(((n + 63) // 64, 4), dtype=torch.float64, device=d)
    cnt = torch.zeros((1,), dtype=torch.int64, device=d)

    def go(t):
        env._dev.sample_step_dev(logits.data_ptr(), 10, 7, cnt.data_ptr(), t, False, keys.data_ptr(), mouse.data_ptr(), logp.data_ptr(),
                                 obs.data_ptr(), rew.data_ptr(), done.data_ptr(), zs.data_ptr(), ep.data_ptr(), part.data_ptr())
    for t in range(20):
        go(t)
    torch.cuda.synchronize()
    reps = 400
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(reps):
        go(t)
    e1.record()
    torch.cuda.synchronize()
    print(f"n={n}: q1env_sample_step {e0.elapsed_time(e1) * 1e3 / reps:.2f} us per launch (back-to-back eager launches), "
          f"checksum logp {float(logp.double().sum()):.6f} keys {int(keys.long().sum())} mouse {float(mouse.double().sum()):.6f}")
    env.close()
