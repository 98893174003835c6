"""Throughput of the other BASELINE.json configs on ONE MI355X (bench.py covers configs[1]):
  C3  262 144 envs, full Config (data/params.yml env_config: random starts, key_press_delay, 10 s limit), masked resets on done
  C4  131 072 envs (one GPU's shard of the 1 M-env / 8-GPU config), random-action throughput ceiling
Prints one line per variant.  Inputs resident in HBM; HIP-event timing on the launch stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from q1physrl_amd import _lib
from q1physrl_amd.tensor_env import TensorVectorEnv
from q1physrl_amd.env import Config

PARAMS_YML = dict(action_range=10, allow_jump=True, allow_yaw=True, auto_jump=False, discrete_yaw_steps=-1, fmove_max=800,
                  smove_max=1060, hover=False, initial_yaw_range=(0, 360), key_press_delay=0.3, max_initial_speed=700,
                  smooth_keys=True, speed_reward=False, time_delta=0.013888888888888, time_limit=10, zero_start_prob=0.01)


def timed(env, fn, reps):
    fn(); torch.cuda.synchronize()
    env._dev.timer_start()
    for _ in range(reps):
        fn()
    return env._dev.timer_stop() / reps


def main():
    T = 500
    # ---- C3
    n = 262144
    env = TensorVectorEnv(Config(num_envs=n, **PARAMS_YML), seed=3)
    env.reset()
    keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device="cuda")
    mouse = (torch.rand((T, n), device="cuda") * 20 - 10).contiguous()
    def step_reset():
        for t in range(T):
            env.step_tensor((keys[t], mouse[t]))
            env.reset_done()
    ms = timed(env, step_reset, 2)
    print(f"C3 {n} envs full Config: step + masked reset_done per tick (2 launches/tick): {ms*1e3/T:8.2f} us/tick  {n*T/ms/1e6:8.2f} G env-steps/s")
    # the same ticks as ONE launch each (q1env_step_autoreset: in-kernel Philox reset of finished episodes, RNG counter in device
    # memory), the T launches replayed from a captured graph
    cnt = torch.zeros((1,), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()

    def autoreset_ticks():
        for t in range(T):
            env._dev.step_autoreset_dev(_lib.ACT_PACKED, keys[t].data_ptr(), mouse[t].data_ptr(), 3, env.obs.data_ptr(), env.reward.data_ptr(),
                                        env.done.data_ptr(), env.zero_start.data_ptr(), counter_dev=cnt.data_ptr())
            cnt.add_(1)
    with torch.cuda.stream(side):
        env.use_current_stream()
        autoreset_ticks()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
        env.use_current_stream()
        autoreset_ticks()
    env.use_current_stream()
    torch.cuda.synchronize()
    ms = timed(env, g.replay, 3)
    print(f"C3 {n} envs full Config: step with in-kernel reset, 1 launch/tick (+1 counter op), graph replay: {ms*1e3/T:8.2f} us/tick  {n*T/ms/1e6:8.2f} G env-steps/s")
    ms = timed(env, lambda: env.step_many((keys, mouse), T, auto_reset=True), 3)
    print(f"C3 {n} envs full Config: q1env_step_autoreset_many (1 launch/tick, one counter node per {T} ticks), graph replay: {ms*1e3/T:8.2f} us/tick  {n*T/ms/1e6:8.2f} G env-steps/s")
    def served():
        env.serve_ticks(keys, mouse, sync=False)
    ms = timed(env, served, 3)
    assert not env._srv["status"].cpu().numpy().any()
    print(f"C3 {n} envs full Config: resident tick server + dependent producer (q1env_step_persistent_pair, in-kernel reset): {ms*1e3/T:8.2f} us/tick  {n*T/ms/1e6:8.2f} G env-steps/s")
    obs = torch.empty((T, n, 6), dtype=torch.float32, device="cuda"); rew = torch.empty((T, n), device="cuda"); done = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    def fused():
        env._dev.rollout_dev(T, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 3, _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), True, 0)
    ms = timed(env, fused, 4)
    print(f"C3 {n} envs full Config: fused rollout, in-kernel reset on done, per-tick outputs:  {ms*1e3/T:8.2f} us/tick  {n*T/ms/1e6:8.2f} G env-steps/s")
    env.close()
    # ---- C4 shard
    n = 131072
    cfg = Config(**{**Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    env = TensorVectorEnv(cfg, seed=4)
    env.reset()
    keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device="cuda")
    mouse = (torch.rand((T, n), device="cuda") * 20 - 10)
    obs = torch.empty((T, n, 6), dtype=torch.float32, device="cuda"); rew = torch.empty((T, n), device="cuda"); done = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    variants = {
        "resident tick server + dependent producer (q1env_step_persistent_pair)": lambda: env.serve_ticks(keys.contiguous(), mouse.contiguous(), sync=False),
        "per-tick step kernel (hipGraph), packed actions from HBM": lambda: env._dev.step_many_dev(T, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), _lib.OBS_F32, env.obs.data_ptr(), env.reward.data_ptr(), env.done.data_ptr(), 0, True),
        "fused rollout, packed actions from HBM, per-tick outputs": lambda: env._dev.rollout_dev(T, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), False, 0),
        "fused rollout, on-device Philox actions, per-tick outputs": lambda: env._dev.rollout_dev(T, _lib.ACT_RANDOM, 0, 0, 7, _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), False, 0),
        "fused rollout, on-device Philox actions, no per-tick outputs": lambda: env._dev.rollout_dev(T, _lib.ACT_RANDOM, 0, 0, 7, _lib.OBS_F32, 0, 0, 0, False, 0),
    }
    for name, fn in variants.items():
        ms = timed(env, fn, 4)
        print(f"C4 {n} envs/GPU zero-start: {name}: {ms*1e3/T:8.2f} us/tick  {n*T/ms/1e6:8.2f} G env-steps/s")
    env.close()


if __name__ == "__main__":
    main()
