"""GPU parity of the device-pointer entry points (q1env_step with every action layout, q1env_step_many with and
without hipGraph, the fused q1env_rollout kernel, the Philox device RNG for actions and resets, in-kernel
auto-reset, shard-count invariance) against the oracle.  Same bar as test_hip_parity.py: integers bit-exact,
floats <= 1e-5 relative and >= 99.9 % bit-identical (observed: bit-identical)."""
import dataclasses

import numpy as np
import pytest

from oracle import np_oracle as O
from oracle import philox as PH

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5


def torch_mod():
    import torch
    return torch


def inject(ora, tenv):
    """Copy the oracle's state into the device env (golden-state injection through q1env_set_state_host)."""
    k = ora.cfg.num_keys
    n = ora.n
    lk = np.full((n, 4), -np.float64(ora.cfg.key_press_delay))
    lk[:, :k] = ora.dec["last_press"]
    keys = np.zeros(n, dtype=np.uint8)
    for j in range(k):
        keys |= ora.dec["last_keys"][:, j].astype(np.uint8) << (3 + j)
    flags = (ora.st["on_ground"].astype(np.uint8) | (ora.st["jump_released"].astype(np.uint8) << 1)
             | (ora.zero_start.astype(np.uint8) << 2) | keys)
    tenv.set_state(vel_x=ora.st["vel"][:, 0], vel_y=ora.st["vel"][:, 1], vel_z=ora.st["vel"][:, 2],
                   pos_x=np.zeros(n), pos_y=np.zeros(n), z_pos=ora.st["z_pos"], yaw=ora.yaw, time_remaining=ora.t_rem,
                   last_key_press_time=lk, flags=flags)


def close(a, b, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    e = np.abs(a - b) / np.maximum(np.abs(b), 1.0)
    assert float(e.max()) <= REL_TOL, (what, float(e.max()))


def same_bits_frac(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape
    u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return float(np.mean(a.view(u) == b.view(u)))


def make_pair(n, seed, **over):
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    cfg = O.OracleConfig.get_default(num_envs=n, **over)
    np.random.seed(seed)
    ora = O.OracleVectorEnv(cfg)
    tenv = TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=seed)
    inject(ora, tenv)
    return cfg, ora, tenv


def gen_actions(rng, ticks, n, cfg):
    keys = rng.integers(0, 1 << cfg.num_keys, size=(ticks, n), dtype=np.uint8)
    mouse = rng.uniform(-float(cfg.action_range), float(cfg.action_range), size=(ticks, n)).astype(np.float32)
    rows = np.concatenate([((keys[:, :, None] >> np.arange(cfg.num_keys)[None, None, :]) & 1).astype(np.float64),
                           mouse[:, :, None].astype(np.float64)], axis=2)
    return keys, mouse, rows


@pytest.mark.parametrize("fmt", ["packed", "f32rows", "f64rows"])
def test_step_tensor_action_layouts(fmt):
    torch = torch_mod()
    n, ticks = 1000, 120
    cfg, ora, tenv = make_pair(n, 11, zero_start_prob=0.3)
    rng = np.random.default_rng(5)
    keys, mouse, rows = gen_actions(rng, ticks, n, cfg)
    for t in range(ticks):
        o1, r1, d1, z1 = ora.vector_step(rows[t])
        if fmt == "packed":
            act = (torch.from_numpy(keys[t]).cuda(), torch.from_numpy(mouse[t]).cuda())
        elif fmt == "f32rows":
            act = torch.from_numpy(rows[t].astype(np.float32)).cuda()
        else:
            act = torch.from_numpy(rows[t]).cuda()
        obs, rew, done = tenv.step_tensor(act)
        assert obs.dtype == torch.float32 and obs.shape == (n, 6)
        assert np.array_equal(obs.cpu().numpy(), o1.astype(np.float32)), t       # f32 obs IS the f64 obs rounded
        assert np.array_equal(rew.cpu().numpy(), r1) and np.array_equal(done.cpu().numpy().astype(bool), d1)
        assert np.array_equal(tenv.zero_start.cpu().numpy().astype(bool), z1)
    st = tenv.get_state()
    assert np.array_equal(st["vel_x"], ora.st["vel"][:, 0]) and np.array_equal(st["z_pos"], ora.st["z_pos"])
    tenv.close()


@pytest.mark.parametrize("mode", ["many_graph", "many_eager", "rollout_f32", "rollout_f64", "rollout_prepared"])
def test_multi_tick_entry_points(mode):
    torch = torch_mod()
    n, ticks = 777, 150            # ragged: 777 = 12 waves + 9 lanes
    cfg, ora, tenv = make_pair(n, 21, zero_start_prob=0.2, time_limit=4.0)
    rng = np.random.default_rng(8)
    keys, mouse, rows = gen_actions(rng, ticks, n, cfg)
    ref_obs, ref_rew, ref_done = [], [], []
    for t in range(ticks):
        o1, r1, d1, _ = ora.vector_step(rows[t])
        ref_obs.append(o1); ref_rew.append(r1); ref_done.append(d1)
    ref_obs, ref_rew, ref_done = np.stack(ref_obs), np.stack(ref_rew), np.stack(ref_done)
    act = (torch.from_numpy(keys).cuda(), torch.from_numpy(mouse).cuda())
    if mode.startswith("many"):
        obs, rew, done = tenv.step_many(act, ticks, outputs=True, use_graph=(mode == "many_graph"))
    elif mode == "rollout_prepared":                       # DeviceEnv.prepare_rollout: the same call with its arguments converted once
        from q1physrl_amd import _lib
        obs = torch.empty((ticks, n, 6), dtype=torch.float32, device="cuda")
        rew = torch.empty((ticks, n), dtype=torch.float32, device="cuda")
        done = torch.empty((ticks, n), dtype=torch.uint8, device="cuda")
        half = ticks // 2
        calls = [tenv._dev.prepare_rollout(half, _lib.ACT_PACKED, act[0].data_ptr(), act[1].data_ptr(), 0, _lib.OBS_F32, obs.data_ptr(),
                                           rew.data_ptr(), done.data_ptr()),
                 tenv._dev.prepare_rollout(ticks - half, _lib.ACT_PACKED, act[0][half:].data_ptr(), act[1][half:].data_ptr(), 0, _lib.OBS_F32,
                                           obs[half:].data_ptr(), rew[half:].data_ptr(), done[half:].data_ptr())]
        for c in calls:
            c()
    else:
        obs, rew, done = tenv.rollout(ticks, act, outputs=True, obs_dtype=torch.float32 if mode == "rollout_f32" else torch.float64)
    torch.cuda.synchronize()
    want_obs = ref_obs.astype(np.float32) if obs.dtype == torch.float32 else ref_obs
    assert np.array_equal(obs.cpu().numpy(), want_obs)
    assert np.array_equal(rew.cpu().numpy(), ref_rew) and np.array_equal(done.cpu().numpy().astype(bool), ref_done)
    st = tenv.get_state()
    assert np.array_equal(st["yaw"], ora.yaw) and np.array_equal(st["time_remaining"], ora.t_rem)
    assert np.array_equal(st["last_key_press_time"][:, :4], ora.dec["last_press"])
    tenv.close()


def test_random_action_rollout_matches_philox_restatement():
    torch = torch_mod()
    n, ticks, seed = 512, 100, 1234
    cfg, ora, tenv = make_pair(n, seed, zero_start_prob=1.0)
    obs, rew, done = tenv.rollout(ticks, None, outputs=True, obs_dtype=torch.float64)
    genv = np.arange(n, dtype=np.uint64)
    for t in range(ticks):
        a = PH.random_actions(cfg, seed, genv, t)          # handle's tick counter starts at 0
        o1, r1, d1, _ = ora.vector_step(a)
        assert np.array_equal(obs[t].cpu().numpy(), o1), t
        assert np.array_equal(rew[t].cpu().numpy(), r1)
    tenv.close()


def test_reset_philox_matches_restatement_and_distribution():
    from scipy import stats
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    n, seed = 200_000, 77
    cfg = O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.25)
    tenv = TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=seed, env_index_base=5_000_000_000)   # > 2^32: 64-bit env index
    tenv.reset()
    st = tenv.get_state()
    zs, yaw, tm, sp, an = PH.reset_draws(cfg, seed, np.arange(n, dtype=np.uint64) + np.uint64(5_000_000_000), 0)
    flags = st["flags"]
    assert np.array_equal((flags & 4) != 0, zs) and np.all((flags & 2) != 0) and not np.any(flags & 1)
    assert np.array_equal(st["yaw"], np.where(zs, 90.0, yaw)) and np.array_equal(st["time_remaining"], np.where(zs, cfg.time_limit, tm))
    speed = np.where(zs, 0.0, sp)
    close(st["vel_x"], (speed * np.cos(an)).astype(np.float32), "vel_x")
    assert same_bits_frac(st["vel_x"], (speed * np.cos(an)).astype(np.float32)) > 0.999
    assert same_bits_frac(st["vel_y"], (speed * np.sin(an)).astype(np.float32)) > 0.999
    assert np.all(st["vel_z"] == np.float32(-12)) and np.all(st["z_pos"] == np.float64(np.float32(32.843201)))
    assert np.all(st["last_key_press_time"] == -cfg.key_press_delay)
    # the reference's distributions: zero-start fraction, yaw ~ U(0,360), and the one-argument uniform quirk:
    # time in (1, 10], speed in (1, 700], angle in (1, 2 pi]
    nz = ~zs
    assert abs(zs.mean() - 0.25) < 4 * np.sqrt(0.25 * 0.75 / n)
    assert stats.kstest(st["yaw"][nz] / 360.0, "uniform").pvalue > 1e-4
    assert stats.kstest((st["time_remaining"][nz] - 1.0) / 9.0, "uniform").pvalue > 1e-4
    spd = np.hypot(st["vel_x"][nz].astype(np.float64), st["vel_y"][nz].astype(np.float64))
    assert stats.kstest((spd - 1.0) / 699.0, "uniform").pvalue > 1e-4 and spd.min() > 0.999 and spd.max() <= 700.001
    ang = np.mod(np.arctan2(st["vel_y"][nz].astype(np.float64), st["vel_x"][nz].astype(np.float64)), 2 * np.pi)
    assert ang.min() > 0.999 and stats.kstest((ang - 1.0) / (2 * np.pi - 1.0), "uniform").pvalue > 1e-4
    tenv.close()


def test_auto_reset_rollout_equals_step_then_reset_done():
    torch = torch_mod()
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    n, ticks, seed = 300, 400, 5
    cfg = Config(**O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.3, time_limit=1.5).__dict__)
    rng = np.random.default_rng(3)
    keys = torch.from_numpy(rng.integers(0, 16, size=(ticks, n), dtype=np.uint8)).cuda()
    mouse = torch.from_numpy(rng.uniform(-10, 10, size=(ticks, n)).astype(np.float32)).cuda()
    a = TensorVectorEnv(cfg, seed=seed)
    b = TensorVectorEnv(cfg, seed=seed)
    a.reset(); b.reset()
    ret = torch.zeros((n,), dtype=torch.float64, device="cuda")
    obs_a, rew_a, done_a = a.rollout(ticks, (keys, mouse), outputs=True, auto_reset=True, return_sum=ret)
    tot = np.zeros(n)
    n_resets = 0
    for t in range(ticks):
        obs, rew, done = b.step_tensor((keys[t], mouse[t]))
        assert torch.equal(obs, obs_a[t]) and torch.equal(rew, rew_a[t]) and torch.equal(done, done_a[t]), t
        tot += rew.cpu().numpy().astype(np.float64)
        n_resets += int(done.sum())
        b.reset_done()
    assert n_resets > n            # every env finished at least one episode
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.allclose(ret.cpu().numpy(), tot, rtol=1e-12, atol=1e-9)
    a.close(); b.close()


def test_shard_count_invariance():
    """Splitting the batch over 1, 2 or 3 handles (env_index_base = shard start) gives identical envs: the counter
    RNG is keyed by the global env index, and no kernel has a cross-env term."""
    torch = torch_mod()
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    from q1physrl_amd.sharding import shard_plan
    n, ticks, seed = 1000, 200, 9
    base = O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.2, time_limit=1.0)
    full = TensorVectorEnv(Config(**base.__dict__), seed=seed)
    full.reset()
    full.rollout(ticks, None, outputs=False, auto_reset=True)
    ref = full.get_state()
    for world in (2, 3):
        parts = []
        for start, count in shard_plan(n, world):
            e = TensorVectorEnv(Config(**dataclasses.replace(base, num_envs=count).__dict__), seed=seed, env_index_base=start)
            e.reset()
            e.rollout(ticks, None, outputs=False, auto_reset=True)
            parts.append(e.get_state())
            e.close()
        for k in ref:
            assert np.array_equal(np.concatenate([p[k] for p in parts], axis=0), ref[k]), (world, k)
    full.close()


def test_config3_full_size_shard_invariance_262144_envs_2000_ticks():
    """BASELINE.json configs[2] at FULL size (262 144 envs, the reference run's params.yml env_config with random starts,
    2 000 ticks with resets on done, on-device random actions): the whole batch on one handle equals 8 shards of 32 768 envs
    with env_index_base = shard start (the multi-GPU partition), env for env, including the accumulated episode returns."""
    torch = torch_mod()
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    from q1physrl_amd.sharding import shard_plan
    n, ticks, seed = 262144, 2000, 3
    params_yml = dict(action_range=10, allow_jump=True, allow_yaw=True, auto_jump=False, discrete_yaw_steps=-1, fmove_max=800,
                      smove_max=1060, hover=False, initial_yaw_range=(0, 360), key_press_delay=0.3, max_initial_speed=700,
                      smooth_keys=True, speed_reward=False, time_delta=0.013888888888888, time_limit=10, zero_start_prob=0.01)

    def run(count, start):
        e = TensorVectorEnv(Config(num_envs=count, **params_yml), seed=seed, env_index_base=start)
        e.reset()
        ret = torch.zeros((count,), dtype=torch.float64, device="cuda")
        e.rollout(ticks, None, outputs=False, auto_reset=True, return_sum=ret)
        torch.cuda.synchronize()
        st = e.get_state()
        e.close()
        return st, ret.cpu().numpy()
    ref, ref_ret = run(n, 0)
    parts = [run(count, start) for start, count in shard_plan(n, 8)]
    for k in ref:
        assert np.array_equal(np.concatenate([p[0][k] for p in parts], axis=0), ref[k]), k
    assert np.array_equal(np.concatenate([p[1] for p in parts]), ref_ret)
    assert ref["time_remaining"].min() >= -0.02 and ref["time_remaining"].max() <= 10.0          # every env kept being reset
    assert np.isfinite(ref_ret).all() and np.abs(ref_ret).max() > 100.0
    assert len(np.unique(ref["time_remaining"])) > 1000                                            # random starts: episodes out of phase


def test_large_batch_16m_envs_indexing():
    """16 777 216 envs on one handle (1.4 GB of state; 32-bit indexing, 65 536-block grids): 40 ticks of a 4 096-env action
    pattern tiled 4 096 times - every replica must equal the first tile (checked on the device through the zero-copy state
    views) and the first tile must equal a 4 096-env handle driven with the same actions."""
    torch = torch_mod()
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    n, base, ticks = 1 << 24, 4096, 40
    cfg = O.OracleConfig.get_default(num_envs=n, zero_start_prob=1.0)
    g = torch.Generator(device="cuda").manual_seed(5)
    keys_b = torch.randint(0, 16, (ticks, base), dtype=torch.uint8, device="cuda", generator=g)
    mouse_b = (torch.rand((ticks, base), device="cuda", generator=g) * 20.16 - 10.08).contiguous()
    big = TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=1)
    big.reset()
    for t in range(ticks):                                      # per-tick kernel: one (n,) action row at a time
        big.step_tensor((keys_b[t].repeat(n // base), mouse_b[t].repeat(n // base)))
    small = TensorVectorEnv(Config(**dataclasses.replace(cfg, num_envs=base).__dict__), device=0, seed=1)
    small.reset()
    small.step_many((keys_b, mouse_b), ticks, outputs=False, use_graph=False)
    torch.cuda.synchronize()
    sb, ss = big.state_tensors(), small.state_tensors()
    for k in ("vel_x", "vel_y", "vel_z", "pos_x", "pos_y", "z_pos", "yaw", "time_remaining", "flags"):
        tiles = sb[k].view(n // base, base)
        assert bool((tiles == ss[k].view(1, base)).all()), k
    lk = sb["last_key_press_time"].view(4, n // base, base)
    assert bool((lk == ss["last_key_press_time"].view(4, 1, base)).all())
    assert float(sb["vel_y"].abs().max()) > 10.0                 # the envs really moved
    big.close(); small.close()


def test_state_tensor_views_alias_device_state():
    torch = torch_mod()
    cfg, ora, tenv = make_pair(64, 2, zero_start_prob=1.0)
    views = tenv.state_tensors()
    assert views["yaw"].dtype == torch.float64 and views["last_key_press_time"].shape == (4, 64)
    views["vel_x"].fill_(3.0)
    torch.cuda.synchronize()
    assert np.all(tenv.get_state()["vel_x"] == 3.0)
    tenv.close()


VARIANTS = {
    "auto_jump": dict(auto_jump=True), "no_jump": dict(allow_jump=False), "hover": dict(hover=True),
    "speed_reward": dict(speed_reward=True), "no_smooth": dict(smooth_keys=False), "delay0": dict(key_press_delay=0.0),
    "discrete_yaw": dict(discrete_yaw_steps=7), "no_yaw": dict(allow_yaw=False),
    "dataclass_defaults": dict(time_delta=0.014, time_limit=5, smove_max=700., smooth_keys=False),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_generic_kernels_on_config_variants(name):
    """Non-default Configs run the SPEC=false kernels (run-time wave-uniform branches).  Device-pointer entry points
    (float32 obs, float32/float64 action rows, per-tick step and the fused rollout) against the oracle, bit-exact."""
    torch = torch_mod()
    n, ticks = 500, 90
    over = dict(VARIANTS[name], zero_start_prob=0.3)
    cfg, ora, tenv = make_pair(n, 31, **over)
    cfg2, ora2, tenv2 = make_pair(n, 31, **over)
    rng = np.random.default_rng(17)
    k = cfg.num_keys
    cols = [(rng.random((ticks, n, k)) < 0.5).astype(np.float64)]
    if cfg.allow_yaw:
        if cfg.discrete_yaw_steps == -1:
            cols.append(rng.uniform(-10, 10, (ticks, n, 1)).astype(np.float32).astype(np.float64))
        else:
            cols.append(rng.integers(0, 2 * cfg.discrete_yaw_steps + 1, (ticks, n, 1)).astype(np.float64))
    rows = np.concatenate(cols, axis=2)
    ref_obs, ref_rew, ref_done = [], [], []
    for t in range(ticks):
        o1, r1, d1, _ = ora.vector_step(rows[t])
        ref_obs.append(o1.astype(np.float32)); ref_rew.append(r1); ref_done.append(d1)
        act = torch.from_numpy(rows[t].astype(np.float32) if t % 2 else rows[t]).cuda()     # alternate f32 / f64 rows
        obs, rew, done = tenv.step_tensor(act)
        assert np.array_equal(obs.cpu().numpy(), ref_obs[-1]), (name, t)
        assert np.array_equal(rew.cpu().numpy(), r1) and np.array_equal(done.cpu().numpy().astype(bool), d1)
    # the same ticks through the fused rollout kernel (generic instantiation), float32 rows tick-major
    obs, rew, done = tenv2.rollout(ticks, torch.from_numpy(rows.astype(np.float32)).cuda(), outputs=True)
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), np.stack(ref_obs)) and np.array_equal(rew.cpu().numpy(), np.stack(ref_rew))
    assert np.array_equal(done.cpu().numpy().astype(bool), np.stack(ref_done))
    s1, s2 = tenv.get_state(), tenv2.get_state()
    for key in ("vel_x", "vel_y", "vel_z", "z_pos", "yaw", "time_remaining", "flags", "last_key_press_time"):
        assert np.array_equal(s1[key], s2[key]), (name, key)
    assert np.array_equal(s1["yaw"], ora.yaw) and np.array_equal(s1["vel_x"], ora.st["vel"][:, 0])
    tenv.close(); tenv2.close()


def test_full_size_rollout_parity_65536_envs_720_ticks():
    """BASELINE.json configs[1] at FULL size: 65 536 envs x 720 ticks (10 s), zero-start, persistent random actions.
    The fused rollout kernel against the multi-threaded C oracle on identical actions: every per-tick reward / done / obs
    and the final state, bit-exact; max |pos - ref| (x, y distance integrals and z) reported and required to be < 1e-5."""
    torch = torch_mod()
    from oracle import c_oracle as CO
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    import os
    n, ticks = 65536, 720
    cfg = O.OracleConfig.get_default(num_envs=n, zero_start_prob=1.0)
    np.random.seed(1)
    ora = CO.COracleVectorEnv(cfg, threads=min(32, os.cpu_count() or 1))
    tenv = TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=1)
    inject(ora, tenv)
    rng = np.random.default_rng(42)
    keys = np.empty((ticks, n), np.uint8)
    cur = rng.integers(0, 16, n, dtype=np.uint8)
    for t in range(ticks):
        flip = np.zeros(n, np.uint8)
        for k in range(4):
            flip |= (rng.random(n) < 0.05).astype(np.uint8) << k
        cur = cur ^ flip
        keys[t] = cur
    mouse = rng.uniform(-10.08, 10.08, (ticks, n)).astype(np.float32)
    obs, rew, done = tenv.rollout(ticks, (torch.from_numpy(keys).cuda(), torch.from_numpy(mouse).cuda()), outputs=True)
    torch.cuda.synchronize()
    rew_g, done_g = rew.cpu().numpy(), done.cpu().numpy().astype(bool)
    dist = np.zeros((n, 2))
    bits = np.arange(4)[None, :]
    dt = cfg.time_delta
    for t in range(ticks):
        a = np.concatenate([((keys[t][:, None] >> bits) & 1).astype(np.float64), mouse[t][:, None].astype(np.float64)], axis=1)
        o, r, d, _ = ora.vector_step(a)
        assert np.array_equal(r, rew_g[t]) and np.array_equal(d, done_g[t]), t
        if t % 60 == 59 or t == ticks - 1:
            assert np.array_equal(o.astype(np.float32), obs[t].cpu().numpy()), t
        dist += dt * ora.st["vel"][:, :2].astype(np.float64)
    st = tenv.get_state()
    err_xy = max(np.abs(st["pos_x"] - dist[:, 0]).max(), np.abs(st["pos_y"] - dist[:, 1]).max())
    err_z = np.abs(st["z_pos"] - ora.st["z_pos"]).max()
    err_v = max(np.abs(st["vel_x"] - ora.st["vel"][:, 0]).max(), np.abs(st["vel_y"] - ora.st["vel"][:, 1]).max())
    print(f"65536 x 720: max|pos_xy - ref| = {err_xy:.3e}, max|z - ref| = {err_z:.3e}, max|vel - ref| = {err_v:.3e}, "
          f"max |y| travelled = {np.abs(dist[:, 1]).max():.1f}")
    assert err_xy < 1e-5 and err_z == 0.0 and err_v == 0.0
    assert np.array_equal(st["yaw"], ora.yaw) and np.array_equal((st["flags"] & 1) != 0, ora.st["on_ground"])
    assert done_g[-1].all() and not done_g[:-1].any()
    tenv.close()


def test_one_million_envs_replication_and_kernel_agreement():
    """BASELINE.json configs[3] size (1 048 576 envs, 720 ticks, zero start) through size-independent properties: the action
    tensor is a 4 096-env pattern tiled 256 times, so (a) every replica must end bit-identical to the first 4 096 envs,
    (b) those equal the C oracle driven with the same 4 096 action columns, and (c) the per-tick step kernels (hipGraph)
    and the fused rollout kernel must agree on all 1 048 576 final states.  Also a checksum of checksums of the state."""
    torch = torch_mod()
    from oracle import c_oracle as CO
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    import os
    n, base, ticks = 1 << 20, 4096, 720
    cfg = O.OracleConfig.get_default(num_envs=n, zero_start_prob=1.0)
    rng = np.random.default_rng(7)
    keys_b = rng.integers(0, 16, (ticks, base), dtype=np.uint8)
    keys_b[1:] = np.where(rng.random((ticks - 1, base)) < 0.8, keys_b[:-1], keys_b[1:])       # some persistence
    mouse_b = rng.uniform(-10.08, 10.08, (ticks, base)).astype(np.float32)
    keys = torch.from_numpy(keys_b).cuda().repeat(1, n // base).contiguous()
    mouse = torch.from_numpy(mouse_b).cuda().repeat(1, n // base).contiguous()
    finals = []
    for mode in ("rollout", "step_many"):
        tenv = TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=1)
        tenv.reset()                                            # zero_start_prob = 1: identical start line for every env
        if mode == "rollout":
            tenv.rollout(ticks, (keys, mouse), outputs=False)
        else:
            tenv.step_many((keys, mouse), ticks, outputs=False, use_graph=True)
        torch.cuda.synchronize()
        finals.append(tenv.get_state())
        tenv.close()
    a, b = finals
    for k in a:
        assert np.array_equal(a[k], b[k]), k                    # (c) both kernels, every env
    for k in ("vel_x", "vel_y", "vel_z", "pos_x", "pos_y", "z_pos", "yaw", "time_remaining", "flags"):
        tiles = a[k].reshape(n // base, base)
        assert np.array_equal(tiles, np.broadcast_to(tiles[0], tiles.shape)), k              # (a) replicas
    lk = a["last_key_press_time"].reshape(n // base, base, 4)
    assert np.array_equal(lk, np.broadcast_to(lk[0], lk.shape))
    # (b) the first 4 096 envs against the C oracle
    ocfg = O.OracleConfig.get_default(num_envs=base, zero_start_prob=1.0)
    np.random.seed(1)
    ora = CO.COracleVectorEnv(ocfg, threads=min(16, os.cpu_count() or 1))
    bits = np.arange(4)[None, :]
    dist = np.zeros((base, 2))
    for t in range(ticks):
        act = np.concatenate([((keys_b[t][:, None] >> bits) & 1).astype(np.float64), mouse_b[t][:, None].astype(np.float64)], axis=1)
        ora.vector_step(act)
        dist += ocfg.time_delta * ora.st["vel"][:, :2].astype(np.float64)
    assert np.array_equal(a["vel_x"][:base], ora.st["vel"][:, 0]) and np.array_equal(a["vel_y"][:base], ora.st["vel"][:, 1])
    assert np.array_equal(a["vel_z"][:base], ora.st["vel"][:, 2]) and np.array_equal(a["z_pos"][:base], ora.st["z_pos"])
    assert np.array_equal(a["yaw"][:base], ora.yaw) and np.array_equal((a["flags"][:base] & 1) != 0, ora.st["on_ground"])
    assert max(np.abs(a["pos_x"][:base] - dist[:, 0]).max(), np.abs(a["pos_y"][:base] - dist[:, 1]).max()) < 1e-5
    # checksum of checksums: per-tile uint64 sums of the raw state words are all equal, and their total is 256 x the first
    words = np.concatenate([a[k].view(np.uint32).reshape(n // base, -1).astype(np.uint64).sum(1, keepdims=True)
                            for k in ("vel_x", "vel_y", "vel_z", "pos_y", "z_pos", "yaw")], axis=1)
    assert (words == words[0]).all() and int(words.sum()) == int(words[0].sum()) * (n // base)


@pytest.mark.parametrize("variant", ["default", "auto_jump"])
def test_step_autoreset_equals_step_then_reset_done(variant):
    """q1env_step_autoreset (one launch) against q1env_step + q1env_reset_philox(done_only) (two launches): same rewards,
    dones, zero_start flags, observations and final state, SPEC and generic instantiations."""
    torch = torch_mod()
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    n, ticks, seed = 777, 300, 5
    over = dict(zero_start_prob=0.3, time_limit=1.0, **({"auto_jump": True} if variant == "auto_jump" else {}))
    cfg = Config(**O.OracleConfig.get_default(num_envs=n, **over).__dict__)
    rng = np.random.default_rng(3)
    nk = 3 if variant == "auto_jump" else 4
    keys = torch.from_numpy(rng.integers(0, 1 << nk, size=(ticks, n), dtype=np.uint8)).cuda()
    mouse = torch.from_numpy(rng.uniform(-10, 10, size=(ticks, n)).astype(np.float32)).cuda()
    a, b = TensorVectorEnv(cfg, seed=seed), TensorVectorEnv(cfg, seed=seed)
    a.reset(); b.reset()
    resets = 0
    for t in range(ticks):
        oa, ra, da = a.step_autoreset((keys[t], mouse[t]))
        za = a.zero_start.clone()
        ob, rb, db = b.step_tensor((keys[t], mouse[t]))
        zb = b.zero_start.clone()
        ob = b.reset_done()
        assert torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(za, zb) and torch.equal(oa, ob), t
        resets += int(da.sum())
    assert resets > n
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    a.close(); b.close()


@pytest.mark.parametrize("n,use_graph", [(4096 + 19, True), (4096 + 19, False), (65536, True)])
def test_step_autoreset_many_equals_single_launches(n, use_graph):
    """q1env_step_autoreset_many: T auto-reset ticks + one counter node, replayed from a cached hipGraph.  Two replays (the second
    draws fresh reset randomness: the Philox counter lives on the device and the graph advances it) equal 2 T single
    q1env_step_autoreset launches driven with the same device counter, tick-major outputs included."""
    torch = torch_mod()
    from q1physrl_amd import _lib
    from q1physrl_amd.tensor_env import TensorVectorEnv
    from q1physrl_amd.env import Config
    T = 150
    cfg = O.OracleConfig.get_default(num_envs=n, zero_start_prob=0.3, time_limit=0.6)
    a = TensorVectorEnv(Config(**cfg.__dict__), seed=11)
    b = TensorVectorEnv(Config(**cfg.__dict__), seed=11)
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device="cuda", generator=g)
    mouse = ((torch.rand((T, n), device="cuda", generator=g) * 2 - 1) * 10).contiguous()
    cnt_b = torch.zeros((1,), dtype=torch.int64, device="cuda")
    outs = []
    for rep in range(2):
        obs, rew, done = a.step_many((keys, mouse), T, outputs=True, use_graph=use_graph, auto_reset=True)
        outs.append((obs.clone(), rew.clone(), done.clone()))
    assert int(a.reset_counter.item()) == 2 * T
    for rep in range(2):
        for t in range(T):
            b._dev.step_autoreset_dev(_lib.ACT_PACKED, keys[t].data_ptr(), mouse[t].data_ptr(), 11, b.obs.data_ptr(), b.reward.data_ptr(),
                                      b.done.data_ptr(), b.zero_start.data_ptr(), counter_dev=cnt_b.data_ptr())
            cnt_b.add_(1)
            assert torch.equal(outs[rep][0][t], b.obs) and torch.equal(outs[rep][1][t], b.reward) and torch.equal(outs[rep][2][t], b.done), (rep, t)
    torch.cuda.synchronize()
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert outs[0][2].sum() > n and not torch.equal(outs[0][0][-1], outs[1][0][-1])     # episodes ended; the replays differ
    a.close(); b.close()
