"""GPU: the resident tick server (q1env_step_persistent_start / _drive) - bit-identical to per-tick q1env_step_autoreset calls and
to the NumPy oracle, across launches, with ragged sizes and in-kernel resets; and its failure mode: without a producer the
server times out, reports it, stores the state it had and the device stays usable (it must never hang)."""
import os
import time

import numpy as np
import pytest

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def make_env(n, seed, **over):
    from q1physrl_amd.env import Config
    from q1physrl_amd.tensor_env import TensorVectorEnv
    cfg = O.OracleConfig.get_default(num_envs=n, **over)
    return cfg, TensorVectorEnv(Config(**cfg.__dict__), device=0, seed=seed)


def actions(n, ticks, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    keys = torch.randint(0, 16, (ticks, n), dtype=torch.uint8, device="cuda", generator=g)     # (a 3-key Config ignores bit 3)
    mouse = (torch.rand((ticks, n), device="cuda", generator=g) * 2 - 1) * float(np.float32(720) * np.float32(0.014))
    return keys, mouse.contiguous()


@pytest.mark.parametrize("n,ticks,over", [(4096 + 37, 150, dict(time_limit=0.5, zero_start_prob=0.3)),
                                          (65536, 100, dict(zero_start_prob=1.0)),
                                          (1000, 90, dict(time_limit=0.4, zero_start_prob=0.5, smooth_keys=False, key_press_delay=0.0)),
                                          (777, 80, dict(time_limit=0.4, zero_start_prob=0.5, auto_jump=True, speed_reward=True)),   # SPEC=false kernels
                                          (600, 70, dict(time_limit=0.4, zero_start_prob=0.2, discrete_yaw_steps=5, hover=True)),
                                          (130, 60, dict(time_limit=0.3, allow_yaw=False, allow_jump=False))])
@pytest.mark.parametrize("two_streams", [False, True])
def test_tick_server_equals_per_tick_kernels(n, ticks, over, two_streams):
    import torch
    cfg, a = make_env(n, 7, **over)
    _, b = make_env(n, 7, **over)
    a.reset(); b.reset()
    keys, mouse = actions(n, ticks, 3)
    half = ticks // 3
    # what the dependent producer receives: every tick's result but the last one of each launch, summed per launch in tick order
    sums = torch.zeros((2, 2, n), dtype=torch.float64, device="cuda")
    for t in range(ticks):
        obs_b, rew_b, done_b = b.step_autoreset((keys[t], mouse[t]))
        if t not in (half - 1, ticks - 1):
            sums[int(t >= half), 0] += rew_b.double()
            sums[int(t >= half), 1] += obs_b[:, 0].double()
    # two launches: the second continues where the first stopped (tags go on)
    r1 = a.serve_ticks(keys[:half].contiguous(), mouse[:half].contiguous(), two_streams=two_streams)
    assert not r1["status"].any(), r1["status"]
    first = r1["checksum"].clone()
    r2 = a.serve_ticks(keys[half:].contiguous(), mouse[half:].contiguous(), two_streams=two_streams)
    st = r2["status"]
    assert not st.any(), st
    torch.cuda.synchronize()
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert torch.equal(r2["obs"], obs_b) and torch.equal(r2["obs_from_granules"], obs_b)
    assert torch.equal(r2["reward"], rew_b) and torch.equal(r2["done"], done_b) and torch.equal(r2["zero_start"], b.zero_start)
    # the data really made the round trip: the producer's sums of what it received (same per-env order: exact in float64)
    assert torch.equal(first, sums[0]) and torch.equal(r2["checksum"], sums[1])
    a.close(); b.close()


@pytest.mark.parametrize("n,two_streams", [(131072 + 77, False),        # two sub-batches per workgroup (pair), ragged last wave
                                           (131072 + 77, True),         # ... and 2 envs per lane on two streams
                                           (200000, True),              # 4 envs per lane next to an external producer
                                           (262144, False)])            # BASELINE configs[2]'s batch as ONE resident grid
def test_tick_server_several_envs_per_lane(n, two_streams):
    """Batches above one env per lane of the resident grid: each wave serves 2 or 4 sub-batches of 64 envs in order.  Same bits as
    the per-tick kernels (state, last results, the producer's float64 sums), with in-kernel resets on."""
    import torch
    ticks = 60
    over = dict(time_limit=0.3, zero_start_prob=0.4)
    cfg, a = make_env(n, 9, **over)
    _, b = make_env(n, 9, **over)
    a.reset(); b.reset()
    keys, mouse = actions(n, ticks, 4)
    sums = torch.zeros((2, n), dtype=torch.float64, device="cuda")
    for t in range(ticks):
        obs_b, rew_b, done_b = b.step_autoreset((keys[t], mouse[t]))
        if t != ticks - 1:
            sums[0] += rew_b.double()
            sums[1] += obs_b[:, 0].double()
    res = a.serve_ticks(keys, mouse, two_streams=two_streams)
    assert not res["status"].any(), res["status"]
    torch.cuda.synchronize()
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert torch.equal(res["obs"], obs_b) and torch.equal(res["obs_from_granules"], obs_b)
    assert torch.equal(res["reward"], rew_b) and torch.equal(res["done"], done_b) and torch.equal(res["zero_start"], b.zero_start)
    assert torch.equal(res["checksum"], sums)
    assert int(done_b.sum()) >= 0 and float(sums[0].abs().sum()) > 0
    a.close(); b.close()


@pytest.mark.parametrize("shape", ["1", "2", "3"])
@pytest.mark.parametrize("n", [4096 + 37, 65536 + 64])
def test_tick_server_pair_shapes(n, shape, monkeypatch):
    """q1env_step_persistent_pair: a (server wave, driver wave) workgroup serves 1, 2 or 3 sub-batches of 64 envs (the smallest count
    whose grid is resident; forced here through the measurement knob).  With more than one, the server rotates the sub-batches' states
    through LDS.  Same bits in every shape, over launches that alternate between the forced shape and the default."""
    import torch
    ticks = 90
    over = dict(time_limit=0.4, zero_start_prob=0.3)
    cfg, a = make_env(n, 5, **over)
    _, b = make_env(n, 5, **over)
    a.reset(); b.reset()
    keys, mouse = actions(n, ticks, 6)
    for launch in range(3):
        sums = torch.zeros((2, n), dtype=torch.float64, device="cuda")
        for t in range(ticks):
            obs_b, rew_b, done_b = b.step_autoreset((keys[t], mouse[t]))
            if t != ticks - 1:
                sums[0] += rew_b.double()
                sums[1] += obs_b[:, 0].double()
        with monkeypatch.context() as m:
            if launch != 1:
                m.setenv("Q1ENV_SERVER_SHAPE", shape)
            res = a.serve_ticks(keys, mouse)
        assert not res["status"].any(), (launch, res["status"])
        assert torch.equal(res["checksum"], sums), launch
        assert torch.equal(res["obs"], obs_b) and torch.equal(res["obs_from_granules"], obs_b)
        assert torch.equal(res["reward"], rew_b) and torch.equal(res["done"], done_b) and torch.equal(res["zero_start"], b.zero_start)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    a.close(); b.close()


def test_tick_server_against_the_numpy_oracle():
    """64 zero-start envs x 300 ticks (no episode end): every tick's state transition is the oracle's, bit for bit."""
    import torch
    n, ticks = 64, 300
    cfg, env = make_env(n, 1, zero_start_prob=1.0)
    env.reset()
    np.random.seed(0)
    ora = O.OracleVectorEnv(cfg)
    keys, mouse = actions(n, ticks, 11)
    res = env.serve_ticks(keys, mouse, auto_reset=False)
    assert not res["status"].any()
    kh, mh = keys.cpu().numpy(), mouse.cpu().numpy()
    total = np.zeros(n)
    for t in range(ticks):
        a = np.concatenate([((kh[t][:, None] >> np.arange(4)) & 1).astype(np.float64), mh[t][:, None].astype(np.float64)], axis=1)
        o, r, d, _ = ora.vector_step(a)
        if t < ticks - 1:
            total += r.astype(np.float64)
    assert np.array_equal(o.astype(np.float32), res["obs"].cpu().numpy()) and np.array_equal(r, res["reward"].cpu().numpy())
    assert np.array_equal(o.astype(np.float32), res["obs_from_granules"].cpu().numpy())
    st = env.get_state()
    assert np.array_equal(st["yaw"], ora.yaw) and np.array_equal(st["z_pos"], ora.st["z_pos"])
    assert np.array_equal(np.stack([st["vel_x"], st["vel_y"], st["vel_z"]], 1), ora.st["vel"])
    assert np.array_equal(res["checksum"][0].cpu().numpy(), total)
    env.close()


def test_tick_server_tags_wrap_around():
    """The 24-bit hand-off tag wraps after 16.7 M ticks: tags run 1 .. 0xFFFFFF and skip 0 (a zeroed mailbox must never look like a
    valid action), and a run that crosses the wrap - here forced by starting 20 ticks before it - stays bit-identical."""
    import torch
    n, ticks = 3000, 64
    cfg, a = make_env(n, 9, time_limit=0.3, zero_start_prob=0.5)
    _, c = make_env(n, 9, time_limit=0.3, zero_start_prob=0.5)
    keys, mouse = actions(n, ticks, 4)
    # same history on both handles: one tick, then a reset (the reset's Philox counter is the handle's tick count)
    a.reset(); c.reset()
    r0 = a.serve_ticks(keys[:1].contiguous(), mouse[:1].contiguous())     # creates the server buffers; uses tag 1
    c.step_autoreset((keys[0], mouse[0]))
    assert not r0["status"].any() and int((a._srv["results"][3, :, 0] >> 40).max()) == 1
    a.reset(); c.reset()
    a._srv["tag"] = 0xFFFFFF - 20                                         # the next launch crosses the wrap-around
    res = a.serve_ticks(keys, mouse)
    assert not res["status"].any()
    assert a._srv["tag"] == (0xFFFFFF - 20 + ticks) % 0xFFFFFF == ticks - 20
    for t in range(ticks):
        obs_c, rew_c, done_c = c.step_autoreset((keys[t], mouse[t]))
    torch.cuda.synchronize()
    sa, sc = a.get_state(), c.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sc[k]), k
    assert torch.equal(res["obs"], obs_c) and torch.equal(res["reward"], rew_c) and torch.equal(res["done"], done_c)
    tags = (a._srv["results"] >> 40).unique()                             # every granule carries the last tick's tag, which is not 0
    assert tags.numel() == 1 and int(tags[0]) == (0xFFFFFF - 20 + ticks - 1) % 0xFFFFFF + 1 == ticks - 20
    a.close(); c.close()


def test_tick_server_without_a_producer_times_out_and_reports_it():
    import torch
    n = 2048
    cfg, env = make_env(n, 5, zero_start_prob=0.5)
    env.reset()
    before = env.get_state()
    mailbox = torch.zeros((n,), dtype=torch.int64, device="cuda")
    results = torch.zeros((4, n, 2), dtype=torch.int64, device="cuda")
    status = torch.zeros((5,), dtype=torch.int32, device="cuda")
    t0 = time.perf_counter()
    env._dev.persistent_start(50, 0, mailbox.data_ptr(), results.data_ptr(), env.obs.data_ptr(), 1, True, status.data_ptr(), timeout_s=0.05)
    torch.cuda.synchronize()
    took = time.perf_counter() - t0
    st = status.cpu().numpy()
    assert st[1] == 1 and st[0] == n // 64 and st[2] == 50 and took < 2.0, (st, took)
    after = env.get_state()
    for k in before:
        assert np.array_equal(before[k], after[k]), k           # no tick was served: the state is what it was
    # the device is fine: a normal tick still runs, and the server works when it does get a producer
    keys, mouse = actions(n, 20, 2)
    res = env.serve_ticks(keys, mouse)
    assert not res["status"].any()
    # a driver without a server times out on its side too
    status.zero_()
    side = torch.cuda.Stream()
    env._dev.persistent_drive(side.cuda_stream, 10, 5000, keys.data_ptr(), mouse.data_ptr(), mailbox.data_ptr(), results.data_ptr(), 0,
                              status.data_ptr(), timeout_s=0.05)
    torch.cuda.synchronize()
    assert status.cpu().numpy()[3] == 1
    from q1physrl_amd import _lib
    with pytest.raises(_lib.Q1EnvError, match="own stream|another stream"):
        env._dev.persistent_drive(torch.cuda.current_stream().cuda_stream, 10, 0, keys.data_ptr(), mouse.data_ptr(), mailbox.data_ptr(),
                                  results.data_ptr(), 0, status.data_ptr())
    env.close()
    # a batch that cannot be resident next to its producer is refused up front (it could only time out)
    cfg, big = make_env(1 << 20, 5)
    with pytest.raises(_lib.Q1EnvError, match="too many envs"):
        big._dev.persistent_start(5, 0, mailbox.data_ptr(), results.data_ptr(), 0, 1, True, status.data_ptr())
    big.close()


def test_tick_server_soak_5000_ticks_65536_envs():
    """Soak: 65 536 envs x 5 040 ticks (seven 720-tick launches, in-kernel resets, cycling actions) = 330 M action hand-offs and
    1.3 G result-granule pairs.  The final state equals the per-tick kernels' and the producer's float64 sums of every reward and
    first observation column it received equal the per-tick path's - one stale, torn or skipped granule anywhere would show."""
    import torch
    n, T, launches = 65536, 720, 7
    cfg, a = make_env(n, 3, zero_start_prob=0.3)
    _, b = make_env(n, 3, zero_start_prob=0.3)
    a.reset(); b.reset()
    keys, mouse = actions(n, T, 8)
    want = torch.zeros((launches, 2, n), dtype=torch.float64, device="cuda")
    for l in range(launches):
        for t in range(T):
            obs_b, rew_b, done_b = b.step_autoreset((keys[t], mouse[t]))
            if t != T - 1:
                want[l, 0] += rew_b.double()
                want[l, 1] += obs_b[:, 0].double()
    for l in range(launches):
        res = a.serve_ticks(keys, mouse, two_streams=(l == 3))             # one of the launches in the two-stream form
        assert not res["status"].any(), (l, res["status"])
        assert torch.equal(res["checksum"], want[l]), l
    torch.cuda.synchronize()
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert torch.equal(res["obs_from_granules"], obs_b) and torch.equal(res["reward"], rew_b) and torch.equal(res["done"], done_b)
    a.close(); b.close()


def test_tick_server_with_a_torch_producer_in_the_policy_seat():
    """The protocol's purpose: an EXTERNAL producer.  A torch function of the observation computes every tick's action on a side
    stream between q1env_step_persistent_collect and _publish while the env side is ONE launch; results equal the per-tick loop
    `obs = observe(); for t: obs, r, d = step_autoreset(policy(obs, t))` bit for bit (actions DEPEND on the observations, so any
    stale or reordered hand-off would change the trajectory)."""
    import torch
    n, ticks = 5000, 120
    over = dict(time_limit=0.7, zero_start_prob=0.4)
    cfg, a = make_env(n, 13, **over)
    _, b = make_env(n, 13, **over)
    a.reset(); b.reset()
    rng = float(np.float32(cfg.action_range))

    def policy(obs, t):
        h = (obs[:, 1] * 97.0 + obs[:, 3] * 31.0 + obs[:, 0] * 1009.0).abs()
        keys = ((h.long() + t) & 15).to(torch.uint8)
        mouse = torch.clamp(obs[:, 4] * 4.0 - obs[:, 3] * 2.0 + 0.01 * t, -rng, rng).float().contiguous()
        return keys.contiguous(), mouse

    rew_a, done_a, st = a.serve_with_policy(policy, ticks)
    assert not st.any(), st
    obs = b.observe().clone()
    for t in range(ticks):
        k, m = policy(obs, t)
        o, r, d = b.step_autoreset((k, m))
        obs = o.clone()
        assert torch.equal(rew_a[t], r) and torch.equal(done_a[t], d), t
    torch.cuda.synchronize()
    assert torch.equal(a.obs, obs) and done_a.sum() > n          # episodes ended and were reset in-kernel on the way
    sa, sb = a.get_state(), b.get_state()
    for k_ in sa:
        assert np.array_equal(sa[k_], sb[k_]), k_
    a.close(); b.close()


def test_assertion_build_reads_every_granule_pair_store_back():
    """The -DQ1_CHECK build of the SAME sources (libq1env_check.so, built by __graft_entry__.build()) reads every inline-assembly
    16-byte sc1 granule-pair store back on the device and compares it with the registers it was issued from - the guard against a
    data hazard behind the inline assembly (round 2 had one: lanes 12-15 stored the next pair's first word).  Runs in a subprocess:
    the assertion library is selected through Q1ENV_LIB_PATH before the binding loads, the product library stays what this process
    uses.  The product build must report "not an assertion build"."""
    import subprocess
    import sys
    from q1physrl_amd import build
    from q1physrl_amd.env import Config
    from q1physrl_amd.tensor_env import TensorVectorEnv
    e = TensorVectorEnv(Config(**dict(Config.get_default().__dict__, num_envs=256)), device=0)
    assert e._dev.debug_counters() == (0, 0, 0)
    e.close()
    so = build.build_lib(check=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_check.py"), "--envs", "3000", "65536", "--launches", "2", "--ticks", "120"],
                       capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, Q1ENV_LIB_PATH=so))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("soak_check total") and last.endswith(" 0 mismatches"), last
