"""GPU: the kernel-written completion signal of q1env_rollout (include/q1env.h "completion signal", ABI v4).
The signal must (a) change no result, (b) arrive exactly once per signalled launch, in order, with the sequence word the host polls,
(c) only after every wave of the launch has retired its stores - the outputs copied to the host right after the wait are complete,
(d) carry device stamps whose difference is a sane kernel duration, and (e) work for ragged batches (a tail wave with inactive lanes)
and for q1env_signal_mark behind a launch that carries no signal itself."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(n, ticks, seed=3):
    import torch
    from q1physrl_amd import _lib, env as E
    from q1physrl_amd.device import DeviceEnv
    cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    d = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(seed)
    keys = torch.randint(0, 16, (ticks, n), dtype=torch.uint8, generator=g).to(d)
    mouse = ((torch.rand((ticks, n), generator=g) * 2 - 1) * float(cfg.action_range)).to(d)
    out = lambda: (torch.zeros((ticks, n, 6), dtype=torch.float32, device=d), torch.zeros((ticks, n), dtype=torch.float32, device=d),   # noqa: E731
                   torch.full((ticks, n), 7, dtype=torch.uint8, device=d))
    return torch, _lib, DeviceEnv, cfg, keys, mouse, out


@pytest.mark.parametrize("n", [65536, 1000, 64, 1])
def test_signalled_rollout_equals_plain_rollout_and_outputs_are_complete(n):
    ticks = 40
    torch, L, DeviceEnv, cfg, keys, mouse, out = _setup(n, ticks)
    a, b = DeviceEnv(cfg, device=0), DeviceEnv(cfg, device=0)
    o1, r1, d1 = out()
    o2, r2, d2 = out()
    a.rollout_dev(ticks, L.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, L.OBS_F32, o1.data_ptr(), r1.data_ptr(), d1.data_ptr())
    a.sync()
    b.rollout_dev(ticks, L.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, L.OBS_F32, o2.data_ptr(), r2.data_ptr(), d2.data_ptr(),
                  auto_reset=L.STAMP_START | L.SIGNAL)
    b.signal_wait(10.0)
    # the copies below are ordered behind the launch on the device anyway, so this checks (a) and that nothing is left behind; the
    # memory model - results readable by another agent the moment the signal is seen - is test_signal_carries_visibility_* below
    assert torch.equal(o1.cpu(), o2.cpu()) and torch.equal(r1.cpu(), r2.cpu()) and torch.equal(d1.cpu(), d2.cpu())
    assert int(d2.max()) <= 1                                  # every done byte was written (initialised to 7)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    el = b.signal_elapsed()
    assert 1e-6 < el < 5e-3, el                                # 40 ticks: tens of microseconds
    a.close(); b.close()


def test_signal_carries_visibility_dma_second_stream_and_concurrent_reader_65536_envs_1000_reps():
    """"done" must mean "consumable".  q1env_rollout writes its outputs with system-scope write-through stores, so the
    kernel-written signal implies every result is at the memory side.  1 000 repetitions at the bench's driver shape (65 536 envs x 20
    ticks): outputs poisoned, rollout launched with Q1ENV_SIGNAL_WAIT, and the instant the call returns - with NO runtime
    synchronisation of the launch stream - (a) a DMA copy on another non-blocking stream brings obs / reward / done to the host, and
    (b) a reader kernel that was already polling the same sequence word on a third stream has compared them with system-scope loads
    microseconds after the signal.  Neither may ever see anything but the plain (fully synchronised) rollout's bytes.
    (tools/visibility_probe.py is the loop; profiles/r5_visibility.txt holds the same probe against the round-4 store forms.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import visibility_probe
    res = visibility_probe.probe(n=65536, ticks=20, reps=1000)
    assert res["reader_timeouts"] == 0, res
    assert res["stale_reps_dma_second_stream"] == 0 and res["stale_reps_reader_kernel"] == 0, res
    assert res["bytes_checked_per_rep"] >= 65536 * 20 * 29


def test_signal_carries_visibility_small_batch_outputs_in_host_coherent_memory():
    """The same guarantee where the consumer is the HOST THREAD itself: outputs of a 1 024-env x 20-tick rollout live in pinned,
    host-coherent memory; the moment the launching call (Q1ENV_SIGNAL_WAIT) returns they are read through the host pointer - no stream
    synchronisation, no copy - and must be complete.  1 000 repetitions from a poisoned buffer."""
    n, ticks = 1024, 20
    torch, L, DeviceEnv, cfg, keys, mouse, out = _setup(n, ticks)
    dev = DeviceEnv(cfg, device=0)
    dev.snapshot_state()
    o1, r1, d1 = out()
    dev.rollout_dev(ticks, L.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, L.OBS_F32, o1.data_ptr(), r1.data_ptr(), d1.data_ptr())
    dev.sync()
    eo, er, ed = o1.cpu().numpy().view(np.uint32), r1.cpu().numpy().view(np.uint32), d1.cpu().numpy()
    ho = torch.empty((ticks, n, 6), dtype=torch.float32).pin_memory()
    hr = torch.empty((ticks, n), dtype=torch.float32).pin_memory()
    hd = torch.empty((ticks, n), dtype=torch.uint8).pin_memory()
    vo, vr, vd = ho.numpy().view(np.uint32), hr.numpy().view(np.uint32), hd.numpy()
    call = dev.prepare_rollout(ticks, L.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, L.OBS_F32, ho.data_ptr(), hr.data_ptr(), hd.data_ptr(),
                               L.STAMP_START | L.SIGNAL_WAIT)
    stale = 0
    for rep in range(1000):
        dev.restore_state()
        dev.sync()
        vo.fill(0xFFFFFFFF); vr.fill(0xFFFFFFFF); vd.fill(0xFF)
        call()                                                 # returns when the kernel-written sequence word has been seen
        if not (np.array_equal(vo, eo) and np.array_equal(vr, er) and np.array_equal(vd, ed)):
            stale += 1
        dev.sync()
    assert stale == 0, stale
    dev.close()


def test_sequence_of_signals_and_signal_mark():
    n, ticks = 4096, 8
    torch, L, DeviceEnv, cfg, keys, mouse, out = _setup(n, ticks)
    dev = DeviceEnv(cfg, device=0)
    o, r, dn = out()
    seq = C.c_uint64()
    last = 0.0
    for rep in range(50):                                      # back-to-back signalled launches: the ticket counter returns to zero each time
        dev.rollout_dev(ticks, L.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, L.OBS_F32, o.data_ptr(), r.data_ptr(), dn.data_ptr(),
                        auto_reset=L.STAMP_START | L.SIGNAL_WAIT)          # launch + wait in one call
        el = dev.signal_elapsed()
        assert 0 < el < 5e-3
        last = el
    assert last > 0
    # a region that ends in a launch without a signal: start stamp on the first rollout, signal_mark behind the reset
    dev.rollout_dev(ticks, L.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, L.OBS_F32, o.data_ptr(), r.data_ptr(), dn.data_ptr(),
                    auto_reset=L.STAMP_START)
    dev.reset_philox_dev(99, 0, True)
    dev.signal_mark()
    dev.signal_wait(10.0)
    el2 = dev.signal_elapsed()
    assert el2 > 0 and el2 < 5e-3
    dev.sync()
    del seq
    dev.close()


def test_signal_wait_without_a_request_is_an_error_not_a_hang():
    from q1physrl_amd import _lib
    torch, L, DeviceEnv, cfg, keys, mouse, out = _setup(64, 2)
    dev = DeviceEnv(cfg, device=0)
    with pytest.raises(_lib.Q1EnvError):
        dev.signal_wait(0.1)
    dev.close()


def test_build_id_matches_the_sources():
    from q1physrl_amd import _lib, build
    assert _lib.build_id() == build.sources_sha16() and len(_lib.lib_sha16()) == 16


@pytest.mark.parametrize("n", [1, 100, 4096])
def test_host_direct_step_and_reset_equal_the_staged_path(n, monkeypatch):
    """Small batches (<= 4 096 envs) take the host-direct form of q1env_step_host / q1env_reset_draws_host / q1env_observe_host /
    q1env_decode_host: the kernel reads the actions
    from and writes obs / reward / done / zero_start to host-coherent pinned memory itself and says so with the completion signal (one
    launch + a poll instead of two copy commands + a stream synchronisation).  It must be indistinguishable from the staged path
    (Q1ENV_HOST_DIRECT=0): same returns on every tick of a seeded trace with RLlib-style reset_at calls, same final state."""
    from q1physrl_amd import env as E
    out = []
    for knob in ("1", "0"):
        monkeypatch.setenv("Q1ENV_HOST_DIRECT", knob)
        np.random.seed(11)
        cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n, "time_limit": 0.5, "zero_start_prob": 0.3})
        env = E.VectorPhysEnv(cfg, device=0)
        rng = np.random.default_rng(5)
        trace = [env.vector_reset().copy()]
        dec = E.ActionDecoder(cfg, device=0)                   # the stand-alone decoder (mkdemo.py:47-55): q1env_decode_host
        dec.vector_reset(np.full(n, 90.0))
        for t in range(60):
            a = np.concatenate([(rng.random((n, 4)) < 0.5).astype(np.float64), rng.uniform(-10, 10, (n, 1))], axis=1)
            obs, rew, done, infos = env.vector_step(a)
            trace += [obs.copy(), rew.copy(), done.copy(), np.array([infos[i]["zero_start"] for i in range(min(n, 8))])]
            if t % 7 == 0:
                trace.append(env._get_obs().copy())            # q1env_observe_host
                yaw, sm, fm, jp = dec.map(a, rng.uniform(-300, 300, n).astype(np.float32), np.full(n, 0.5 - t / 144.0))
                trace += [np.asarray(yaw).copy(), np.asarray(sm).copy(), np.asarray(fm).copy(), np.asarray(jp).copy()]
            for i in np.flatnonzero(done)[:16]:
                trace.append(env.reset_at(int(i)).copy())
        st = env._dev.get_state()
        trace += [st[k] for k in sorted(st)]
        env.close()
        out.append(trace)
    assert len(out[0]) == len(out[1])
    for x, y in zip(*out):
        assert x.dtype == y.dtype and np.array_equal(x.view(np.uint8), y.view(np.uint8))
