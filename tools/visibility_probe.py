#!/usr/bin/env python3
"""Does q1env_rollout's completion signal carry VISIBILITY?  (include/q1env.h "completion signal".)

    python tools/visibility_probe.py [--envs 65536] [--ticks 20] [--reps 1000] [--lib PATH]

Per repetition: the env state is restored, the three tick-major output tensors are POISONED (0xFF bytes, made memory-resident by a
device-wide synchronisation), then
  (1) a reader kernel is enqueued on a second, non-blocking stream; it polls the sequence word the host polls and, the moment the
      launch's number is there, compares all outputs with the expected ones, NEWEST TICK FIRST, using system-scope loads
      (q1env_diag_signal_reader); its reaction time behind the kernel's own end stamp is reported;
  (2) the rollout is launched with Q1ENV_SIGNAL_WAIT (launch + poll in one call);
  (3) the instant that call returns - NO runtime synchronisation of the launch stream, no event - obs / reward / done are copied to
      pinned host memory on a third non-blocking stream (a DMA read that is not ordered behind the launch) and compared on the host.
Expected outputs come from the same rollout followed by a full device synchronisation.  A repetition is STALE if either consumer saw
anything else.  --lib runs the same loop against another build of the library (tools/_bin/libq1env_outstores1.so: round 4's
non-temporal stores, the control) in a fresh process.  Prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def probe(n=65536, ticks=20, reps=1000, seed=3, dma=True, reader=True, reader_workgroups=256):
    import numpy as np
    import torch
    from q1physrl_amd import _lib as L, env as E
    from q1physrl_amd.device import DeviceEnv
    d = torch.device("cuda", 0)
    cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    g = torch.Generator(device="cpu").manual_seed(seed)
    keys = torch.randint(0, 16, (ticks, n), dtype=torch.uint8, generator=g).to(d)
    mouse = ((torch.rand((ticks, n), generator=g) * 2 - 1) * float(cfg.action_range)).to(d)
    # one arena for the three outputs (obs | reward | done, each 256-B aligned) so that the reader kernel checks them in one pass
    so, sr, sd = ticks * n * 24, ticks * n * 4, ticks * n
    al = lambda x: (x + 255) // 256 * 256                      # noqa: E731
    total = al(so) + al(sr) + al(sd)
    total = (total + 7) // 8 * 8
    arena = torch.empty((total,), dtype=torch.uint8, device=d)
    expect = torch.empty_like(arena)
    host = torch.empty((total,), dtype=torch.uint8).pin_memory()
    result = torch.zeros((4,), dtype=torch.int64, device=d)

    def ptrs(t):
        b = t.data_ptr()
        return b, b + al(so), b + al(so) + al(sr)

    dev = DeviceEnv(cfg, device=0)
    dev.snapshot_state()
    expect.fill_(0xFF)
    o, r, dn = ptrs(expect)
    dev.rollout_dev(ticks, L.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, L.OBS_F32, o, r, dn)
    dev.sync()
    torch.cuda.synchronize()
    expect_h = expect.cpu().numpy().copy()
    s_reader, s_dma = torch.cuda.Stream(device=d), torch.cuda.Stream(device=d)
    o, r, dn = ptrs(arena)
    call = dev.prepare_rollout(ticks, L.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), 0, L.OBS_F32, o, r, dn, L.STAMP_START | L.SIGNAL_WAIT)
    stale_dma = stale_reader = timeouts = 0
    react_us, newest_us = [], []
    worst_dma_bytes = worst_reader_words = 0
    t_begin = time.perf_counter()
    for rep in range(reps):
        dev.restore_state()
        arena.fill_(0xFF)
        result.zero_()
        dev.sync()
        torch.cuda.synchronize()                               # poison and state are in memory; nothing in flight
        if reader:
            dev.diag_signal_reader(s_reader.cuda_stream, arena.data_ptr(), expect.data_ptr(), al(so), al(so) + al(sr), ticks, result.data_ptr(),
                                   reader_workgroups)
        call()                                                 # launch + poll the kernel-written sequence word; returns when it is seen
        if dma:
            with torch.cuda.stream(s_dma):
                host.copy_(arena, non_blocking=True)           # DMA read on a stream that is NOT ordered behind the launch
            s_dma.synchronize()
            bad = int(np.count_nonzero(host.numpy() != expect_h))
            if bad:
                stale_dma += 1
                worst_dma_bytes = max(worst_dma_bytes, bad)
        if reader:
            s_reader.synchronize()
            res = result.cpu().numpy()
            if res[1]:
                timeouts += 1
            if res[0]:
                stale_reader += 1
                worst_reader_words = max(worst_reader_words, int(res[0]))
            a_, b_, hz = C.c_uint64(), C.c_uint64(), C.c_double()
            L.check(dev._lib.q1env_signal_read(dev._h, C.byref(a_), C.byref(b_), C.byref(hz)))
            if res[2]:
                react_us.append((int(res[2]) - int(b_.value)) / hz.value * 1e6)       # signal seen by the reader - the kernel's end stamp
                newest_us.append((int(res[3]) - int(b_.value)) / hz.value * 1e6)      # newest tick fully checked - end stamp
        dev.sync()
    wall = time.perf_counter() - t_begin
    out = {"lib": os.environ.get("Q1ENV_LIB_PATH") or "q1physrl_amd/libq1env.so", "build_id": L.build_id(), "envs": n, "ticks": ticks, "reps": reps,
           "bytes_checked_per_rep": total, "stale_reps_dma_second_stream": stale_dma if dma else None, "worst_dma_bytes_differing": worst_dma_bytes,
           "stale_reps_reader_kernel": stale_reader if reader else None, "worst_reader_words_differing": worst_reader_words,
           "reader_timeouts": timeouts, "wall_s": wall,
           "reader_saw_signal_us_after_end_stamp_median": float(np.median(react_us)) if react_us else None,
           "reader_saw_signal_us_after_end_stamp_min": float(np.min(react_us)) if react_us else None,
           "reader_checked_newest_tick_us_after_end_stamp_median": float(np.median(newest_us)) if newest_us else None}
    dev.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--ticks", type=int, default=20)
    ap.add_argument("--reps", type=int, default=1000)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    if a.lib:
        os.environ["Q1ENV_LIB_PATH"] = os.path.abspath(a.lib)
    print(json.dumps(probe(a.envs, a.ticks, a.reps)), flush=True)


if __name__ == "__main__":
    main()
