#!/usr/bin/env python3
"""bench.py - env-steps/s of the q1physrl hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--mode auto|rollout|step|server]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no RANK / WORLD_SIZE in the environment launches its own N workers (one process
per GPU, LOCAL_RANK = device index, rendezvous on 127.0.0.1, every rank pinned to the CPU cores of its GPU's NUMA node) and rank 0
prints the JSON line; under torch.distributed.run the launcher's environment is used as is.  BASELINE configs[3] (1 048 576 envs on
8 GPUs) is `--gpus 8 --envs 131072`.

A "step" is one tick (one VectorPhysEnv.vector_step, reference env.py:482-510) of the whole batch.
Workload at N=1 = BASELINE.json configs[1]: 65 536 envs, zero-start 100 m run (zero_start_prob = 1,
get_default Config, dt = 1/72, 720-tick episodes), random actions (keys flip with p = 0.05 per tick,
mouse ~ U(-action_range, action_range) float32).  Inputs (the packed 5 B/env action tensor of a whole
720-tick episode) are resident in HBM before the timed region.  In EVERY primary-capable mode each tick's obs float32 (N,6),
reward float32 and done uint8 are written to HBM where any later kernel / the host can read them (what vector_step returns,
env.py:507-510); all envs are reset on device at each episode end (inside the timed region).

  --mode auto     (default) = rollout.
  --mode rollout  q1env_rollout: the K ticks as ONE launch per episode chunk (an episode boundary splits a launch), env state in
                  registers between ticks, tick t's packed action read from the resident action tensor, tick t's obs / reward / done
                  written TICK-MAJOR to HBM ((T,N,6) / (T,N) / (T,N) tensors).  This is configs[1] as stated - "random actions" are
                  known in advance - and the boundary's multi-tick entry point (include/q1env.h q1env_rollout).  Bound: float64-heavy
                  VALU issue of one wave per SIMD, not HBM: `roofline` reports VALU-busy cycles against the chip's VALU cycles and the
                  measured HBM bytes next to it.
  --mode step     ONE step_kernel launch per tick, 720 launches replayed from one hipGraph.  The granularity the drop-in API has
                  (a policy can sit between ticks); HBM-bound formulation (reads + writes the whole SoA state every tick); reported as
                  "per_tick_step" when it is not the primary mode.
  --mode server   the resident tick server (q1env_step_persistent_pair): ONE dispatch serves all K ticks; a server wave and a
                  DEPENDENT stand-in producer wave per workgroup hand actions / results over through LDS every tick.  Per-tick outputs
                  do NOT reach HBM in this mode (only the last tick's do) - it measures the tick-to-tick round trip a resident policy
                  would see, and is therefore never the primary mode; reported as "persistent_server".

Multi-GPU: one process per GPU, the batch is split (65 536 envs per GPU, weak scaling), no collective on
the data path; ranks only meet in the barriers around the timed region and in the MAX of the elapsed time.

Timed region (every rank): graphs instantiated and uploaded, warm-up ticks, barrier + torch.cuda.synchronize(); t0; EXACTLY K ticks
enqueued; completion; t1; torch.cuda.synchronize(); barrier.  `value` uses the MAX over ranks of t1 - t0 (host wall clock).
  rollout mode (the default): no marker packets ride with the launch and completion is the kernel's own signal - the last wave to have its
      stores acknowledged writes an end stamp and a sequence number into host-coherent pinned memory, which the host polls (q1env_signal_wait,
      include/q1env.h "completion signal").  Since round 5 the signal CARRIES VISIBILITY: every per-tick output and the final state are
      written with system-scope write-through stores, whose acknowledgement comes from the memory side (Infinity Cache / HBM), so t1 is a
      time after which obs / reward / done need no further wait to be read by the host, a DMA engine or another kernel - held to that by
      tests/test_hip_signal.py (DMA copy on a second stream + a concurrent reader kernel, 1 000 repetitions; plain stores are caught stale
      every time, profiles/r5_visibility.txt).  The torch.cuda.synchronize() that follows t1 is then pure runtime cost (the dispatch's
      completion packet travelling through the runtime); its duration is reported (`timed_region_us.post_sync_us`), as is
      `ms_per_step_incl_runtime_sync`, the same region's wall time had t1 been taken after it.  Kernel time of the timed region = device
      wall-clock stamps written by its first and last wave (`device_stamp_us`); the SAME K ticks are then repeated from the same state
      between two HIP events on the launch stream (`roofline.avg_launch_us`, what the roofline uses).
  step / server modes: HIP events around the launches inside the timed region, ONE torch.cuda.synchronize() ends it (t1 after it).

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import functools
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 204.0            # algorithmic bytes per env-step of the PER-TICK formulation: 2 x 85 B state + 5 B action + 29 B outputs
B_FUSED = 34.0           # algorithmic bytes per env-step of a register-resident multi-tick kernel: 5 B action in + 29 B outputs out ...
B_STATE = 170.0          # ... + the 85 B state read and written ONCE per launch, per env
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
PEAK_CLOCK_GHZ = 2.4     # MI355X_MICROARCH.md: max shader clock
EPISODE_TICKS = 720
PRIMARY_AUTO = "rollout"  # what --mode auto measures: every tick's obs / reward / done reach HBM


# data/params.yml:16-33 env_config of the reference (BASELINE configs[2] / configs[4])
PARAMS_YML = dict(action_range=10, allow_jump=True, allow_yaw=True, auto_jump=False, discrete_yaw_steps=-1, fmove_max=800, smove_max=1060,
                  hover=False, initial_yaw_range=(0, 360), key_press_delay=0.3, max_initial_speed=700, smooth_keys=True, speed_reward=False,
                  time_delta=0.013888888888888, time_limit=10, zero_start_prob=0.01)


def make_actions(n, ticks, action_range, seed):
    """Packed actions for one episode, tick-major: keys uint8 (ticks, n) bit k = Key k, mouse float32 (ticks, n)."""
    rng = np.random.default_rng(seed)
    keys = np.empty((ticks, n), dtype=np.uint8)
    cur = rng.integers(0, 16, size=n, dtype=np.uint8)
    for t in range(ticks):
        flip = np.zeros(n, dtype=np.uint8)
        for k in range(4):
            flip |= (rng.random(n) < 0.05).astype(np.uint8) << k
        cur = cur ^ flip
        keys[t] = cur
    mouse = rng.uniform(-action_range, action_range, size=(ticks, n)).astype(np.float32)
    return keys, mouse


def cpu_baseline(n, action_range, budget_s=10.0, gpu_check=None, extras=True):
    """The NumPy oracle (from-scratch restatement of the reference's NumPy path, bit-pinned to the reference by
    tests/golden) timed on this box's host: 1 process, 1 thread (NumPy elementwise kernels are single-threaded).
    While it runs it also serves as the CHECKER of the metric's second half: after 720 ticks (10 s of game time) its
    state is compared with the GPU env driven by the same actions (`gpu_check`) -> max |pos - NumPy ref|.
    Also the C oracle on all host cores (OpenMP), reported as extra information."""
    from oracle import np_oracle as O
    np.random.seed(0)
    env = O.OracleVectorEnv(O.OracleConfig.get_default(num_envs=n, zero_start_prob=1.0))
    rng = np.random.default_rng(1)
    acts = [np.concatenate([(rng.random((n, 4)) < 0.5).astype(np.float64),
                            rng.uniform(-action_range, action_range, (n, 1)).astype(np.float32).astype(np.float64)], axis=1)
            for _ in range(8)]
    dist = np.zeros((n, 2))
    snap = None
    spent, ticks = 0.0, 0
    while spent < budget_s or ticks < EPISODE_TICKS - 1:
        t0 = time.perf_counter()
        _, _, done, _ = env.vector_step(acts[ticks % len(acts)])
        spent += time.perf_counter() - t0
        ticks += 1
        if ticks <= EPISODE_TICKS - 1:                 # stop integrating before the episode-ending tick
            dist += env.cfg.time_delta * env.st["vel"][:, :2].astype(np.float64)
        if ticks == EPISODE_TICKS - 1:
            snap = {"dist": dist.copy(), "z": env.st["z_pos"].copy(), "vel": env.st["vel"].copy(), "yaw": env.yaw.copy(),
                    "on_ground": env.st["on_ground"].copy(), "t_rem": env.t_rem.copy()}
    dt = spent
    out = {"value": n * ticks / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{ticks} ticks x {n} envs, oracle/np_oracle.py (NumPy {np.__version__}, ndarray actions), {dt:.1f} s; 1 of {os.cpu_count()} cores"}
    if gpu_check is not None and snap is not None:
        g = gpu_check(acts, EPISODE_TICKS - 1)
        out["parity_vs_gpu_after_719_ticks"] = {
            "envs": n,
            "max_abs_pos_xy_diff": float(max(np.abs(g["pos_x"] - snap["dist"][:, 0]).max(), np.abs(g["pos_y"] - snap["dist"][:, 1]).max())),
            "max_abs_z_diff": float(np.abs(g["z_pos"] - snap["z"]).max()),
            "max_abs_vel_diff": float(max(np.abs(g["vel_x"] - snap["vel"][:, 0]).max(), np.abs(g["vel_y"] - snap["vel"][:, 1]).max(),
                                          np.abs(g["vel_z"] - snap["vel"][:, 2]).max())),
            "max_abs_yaw_diff": float(np.abs(g["yaw"] - snap["yaw"]).max()),
            "vel_bit_identical_fraction": float(np.mean(np.stack([g["vel_x"], g["vel_y"], g["vel_z"]], 1).view(np.uint32) == snap["vel"].view(np.uint32))),
            "on_ground_mismatches": int(np.count_nonzero(((g["flags"] & 1) != 0) != snap["on_ground"])),
            "max_abs_time_remaining_diff": float(np.abs(g["time_remaining"] - snap["t_rem"]).max()),
            "max_abs_y_travelled": float(np.abs(snap["dist"][:, 1]).max())}
    if not extras:
        return out
    # the reference's verbatim call pattern: RLlib hands vector_step a list of N tuples (scalars + (1,) arrays), which
    # _fix_actions (env.py:221-223) converts with a Python double loop - 86 % of the reference's wall time at this size
    rows = acts[0]
    as_list = [tuple([int(x) for x in r[:4]] + [np.array([r[4]], dtype=np.float32)]) for r in rows]
    t0 = time.perf_counter()
    k_list = 3
    for _ in range(k_list):
        env.vector_step(as_list)
    out["numpy_port_rllib_list_actions"] = {"value": n * k_list / (time.perf_counter() - t0), "unit": "env-steps/s", "cores": 1,
                                            "sample": f"{k_list} ticks of {n} envs with list-of-tuples actions (per-element Python conversion)"}
    try:
        from oracle import c_oracle as CO
        threads = max(1, min(os.cpu_count() or 1, 64))
        dt1 = CO.time_rollout(n, 32, threads, action_range)   # warm-up (page faults, thread pool) + rate estimate
        tk = int(min(max(32, 4.0 / max(dt1 / 32, 1e-6)), 20000))   # about 4 s of work
        dtc = CO.time_rollout(n, tk, threads, action_range)
        out["c_port_openmp"] = {"value": n * tk / dtc, "unit": "env-steps/s", "cores": threads,
                                "sample": f"{tk} ticks of {n} envs, oracle/q1_oracle.c, {threads} OpenMP threads, {dtc:.2f} s"}
    except Exception as ex:   # noqa: BLE001 - the C oracle is optional extra information
        out["c_port_openmp"] = {"error": repr(ex)}
    return out


def size_sweep(dev_index, sizes=(262144, 1048576, 2097152, 4194304), ticks=72, reps=4):
    """step_kernel at larger batches (same Config, packed random actions resident in HBM, hipGraph of `ticks` launches):
    where the per-tick kernel stops being launch/latency-bound.  Extra information next to the contract fields."""
    import torch
    from q1physrl_amd import _lib, env as E
    from q1physrl_amd.device import DeviceEnv
    d = torch.device("cuda", dev_index)
    rows = []
    for n in sizes:
        cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
        dev = DeviceEnv(cfg, device=dev_index)
        keys = torch.randint(0, 16, (ticks, n), dtype=torch.uint8, device=d)
        mouse = torch.rand((ticks, n), device=d) * 20.0 - 10.0
        obs = torch.empty((n, 6), dtype=torch.float32, device=d)
        rew = torch.empty((n,), dtype=torch.float32, device=d)
        done = torch.empty((n,), dtype=torch.uint8, device=d)
        torch.cuda.synchronize()

        def go():
            dev.step_many_dev(ticks, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(),
                              done.data_ptr(), out_stride_ticks=0, use_graph=True)
        go()
        dev.sync()
        dev.timer_start()
        for _ in range(reps):
            go()
        us = dev.timer_stop() * 1e3 / (reps * ticks)
        gbps = B_ALG * n / (us * 1e-6) / 1e9
        # on-box copy ceiling in the kernel's own access pattern: calib_copy_kernel reads and writes the 85-B state
        dev.calibrate_traffic(4)
        dev.timer_start()
        dev.calibrate_traffic(32)
        copy_us = dev.timer_stop() * 1e3 / 32
        copy_gbps = 170.0 * n / (copy_us * 1e-6) / 1e9
        #  a batch whose state (85 B read + 85 B written per env) fits the 256 MB Infinity Cache is served from there from the
        # second tick on: its GB/s is cache bandwidth and is NOT reported as a fraction of the HBM peak
        in_mall = 170.0 * n < 256e6
        rows.append({"envs": n, "us_per_tick": us, "env_steps_per_s": n / (us * 1e-6), "achieved_GBps": gbps,
                     "served_from": "Infinity Cache (state working set %.0f MB < 256 MB): cache bandwidth, not an HBM fraction" % (170.0 * n / 1e6) if in_mall else "HBM",
                     "frac": None if in_mall else gbps / HBM_PEAK_GBPS,
                     "state_copy_kernel_GBps": copy_gbps, "frac_of_copy_kernel": gbps / copy_gbps})
        dev.close()
        del keys, mouse, obs, rew, done
    return rows


def sampler_block(dev_index, sizes=(32768, 262144), horizon=128, reps=4):
    """BASELINE configs[4] next to the contract fields: the env inside a sampler loop with the policy in it (params.yml Config,
    random-init policy of the reference's shape) at the per-GPU shard (32 768 envs) and at the whole batch on ONE GPU (262 144) - the
    two-launch tick (fused matrix-core forward + fused sample/step/reset, the horizon captured in a hipGraph) and the resident sampler
    (one dispatch per horizon + one batched value forward; above 65 536 envs its workgroups run as successive sets).  128-tick horizons (the
    training configuration), two warm-up horizons, best of `reps`, HIP events."""
    import torch
    from q1physrl_amd import policy as P
    from q1physrl_amd.env import Config
    from q1physrl_amd.sampler import GpuSampler
    from q1physrl_amd.tensor_env import TensorVectorEnv
    params_yml = PARAMS_YML
    rows = []
    for n in sizes:
        row = {"envs": n, "horizon": horizon, "workload": "BASELINE configs[4]: sampler loop with the policy forward in it, params.yml Config"
               + (" (per-GPU shard of 8)" if n == 32768 else " (the whole batch on one GPU)")}
        for label, kw in (("two_launch", dict(use_graph=True)), ("resident", dict(resident=True))):
            env = TensorVectorEnv(Config(num_envs=n, **params_yml), device=dev_index, seed=1)
            s = GpuSampler(env, P.FusedPolicyForward(P.Q1Policy().cuda(), env), horizon=horizon, **kw)
            s.collect(); s.collect()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e30
            for _ in range(reps):
                env.use_current_stream()
                e0.record(); s.collect(check_status=False); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / horizon)
            if label == "resident" and s.resident_status().any():
                raise RuntimeError(f"resident sampler timed out: status {s.resident_status()}")
            row[label + "_us_per_tick"] = best
            row[label + "_env_steps_per_s"] = n / (best * 1e-6)
            env.close()
            del s, env
            torch.cuda.empty_cache()
        rows.append(row)
    return rows


def load_pmc(mode, n, lib_build_id=None):
    """(entry, stale): hardware-counter figures of this mode's kernel at this batch size from the round's rocprofv3 PMC passes
    (tools/profile_round.sh -> tools/summarize_pmc.py -> profiles/pmc.json), or (None, False) if that size was not profiled.
    Entry: {"kernel": exact name, "ticks_per_launch": T the passes ran at, "fetch_x2_B", "write_B": HBM bytes per launch (FETCH_SIZE x 2
    per MI355X_MICROARCH.md's gfx950 correction, calibrated on calib_copy_kernel), "valu_busy_cycles": 4 x SQ_ACTIVE_INST_VALU per launch
    (cycles in which a SIMD's VALU executes an instruction, summed over SIMDs), "insts_valu": SQ_INSTS_VALU per launch, "waves",
    "build_id": q1env_build_id() of the library the passes profiled}.
    Staleness guard: counters describe ONE build of the kernels.  When the entry's build_id (or the file's
    top-level "_build_id") differs from the library this process loaded, the entry is withheld and stale = True: `traffic` and `valu`
    become null instead of silently describing other code."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
            doc = json.load(f)
    except Exception:   # noqa: BLE001
        return None, False
    ent = doc.get(f"{mode}_{n}")
    if ent is None:
        return None, False
    bid = ent.get("build_id") or doc.get("_build_id")
    if lib_build_id is not None and bid != lib_build_id:
        return None, True
    return ent, False


def traffic_per_launch(pmc, n, ticks_per_launch, resident_state):
    """Measured HBM bytes of one launch of `ticks_per_launch` ticks, from a PMC entry taken at pmc["ticks_per_launch"] ticks: a
    register-resident kernel moves the 170 B/env state once per launch and the rest per tick, so its per-tick part is scaled."""
    if pmc is None or pmc.get("fetch_x2_B") is None or pmc.get("write_B") is None:
        return None
    total = float(pmc["fetch_x2_B"]) + float(pmc["write_B"])
    t0 = float(pmc.get("ticks_per_launch", 1))
    if not resident_state:
        return total * ticks_per_launch / t0
    per_tick = max(total - B_STATE * n, 0.0) / t0
    return B_STATE * n + per_tick * ticks_per_launch


# ---- the ONE stdout line ---------------------------------------------------------------------------------------------------------
REGION_REPS = 15           # repetitions of the --steps region reported next to the single-shot `value` (median / min / max: timed_region_us)
LINE_MAX_BYTES = 4096      # the driver keeps the last 8 KB of stdout and parses the line from it (round 4's 20.7 KB line was lost: parsed null)


def _sig(x, digits=7):
    """floats to `digits` significant digits (the line is for reading and for the driver's arithmetic checks: 1e-6 relative is plenty)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def extra_path_default():
    return os.environ.get("Q1_BENCH_EXTRA") or os.path.join(ROOT, "bench_extra.json")


def write_extra(out, path=None):
    """Everything bench.py measured (per-rank rows, host splits, secondary modes, 720-tick steady state, size sweep, sampler block, notes)
    goes to a side file; the stdout line only names it.  Returns the path written, or None when the directory is not writable."""
    path = path or extra_path_default()
    try:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        os.replace(tmp, path)
        return path
    except OSError as ex:
        sys.stderr.write(f"bench.py: could not write {path}: {ex!r}\n")
        return None


def contract_line(out, extra_path):
    """The contract line: the task statement's fields + `roofline` + `cpu_baseline` + one number per steady-state mode, < LINE_MAX_BYTES
    for any rank count (tests/test_bench_launcher.py, tests/test_bench_gpu.py assert the length at 1, 2 and 8 ranks)."""
    ro = out.get("roofline") or {}
    hs = ro.get("host_split_us") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    line["config"] = out["config"]
    line["roofline"] = {k: ro.get(k) for k in ("bound", "frac_axis", "achieved", "peak", "unit", "frac", "frac_8d_204B", "traffic", "kernel", "avg_launch_us",
                                               "ticks_per_launch", "launches", "pmc_stale")}
    if isinstance(line["roofline"].get("kernel"), str):
        line["roofline"]["kernel"] = line["roofline"]["kernel"].split("  (")[0]      # the instantiation; the legend of its arguments is in the side file
    valu = ro.get("valu") or {}
    if valu.get("valu_busy_frac") is not None:
        line["roofline"]["valu_busy_frac"] = valu["valu_busy_frac"]
    cb = out.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
        par = cb.get("parity_vs_gpu_after_719_ticks")
        if par:
            line["cpu_baseline"]["max_abs_pos_diff_vs_gpu_10s"] = max(par["max_abs_pos_xy_diff"], par["max_abs_z_diff"])
            line["cpu_baseline"]["vel_bit_identical_fraction"] = par["vel_bit_identical_fraction"]
    else:
        line["cpu_baseline"] = None
    for k in ("mode", "mode_fallback", "env_impl", "lib_sha16", "lib_build_id", "ms_per_step_incl_runtime_sync"):
        line[k] = out.get(k)
    line["completion"] = out.get("completion")
    if hs or out.get("timed_region_reps"):
        line["timed_region_us"] = {k: hs.get(k) for k in ("launch_to_signal_seen_us", "device_stamp_us", "hip_event_us", "post_sync_us",
                                                          "wall_incl_runtime_sync_us") if hs.get(k) is not None}
        rs = out.get("timed_region_reps") or {}
        line["timed_region_us"].update({k: rs.get(k) for k in ("reps", "median_us", "min_us", "max_us") if rs.get(k) is not None})
    ss = out.get("steady_state_720_ticks")
    if ss:
        line["steady_state_us_per_tick"] = {m: v.get("us_per_tick") for m, v in ss.items()}
    line["extra"] = extra_path
    text = json.dumps(_sig(line), separators=(",", ":"))
    if len(text) >= LINE_MAX_BYTES:          # never let prose push the numbers out of the driver's window
        line["config"] = dict(line["config"], workload=str(line["config"].get("workload"))[:200])
        if line.get("cpu_baseline"):
            line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample"))[:120]
        line["mode_fallback"] = str(line["mode_fallback"])[:120] if line.get("mode_fallback") else line.get("mode_fallback")
        text = json.dumps(_sig(line), separators=(",", ":"))
    assert len(text) < LINE_MAX_BYTES, len(text)
    return text


# ---- CPU placement of a rank: the cores of its GPU's NUMA node -----------------------------------------------
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_cpus_of_pci(pci_bus_id, sysfs="/sys"):
    """(numa_node, set of CPUs) of the PCI device `dddd:bb:dd.f` from sysfs; (-1, None) when the platform does not say."""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return -1, None
        with open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        return node, (cpus or None)
    except (OSError, ValueError):
        return -1, None


def pin_rank(pci_bus_id, local_rank, local_world, sysfs="/sys", apply=True):
    """Pin this process to the cores of its GPU's NUMA node, split among the ranks that share the node so that no two ranks compete
    for a core (a 20-tick timed region is ~50 us of host work: one rank on the wrong socket, or two on one core, sets the MAX over
    ranks).  Without NUMA information the allowed CPUs are split evenly by local rank.  Returns what was done (for `per_rank`)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return {"pinned": False, "reason": "sched_getaffinity unavailable"}
    node, cpus = numa_cpus_of_pci(pci_bus_id, sysfs) if pci_bus_id else (-1, None)
    pool = sorted(set(allowed) & cpus) if cpus else []
    how = f"numa node {node}"
    if not pool:
        pool, how = allowed, "no NUMA information: even split of the allowed CPUs"
    share = max(1, len(pool) // max(1, local_world))
    mine = pool[(local_rank % max(1, local_world)) * share:][:share] or pool
    info = {"pinned": False, "numa_node": node, "how": how, "cpus": f"{mine[0]}-{mine[-1]}" if mine else "", "n_cpus": len(mine)}
    if apply and os.environ.get("Q1_BENCH_NO_PIN") != "1":
        try:
            os.sched_setaffinity(0, mine)
            info["pinned"] = True
        except OSError as ex:
            info["reason"] = repr(ex)
    return info


def device_pci_bus_id(dev_index):
    """PCI bus id of HIP device `dev_index` ("0000:05:00.0"), through torch's device properties or the HIP runtime torch loaded."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev_index)
        if hasattr(pr, "pci_bus_id") and hasattr(pr, "pci_device_id") and hasattr(pr, "pci_domain_id"):
            return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:   # noqa: BLE001
        pass
    try:
        import ctypes
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                hip = ctypes.CDLL(name)
                break
            except OSError:
                hip = None
        if hip is None:
            return None
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(dev_index)) == 0:
            return buf.value.decode().lower()
    except Exception:   # noqa: BLE001
        pass
    return None


SERVER_AUTO_MAX_ENVS = 294912      # auto mode: resident tick server up to its resident capacity on an MI355X, per-tick kernels above


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7200)
    ap.add_argument("--warmup", type=int, default=720)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU (131072 = BASELINE configs[3]'s shard)")
    ap.add_argument("--mode", choices=("auto", "step", "rollout", "server"), default="auto",
                    help="auto = rollout (one launch per episode chunk, every tick's obs / reward / done written to HBM)")
    ap.add_argument("--config", choices=("default", "params_yml"), default="default",
                    help="default = get_default with zero_start_prob 1 (BASELINE configs[1] / [3]); params_yml = the reference's training Config "
                         "(data/params.yml:16-33: random starts, truncated dt, action_range 10) with IN-KERNEL reset of finished episodes "
                         "(BASELINE configs[2]; --mode rollout only)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary (other-mode) measurement")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_workers(args, argv):
    """`python bench.py --gpus N` outside a distributed launcher: start N copies of this script, one per GPU, with the
    environment torch.distributed.run would have given them.  Rank 0's stdout (the JSON line) is this process's stdout; the other
    ranks' stdout goes to stderr.  Exit status: the first non-zero worker status."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), Q1_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = None
    alive = list(procs)
    while alive:
        for p in list(alive):
            st = p.poll()
            if st is None:
                continue
            alive.remove(p)
            if st != 0 and rc == 0:
                rc = st
                deadline = time.time() + 30.0          # a rank died: the others would wait in a barrier forever
        if deadline is not None and time.time() > deadline:
            for p in alive:
                p.kill()
        time.sleep(0.05)
    return rc


def load_env_class():
    """The per-GPU env handle and its import path (printed as `env_impl`).  Q1_BENCH_ENV_FACTORY = "module:attr" exists for
    tests/test_bench_launcher.py only: this container has no GPU, so the world-size-2 launcher test substitutes an oracle-backed
    stand-in with DeviceEnv's methods and runs every other line of this file (launcher, rendezvous, timed region, rank reduction,
    JSON) on CPU over gloo.  A bench whose timed region could silently execute the oracle would be worthless, so the variable is
    REFUSED unless Q1_BENCH_ALLOW_FAKE=1 is set as well, and the JSON line names the class that ran either way."""
    spec = os.environ.get("Q1_BENCH_ENV_FACTORY")
    if spec:
        if os.environ.get("Q1_BENCH_ALLOW_FAKE") != "1":
            raise SystemExit("bench.py: Q1_BENCH_ENV_FACTORY is set but Q1_BENCH_ALLOW_FAKE=1 is not - refusing to time anything but "
                             "q1physrl_amd.device.DeviceEnv (the variable exists for the CPU launcher tests only)")
        mod, attr = spec.split(":")
        return getattr(importlib.import_module(mod), attr), True, f"{mod}.{attr} (INJECTED through Q1_BENCH_ENV_FACTORY: not the HIP path)"
    from q1physrl_amd.device import DeviceEnv
    return DeviceEnv, False, "q1physrl_amd.device.DeviceEnv"


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_workers(args, argv))

    import torch
    import torch.distributed as dist
    from q1physrl_amd import _lib, env as E, sharding
    DeviceEnv, injected, env_impl = load_env_class()
    # which binary ran: sha256 of the loaded libq1env.so and the source hash compiled into it (q1physrl_amd/build.py sources_sha16)
    try:
        lib_sha16, lib_build_id = _lib.lib_sha16(), _lib.build_id()
    except Exception as ex:   # noqa: BLE001 - only possible with the injected CPU stand-in on a box without the library
        if not injected:
            raise
        lib_sha16, lib_build_id = None, f"unavailable ({ex!r})"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, "
                         f"or run plain `python bench.py --gpus {args.gpus}` and let it start its own workers)")
    if injected:
        dev_index, ndev = 0, 0
        d = torch.device("cpu")
        dsync = lambda: None                                         # noqa: E731
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
        ndev = torch.cuda.device_count()
        if world > ndev and not os.environ.get("Q1_BENCH_OVERSUBSCRIBE"):
            raise SystemExit(f"bench.py: --gpus {world} but only {ndev} HIP device(s) visible")
        dev_index = local_rank % ndev
        torch.cuda.set_device(dev_index)
        d = torch.device("cuda", dev_index)
        dsync = torch.cuda.synchronize
    # CPU placement: this rank's host thread (launch calls, the one synchronisation) on the cores next to its GPU
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    pci = None if injected else device_pci_bus_id(dev_index)
    placement = pin_rank(pci, local_rank, local_world) if world > 1 or os.environ.get("Q1_BENCH_PIN") == "1" else {"pinned": False, "how": "single rank: not pinned"}
    placement["device"] = dev_index
    placement["pci_bus_id"] = pci
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the env path has no collective: ranks only meet in barriers and one MAX of a scalar, which gloo serves from the host
        # (no GPU all-reduce latency inside the timed region's brackets); Q1_BENCH_BACKEND=nccl selects RCCL instead
        backend = os.environ.get("Q1_BENCH_BACKEND", "gloo")
        # rank 0's stdout carries exactly ONE line, the JSON: the native libraries' connection banners ("[Gloo] Rank 0 is connected
        # to ..." goes to the C++ stdout) are sent to stderr while the process group comes up
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=d)
            else:
                dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    n = args.envs
    start, _ = sharding.shard_range(n * world, rank, world)                  # contiguous batch split, weak scaling
    full_cfg = args.config == "params_yml"
    if full_cfg and args.mode not in ("auto", "rollout"):
        raise SystemExit("bench.py: --config params_yml is measured with --mode rollout (in-kernel reset of finished episodes)")
    if full_cfg:
        cfg = E.Config(num_envs=n, **PARAMS_YML)
    else:
        cfg = E.Config(**{**E.Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    dev = DeviceEnv(cfg, device=dev_index, env_index_base=start)             # own stream; global env index keys the RNG
    if full_cfg:
        dev.reset_philox_dev(99, 0, False)                                   # random starts (env.py:428-455's distributions, counter RNG)
    ar = float(cfg.action_range)
    keys_h, mouse_h = make_actions(n, EPISODE_TICKS, ar, seed=1234 + rank)
    keys = torch.from_numpy(keys_h).to(d)
    mouse = torch.from_numpy(mouse_h).to(d)
    obs1 = torch.empty((n, 6), dtype=torch.float32, device=d)
    rew1 = torch.empty((n,), dtype=torch.float32, device=d)
    done1 = torch.empty((n,), dtype=torch.uint8, device=d)
    obsT = rewT = doneT = None
    srv = None                                   # buffers of the resident tick server (--mode server), made on first use
    dsync()

    def server_state():
        nonlocal srv
        if srv is None:
            srv = {"mailbox": torch.zeros((n,), dtype=torch.int64, device=d), "results": torch.zeros((4, n, 2), dtype=torch.int64, device=d),
                   "status": torch.zeros((5,), dtype=torch.int32, device=d), "tag": 0,
                   # (a second stream only for the two-stream form: every extra stream is one more queue a device-wide sync visits)
                   "stream": torch.cuda.Stream(device=d, priority=-1) if os.environ.get("Q1_BENCH_SERVER_TWO_STREAMS") and not injected else None}
            dsync()
        return srv

    def plan_ticks(mode, k, tick0, prepare=False, timed=False):
        """The calls that run k ticks starting at episode phase tick0 % 720 (all envs are reset on device at each episode end),
        as a list of zero-argument callables with every pointer and flag already resolved - so that the timed region is
        nothing but the calls themselves - plus the number of kernel launches they make.
        prepare=True: only build (instantiate + upload) the hipGraphs the sequence replays, no launch.
        timed=True: bracket the sequence with the handle's timer events (HIP events on the launch stream); in step mode they are
        recorded inside the q1env_step_many call of the first / last chunk, next to the launch itself."""
        nonlocal obsT, rewT, doneT
        if mode == "rollout" and obsT is None:        # tick-major per-tick outputs of a whole episode
            obsT = torch.empty((EPISODE_TICKS, n, 6), dtype=torch.float32, device=d)
            rewT = torch.empty((EPISODE_TICKS, n), dtype=torch.float32, device=d)
            doneT = torch.empty((EPISODE_TICKS, n), dtype=torch.uint8, device=d)
        calls = []
        waits_itself = False
        t, left, launches = tick0, k, 0
        started = stopped = False
        o1, r1, d1 = obs1.data_ptr(), rew1.data_ptr(), done1.data_ptr()
        while left > 0:
            ph = t % EPISODE_TICKS
            chunk = min(left, EPISODE_TICKS - ph)
            ka = keys.data_ptr() + ph * n
            ma = mouse.data_ptr() + ph * n * 4
            ends_episode = (t + chunk) % EPISODE_TICKS == 0 and not full_cfg    # (params_yml: episodes end per env, reset in-kernel)
            if mode == "server":
                # the resident tick server on the handle's stream + its dependent producer on a second stream: `chunk` ticks,
                # one launch each; the in-kernel reset at the episode-ending tick replaces the reset launch of the other modes
                if not prepare:
                    sv = server_state()
                    ps = 1 if sv["stream"] is None else sv["stream"].cuda_stream
                    two = bool(os.environ.get("Q1_BENCH_SERVER_TWO_STREAMS"))
                    pair_flags = 1                                         # bit 0: in-kernel reset of finished episodes
                    if timed and not started:
                        if two:
                            calls.append(dev.timer_start)
                        else:
                            pair_flags |= _lib.TIMER_START
                        started = True
                    if timed and left == chunk and not two:
                        pair_flags |= _lib.TIMER_STOP
                        stopped = True
                    if two:                                                # producer on its own (high-priority) stream
                        calls.append(functools.partial(dev.persistent_start, chunk, sv["tag"], sv["mailbox"].data_ptr(), sv["results"].data_ptr(),
                                                       o1, 99, True, sv["status"].data_ptr(), 2.0))
                        calls.append(functools.partial(dev.persistent_drive, ps, chunk, sv["tag"], ka, ma, sv["mailbox"].data_ptr(),
                                                       sv["results"].data_ptr(), 0, sv["status"].data_ptr(), 2.0))
                    else:                                                  # server + producer as one dispatch: co-resident by construction
                        calls.append(functools.partial(dev.persistent_pair, chunk, sv["tag"], ka, ma, sv["mailbox"].data_ptr(),
                                                       sv["results"].data_ptr(), o1, 99, pair_flags, 0, sv["status"].data_ptr(), 2.0))
                    sv["tag"] = (sv["tag"] + chunk) % 0xFFFFFF
                launches += 1
                t += chunk
                left -= chunk
                continue
            if mode == "step":
                if prepare:
                    if not args.no_graph:
                        calls.append(functools.partial(dev.step_many_dev, chunk, _lib.ACT_PACKED, ka, ma, _lib.OBS_F32, o1, r1, d1, 0, 2))
                else:
                    flags = 0 if args.no_graph else 1
                    if timed and not started:
                        flags |= _lib.TIMER_START
                        started = True
                    if timed and left == chunk and not ends_episode:
                        flags |= _lib.TIMER_STOP
                        stopped = True
                    calls.append(functools.partial(dev.step_many_dev, chunk, _lib.ACT_PACKED, ka, ma, _lib.OBS_F32, o1, r1, d1, 0, flags))
                launches += chunk
            else:
                if not prepare:
                    # (the timer events are recorded inside the q1env_rollout call of the first / last chunk, next to the launch itself)
                    # timed = "signal": device stamps + the kernel-written completion signal (the timed region proper);
                    # timed = "events": HIP events around the same launches (the cross-check pass that follows it)
                    rflags = 1 if full_cfg else 0                          # bit 0: in-kernel Philox reset of envs whose episode ended
                    sig = timed in ("signal", "signal_wait")
                    if timed and not started:
                        rflags |= _lib.STAMP_START if sig else _lib.TIMER_START
                        started = True
                    if timed and left == chunk and not ends_episode and not (sig and os.environ.get("Q1_BENCH_SIGNAL_MARK") == "1"):
                        # (Q1_BENCH_SIGNAL_MARK=1: A/B knob - the signal from a one-wave kernel behind the launch instead of its own last wave)
                        # "signal_wait": the launching call polls for the signal itself (Q1ENV_SIGNAL_WAIT: launch + wait in ONE call across the ABI)
                        rflags |= (_lib.SIGNAL_WAIT if timed == "signal_wait" else _lib.SIGNAL) if sig else _lib.TIMER_STOP
                        stopped = True
                        waits_itself = timed == "signal_wait"
                    # (arguments converted once, outside the timed region: DeviceEnv.prepare_rollout)
                    calls.append(dev.prepare_rollout(chunk, _lib.ACT_PACKED, ka, ma, 99 if full_cfg else 0, _lib.OBS_F32, obsT.data_ptr(),
                                                     rewT.data_ptr(), doneT.data_ptr(), rflags))
                launches += 1
            t += chunk
            left -= chunk
            if ends_episode and not prepare:
                calls.append(functools.partial(dev.reset_philox_dev, 99, 0, True))   # zero_start_prob = 1: every env back to the start line
        if timed and not stopped:
            calls.append(dev.signal_mark if timed in ("signal", "signal_wait") else dev.timer_mark)
        plan_ticks.waits_itself = waits_itself
        return calls, launches

    def run(calls):
        for f in calls:
            f()

    def barrier():
        dev.sync()
        dsync()
        if world > 1:
            dist.barrier()

    host_split = {}                              # per mode: host time spent enqueueing the timed region / waiting in the synchronisation

    def agree(ok):
        """Collective AND over the ranks: every rank reaches the same verdict at the same point, so a failure on one GPU can never
        leave the others waiting in a barrier."""
        if world == 1:
            return bool(ok)
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        return all(flags)

    def server_ok():
        return not server_state()["status"].cpu().numpy().any()      # accumulated since it was zeroed at set-up; written on failure only

    region_reps = {}                              # mode -> per-repetition wall seconds of the --steps region (this rank's)

    def measure(mode, steps, warmup, reps=0):
        # Preparation (untimed): instantiate + upload every graph the two sequences replay, and replay them ONCE with the env
        # state saved and restored around it, so that neither the warm-up nor the timed region pays a first-replay cost (kernel
        # code objects, the graph's packets, the TLB entries of the action slabs) - in production a graph is replayed thousands
        # of times.  The env state the W warm-up ticks start from is the state before this preparation.
        # (server mode: the plans hold the hand-off tags, so each is made right before it runs, in execution order)
        err = None
        try:
            run(plan_ticks(mode, warmup, 0, prepare=True)[0])
            run(plan_ticks(mode, steps, warmup, prepare=True)[0])
            dev.snapshot_state()
            run(plan_ticks(mode, warmup, 0)[0])
            run(plan_ticks(mode, steps, warmup)[0])
        except Exception as ex:   # noqa: BLE001 - e.g. the tick-server pair does not fit this device: reported collectively below
            err = ex
        barrier()
        if not agree(err is None and (mode != "server" or server_ok())):
            raise RuntimeError(f"{mode} mode failed in the dry run on at least one rank" + (f": {err!r}" if err is not None else
                               " (tick server / producer timed out)" if mode == "server" else ""))
        dev.restore_state()
        signalled = mode == "rollout" and hasattr(dev, "signal_wait") and os.environ.get("Q1_BENCH_NO_SIGNAL") != "1"
        if signalled:                             # the signal's pinned words and ticket counters exist and have been touched before the region -
            dev.signal_mark()                     # and BEFORE the warm-up ticks: the runtime enqueues a kernel 1.5-2 us faster when the
            dev.signal_wait()                     # previous launch was the same kernel (profiles/r4_bench_probe_host_knobs.txt)
        # every call of the timed region is resolved BEFORE the warm-up ticks, so that nothing but the synchronisation sits between the
        # warm-up's last kernel and the timed launch: a queue that has been idle takes longer to start a kernel (Q1_BENCH_SPIN_US: 300 us of
        # host-only spinning between the synchronisation and t0 cost 5-6 us of the 26; profiles/r4_bench_probe_host_knobs.txt)
        timed_calls, launches = plan_ticks(mode, steps, warmup, timed="signal" if signalled else "events")
        wait = dev.signal_wait if signalled else None
        # a region that is ONE launch (the driver-style 20-tick line) runs as one call across the ABI: launch + poll inside q1env_rollout
        # (Q1ENV_SIGNAL_WAIT) instead of two calls from Python; the enqueue / wait split is then taken in the cross-check pass below
        one_call = signalled and launches == 1 and len(timed_calls) == 1 and os.environ.get("Q1_BENCH_TWO_CALLS") != "1"
        if one_call:
            timed_calls, launches = plan_ticks(mode, steps, warmup, timed="signal_wait")
            one_call = plan_ticks.waits_itself and len(timed_calls) == 1
            if not one_call:
                timed_calls, launches = plan_ticks(mode, steps, warmup, timed="signal")
        warm_calls = plan_ticks(mode, warmup, 0)[0]
        run(warm_calls)                           # the W untimed warm-up ticks
        barrier()
        spin_us = float(os.environ.get("Q1_BENCH_SPIN_US", "0"))
        if spin_us > 0:                           # A/B knob: keep the host core busy between the synchronisation and t0 (no GPU work)
            ts = time.perf_counter()
            while (time.perf_counter() - ts) * 1e6 < spin_us:
                pass
        if signalled:
            if one_call:
                f = timed_calls[0]
                t0 = time.perf_counter()
                f()                               # EXACTLY `steps` ticks in one launch; the call returns when the kernel's signal has arrived
                t1 = time.perf_counter()
                t_enq = None
            else:
                t0 = time.perf_counter()
                run(timed_calls)                  # EXACTLY `steps` ticks: first wave stamps the start, last wave stamps the end + signals
                t_enq = time.perf_counter()
                wait()                            # poll the host-coherent sequence word the kernel writes (no runtime synchronisation)
                t1 = time.perf_counter()
            dsync()                               # untimed: nothing else may have been in flight (and nothing was: see post_sync_us)
            t2 = time.perf_counter()
            own = t1 - t0
            stamp_ms = dev.signal_elapsed() * 1e3
            host_split[mode] = {"launch_to_signal_seen_us": (t1 - t0) * 1e6,
                                "post_sync_us": (t2 - t1) * 1e6, "wall_incl_runtime_sync_us": (t2 - t0) * 1e6,
                                "device_stamp_us": stamp_ms * 1e3,
                                "completion": "kernel-written signal, results written through at system scope (readable by any agent when it arrives: "
                                              "tests/test_hip_signal.py), polled by the host" + (" inside the launching call" if one_call else "")}
            if t_enq is not None:
                host_split[mode]["enqueue_us"] = (t_enq - t0) * 1e6
                host_split[mode]["enqueue_to_signal_seen_us"] = (t1 - t_enq) * 1e6
            # the same K ticks again, from the same state, between two HIP events on the launch stream (what `roofline` uses)
            dev.restore_state()
            run(plan_ticks(mode, warmup, 0)[0])
            ev_calls, _l = plan_ticks(mode, steps, warmup, timed="events")
            barrier()
            te0 = time.perf_counter()
            run(ev_calls)
            te1 = time.perf_counter()
            dsync()
            ev_ms = dev.timer_elapsed()
            host_split[mode]["hip_event_us"] = ev_ms * 1e3
            if one_call:
                # the enqueue / wait split of the same launch as TWO calls from Python (q1env_rollout with Q1ENV_SIGNAL, then q1env_signal_wait),
                # taken here, outside the timed region: what the one-call form saves is two_call_wall_us - launch_to_signal_seen_us
                host_split[mode]["enqueue_us_events_pass"] = (te1 - te0) * 1e6
                dev.restore_state()
                run(plan_ticks(mode, warmup, 0)[0])
                two_calls, _l = plan_ticks(mode, steps, warmup, timed="signal")
                barrier()
                ta = time.perf_counter()
                run(two_calls)
                tb = time.perf_counter()
                wait()
                tc = time.perf_counter()
                dsync()
                host_split[mode]["enqueue_us"] = (tb - ta) * 1e6
                host_split[mode]["enqueue_to_signal_seen_us"] = (tc - tb) * 1e6
                host_split[mode]["two_call_wall_us"] = (tc - ta) * 1e6
                host_split[mode]["two_call_device_stamp_us"] = dev.signal_elapsed() * 1e6
                host_split[mode]["enqueue_us_source"] = "a repetition of the timed launch as two calls, right after the timed region"
            # what an EMPTY signalled launch costs here (one-wave kernel that only stamps and signals, then the same poll): the floor
            # under (wall - device_stamp_us) that no kernel of ours can go below on this runtime / firmware
            rt = []
            for _ in range(7):
                dsync()
                ta = time.perf_counter()
                dev.signal_mark()
                dev.signal_wait()
                rt.append((time.perf_counter() - ta) * 1e6)
            host_split[mode]["empty_signalled_launch_roundtrip_us"] = sorted(rt)[len(rt) // 2]
        else:
            t0 = time.perf_counter()
            run(timed_calls)                      # EXACTLY `steps` ticks; HIP events on the launch stream around them
            t_enq = time.perf_counter()
            dsync()                               # the ONE synchronisation that ends the timed region (device-wide: covers both streams)
            own = time.perf_counter() - t0
            host_split[mode] = {"enqueue_us": (t_enq - t0) * 1e6, "enqueue_to_sync_return_us": (t0 + own - t_enq) * 1e6,
                                "completion": "torch.cuda.synchronize()"}
            ev_ms = dev.timer_elapsed()           # both events have completed: no further wait
        # The contract's `value` is the ONE region above.  A 20-tick region is ~28 us: one sample of launch jitter - so the same region is
        # repeated `reps` times (state restored, the W warm-up ticks, a barrier, the same calls and the same completion criterion) and the
        # line carries median / min / max next to the single shot.
        if reps > 0:
            kind = ("signal_wait" if one_call else "signal") if signalled else "events"
            times = []
            for _ in range(reps):
                dev.restore_state()
                run(plan_ticks(mode, warmup, 0)[0])
                calls_r, _l = plan_ticks(mode, steps, warmup, timed=kind)
                barrier()
                ta = time.perf_counter()
                run(calls_r)
                if signalled and not one_call:
                    wait()
                elif not signalled:
                    dsync()
                tb = time.perf_counter()
                dsync()
                times.append(tb - ta)
            region_reps[mode] = times
        if world > 1:
            dist.barrier()
        if not agree(mode != "server" or server_ok()):
            raise RuntimeError("tick server / producer timed out in the timed region on at least one rank")
        wall = sharding.max_over_ranks(own, device=d)
        return wall, ev_ms, launches, own

    def per_rank(own, ev_ms, steps, mode):
        """Every rank's own wall / event time of the region it timed, its host-side split, device and CPU placement, gathered on all
        ranks (host objects, after the timing)."""
        mine = {"rank": rank, "wall_ms": own * 1e3, "event_ms": ev_ms, "env_steps_per_s": float(n) * steps / own, "mode": mode,
                "env_index_base": int(start), "envs": int(n), "host_split_us": host_split.get(mode), "placement": placement,
                "env_impl": env_impl}
        if world == 1:
            return [mine]
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        return rows

    fallback = None
    if args.mode == "auto":
        args.mode = PRIMARY_AUTO
        try:
            wall, ev_ms, launches, own = measure(args.mode, args.steps, args.warmup, reps=REGION_REPS)
        except Exception as ex:   # noqa: BLE001 - measure() reaches its verdict collectively: every rank falls back at the same point
            fallback = f"{args.mode} mode failed ({ex!r}); measured with per-tick launches instead"
            sys.stderr.write("bench.py: " + fallback + "\n")
            args.mode = "step"
            wall, ev_ms, launches, own = measure(args.mode, args.steps, args.warmup, reps=REGION_REPS)
    else:
        wall, ev_ms, launches, own = measure(args.mode, args.steps, args.warmup, reps=REGION_REPS)
    ranks = per_rank(own, ev_ms, args.steps, args.mode)
    value = float(n) * args.steps * world / wall
    # the repetitions of the same region: per repetition the slowest rank (what `value` is built on), then median / min / max over the repetitions
    reps_stat = None
    if region_reps.get(args.mode):
        tt = torch.tensor(region_reps[args.mode], dtype=torch.float64)
        if world > 1:
            tt = tt.to(d if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tt = tt.cpu()
        us = sorted(float(x) * 1e6 for x in tt)
        reps_stat = {"reps": len(us), "median_us": us[len(us) // 2], "min_us": us[0], "max_us": us[-1],
                     "median_env_steps_per_s": float(n) * args.steps * world / (us[len(us) // 2] * 1e-6),
                     "what": f"the --steps region repeated {len(us)} times from the same state (state restored + the W warm-up ticks before each), same calls and "
                             "completion criterion as the single region `value` is built on; per repetition the slowest rank"}
    spec_cfg = True                                   # get_default's action / episode structure: the SPEC = true instantiations

    def pair_es(n_envs):
        """sub-batches of 64 envs per workgroup the pair kernel runs with at this size (q1env_step_persistent_pair picks the
        smallest whose grid is resident: 256 CUs x 8 workgroups x 64 x ES on an MI355X)"""
        return 1 if n_envs <= 131072 else (2 if n_envs <= 262144 else 3)

    def kernel_name(mode, ticks_per_launch=None):
        sp = "true" if spec_cfg else "false"
        # rollout: the action is requested two ticks ahead from 32 ticks per launch (q1env_core.hip ROLLOUT_DEPTH2_MIN_TICKS), one below
        depth = 2 if (ticks_per_launch or 0) >= 32 else 1
        return {"step": f"step_kernel<float, {sp}, 2>  (OBS_T = float, SPEC, FMT_PACKED)",
                "rollout": f"rollout_kernel<float, {sp}, 2, {'true' if full_cfg else 'false'}, 1, false, {depth}>  (OBS_T = float, SPEC, FMT_PACKED, HAS_RESET = "
                           f"{'true' if full_cfg else 'false'}, OUT_MODE = 1: obs, reward, done every tick, RET = false, DEPTH = action prefetch distance)",
                "server": f"tick_pair_lds_kernel<{sp}, {pair_es(n)}>  (SPEC, ES = sub-batches per workgroup; server wave + dependent stand-in producer wave)"}[mode]

    def workload(mode):
        """one clause per item (the full description of each mode is this file's docstring)"""
        if full_cfg:
            head = f"BASELINE configs[2]: {n} envs/GPU, params.yml Config, random starts, random packed actions resident in HBM, finished episodes reset IN-KERNEL; "
        else:
            head = (f"BASELINE configs[1]: {n} envs/GPU" if n != 131072 else
                    f"BASELINE configs[3] shard: 131072 envs/GPU ({131072 * world} envs on {world} GPU(s))")
            head += ", zero-start 100 m run, get_default Config, random packed actions resident in HBM, on-device reset at each 720-tick episode end; "
        return head + {
            "rollout": "mode=rollout: q1env_rollout, one launch per episode chunk, state in registers, every tick's obs f32 (N,6) / reward f32 / "
                       "done u8 written tick-major to HBM",
            "step": "mode=step" + ("+hipGraph" if not args.no_graph else "") + ": one step_kernel launch per tick (SoA state read + written every "
                    "tick), every tick's obs / reward / done written to HBM",
            "server": "mode=server: resident tick server + dependent stand-in producer as one dispatch, LDS hand-offs; per-tick outputs are "
                      "NOT written to HBM (only the last tick's)"
                      if not os.environ.get("Q1_BENCH_SERVER_TWO_STREAMS") else
                      "mode=server (two streams): tick server + producer kernel on a second stream, results cross as 8-byte tagged granules through L2 / HBM"}[mode]

    def roofline(mode, steps, launches_, ev_ms_, wall_):
        """The roofline statement of `mode`'s dominant kernel for a region of `steps` ticks in `launches_` launches and ev_ms_ of HIP-event
        time on the launch stream.  Always `bound: hbm` (the path has no dense contraction):
            achieved = ALGORITHMIC bytes per launch / average launch duration, frac = achieved / 8 TB/s
            traffic  = MEASURED HBM bytes per launch (rocprofv3 PMC, profiles/pmc.json) - null when that file was taken from another
                       build of the library (`pmc_stale`) or does not hold this mode / size
        step:             204 B per env-step (reads + writes the whole SoA state every tick).
        rollout / server: the state stays in registers between ticks, so a launch of T ticks must move 34 B per env-step (5 B action
                          in, 29 B obs / reward / done out) + the 170 B state once: that is the algorithmic figure, and `frac` says how
                          far from the HBM roof the kernel is.  WHY it is that far - float64 VALU issue, one wave per SIMD - is in the
                          `valu` sub-object (counters from the same PMC file, null when stale); frac_nominal_204B keeps the per-tick
                          formulation's bytes on this kernel's time (passes 1 by construction: NOT a roofline fraction)."""
        tpl = steps / launches_
        kern_us = ev_ms_ * 1e3 / launches_
        pmc, pmc_stale = load_pmc(mode + ("_params" if full_cfg else ""), n, lib_build_id)
        resident = mode != "step"
        traffic = traffic_per_launch(pmc, n, tpl, resident)
        alg_bytes = (B_ALG * n * tpl) if not resident else (B_FUSED * n * tpl + B_STATE * n)
        achieved = alg_bytes / (kern_us * 1e-6) / 1e9
        nominal = B_ALG * n * tpl / (kern_us * 1e-6) / 1e9
        hs = host_split.get(mode) or {}
        stamp_us = hs.get("device_stamp_us")
        # `bound`: what limits the kernel.  The per-tick step kernel moves the whole state every launch: memory ("hbm").  The register-resident
        # kernels are bound by float64 VALU issue at one wave per SIMD (`valu`, profiles/r5_tick_floor.txt) - `frac` is still their distance from
        # the HBM roof (`frac_axis`), which is the axis SURVEY 8(d) prescribes; frac_8d_204B prices the same launch at 8(d)'s literal 204 B per
        # env-step (> 1 for a register-resident kernel: the 170 B of state it does not move - not a fraction of anything).
        r = {"bound": "hbm" if not resident else "valu_f64", "frac_axis": "hbm", "frac_8d_204B": nominal / HBM_PEAK_GBPS, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
             "kernel": kernel_name(mode, tpl), "avg_launch_us": kern_us, "ticks_per_launch": tpl, "launches": launches_,
             "algorithmic_bytes_per_launch": alg_bytes,
             "alg_bytes_per_env_step": B_ALG if not resident else B_FUSED, "alg_bytes_per_env_per_launch": 0.0 if not resident else B_STATE,
             "event_ms_per_step": ev_ms_ / steps, "us_per_tick": ev_ms_ * 1e3 / steps,
             "launch_duration_source": "HIP events on the launch stream" + (" (the K ticks repeated from the same state right after the timed "
                                       "region; the timed region itself carries no marker packets - its kernel time is device_stamp_us)" if stamp_us else ""),
             "device_stamp_us": stamp_us,
             # the same roofline arithmetic on the timed region's OWN kernel time (device stamps; for a launch this short it is what
             # rocprofv3's kernel-trace reports - profiles/r4_driver_steps20_kernel_trace.txt - while two HIP-event marker packets add ~3.5 us)
             "avg_launch_us_device_stamps": (stamp_us / launches_) if stamp_us else None,
             "achieved_device_stamps": (alg_bytes / (stamp_us / launches_ * 1e-6) / 1e9) if stamp_us else None,
             "frac_device_stamps": (alg_bytes / (stamp_us / launches_ * 1e-6) / 1e9 / HBM_PEAK_GBPS) if stamp_us else None,
             "wall_over_event": (wall_ * 1e6 / stamp_us) if stamp_us else (wall_ * 1e3 / ev_ms_ if ev_ms_ > 0 else None),
             "wall_over_event_basis": "device stamps of the timed region" if stamp_us else "HIP events of the timed region",
             # the same ratio on round 3's basis (2.14 then): wall of the timed region over the HIP-event time of the same K ticks (cross-check pass)
             "wall_over_hip_event": (wall_ * 1e3 / ev_ms_) if ev_ms_ and ev_ms_ > 0 else None,
             "host_split_us": hs or None, "env_steps_per_launch": n * tpl, "frac_nominal_204B": nominal / HBM_PEAK_GBPS,
             "traffic_frac_of_peak": (traffic / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if traffic is not None else None,
             "traffic_over_algorithmic": (traffic / alg_bytes) if traffic is not None else None,
             "pmc_stale": pmc_stale, "pmc_build_id": (pmc or {}).get("build_id"), "lib_build_id": lib_build_id,
             "profile": (pmc or {}).get("source")}
        if not resident:
            r["note"] = ("per-tick formulation: every launch reads and writes the whole SoA state (204 B per env-step).  At 65 536 envs a launch "
                         "moves 13 MB (1.7 us at 8 TB/s) behind a ~1.8 us dependent-dispatch boundary: latency-bound by construction "
                         "(DESIGN.md 6.1); the size sweep shows the same kernel at large batches.")
            return r
        simds = 1024.0                                  # 256 CUs x 4 SIMDs
        valu = None
        if pmc is not None and pmc.get("valu_busy_cycles"):
            t0_ = float(pmc.get("ticks_per_launch", 1))
            waves = max(float(pmc.get("waves", 1.0)), 1.0)
            busy = float(pmc["valu_busy_cycles"]) * tpl / t0_                                            # per launch of tpl ticks
            insts = float(pmc.get("insts_valu", 0.0)) / waves / t0_
            other = (float(pmc.get("insts_salu", 0.0)) + float(pmc.get("insts_mem", 0.0))) / waves / t0_
            valu = {"valu_busy_Gcycles_per_s": busy / (kern_us * 1e-6) / 1e9, "chip_Gcycles_per_s": simds * PEAK_CLOCK_GHZ,
                    "valu_busy_frac": busy / (kern_us * 1e-6) / 1e9 / (simds * PEAK_CLOCK_GHZ),
                    "valu_insts_per_tick_per_wave": insts, "salu_and_memory_insts_per_tick_per_wave": other}
            # the issue limit of a LONE wave on its SIMD (what 65 536 envs on 1 024 SIMDs are): one instruction per 5.17 cycles whatever
            # its type - vector, scalar or memory (tools/ubench_f64.hip, tools/ubench_select.hip; 16.8 for a float64 transcendental)
            if max(1.0, n / 64.0 / simds) <= 1.0 and insts > 0:
                floor_us = (insts + other) * 5.17 / (PEAK_CLOCK_GHZ * 1e3)
                valu["lone_wave_issue_floor_us_per_tick"] = floor_us
                valu["frac_of_lone_wave_issue_floor"] = floor_us / (ev_ms_ * 1e3 / steps)
        r["valu"] = valu
        r["note"] = ("register-resident kernel: the env state stays in registers between ticks, so a launch must move 34 B per env-step (5 B action "
                     "in, 29 B obs / reward / done out) + the 170 B state once - `achieved` is that figure over the measured launch time, `frac` "
                     "its fraction of 8 TB/s, `traffic` the bytes the counters saw.  The kernel is NOT HBM-bound: its limiter is float64 VALU issue "
                     "with one wave per SIMD (`valu`: busy cycles = 4 x SQ_ACTIVE_INST_VALU, instruction counts per tick and wave, and the "
                     "lone-wave issue floor of tools/ubench_f64.hip).")
        if mode == "server":
            r["note"] += ("  server mode: per-tick outputs go to the stand-in producer wave through LDS and are not written to HBM; the tick-to-tick "
                          "latency additionally contains two LDS hand-offs.")
        return r

    roof = roofline(args.mode, args.steps, launches, ev_ms, wall)
    out = {
        "metric": "env-steps/sec @ 64k envs, 1/2/4/8 MI355X; max |pos - NumPy ref| over 10 s",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload(args.mode), "total_envs": n * world,
                   "envs_per_gpu": n, "parallelism": f"batch-split x{world}, no collective",
                   "arithmetic": "float64 (float32 storage of vel/obs/reward), bit-identical to the NumPy reference"},
        "roofline": roof,
        "mode": args.mode, "mode_fallback": fallback, "env_impl": env_impl, "lib_sha16": lib_sha16, "lib_build_id": lib_build_id,
        "ms_per_step_incl_runtime_sync": ((host_split.get(args.mode) or {}).get("wall_incl_runtime_sync_us") or wall * 1e6) / 1e3 / args.steps
                                         if world == 1 else None,
        "completion": (host_split.get(args.mode) or {}).get("completion"),
        "timed_region_reps": reps_stat,
        "per_rank": ranks,
        "parity": "max |pos - NumPy ref| over the 10 s rollout: measured live in cpu_baseline.parity_vs_gpu_after_719_ticks (N=1 runs); "
                  "tests/test_hip_fastpath.py::test_full_size_rollout_parity_65536_envs_720_ticks checks all 65 536 x 720 env-steps bit-exactly",
    }
    if not args.no_secondary:
        names = {"rollout": "fused_rollout", "step": "per_tick_step", "server": "persistent_server"}
        for other in ("step", "rollout", "server"):
            if other == args.mode or (other == "server" and (injected or n > SERVER_AUTO_MAX_ENVS)) or full_cfg:
                continue                                 # (params_yml: only the rollout with in-kernel reset is that workload)
            try:
                w2, ev2, l2, own2 = measure(other, args.steps, args.warmup)
            except Exception as ex:   # noqa: BLE001 - a secondary measurement must not take the contract line down
                out[names[other]] = {"error": repr(ex)}
                continue
            out[names[other]] = {"value": float(n) * args.steps * world / w2, "unit": "env-steps/s", "ms_per_step": w2 * 1e3 / args.steps,
                                 "launches": l2, "workload": workload(other), "roofline": roofline(other, args.steps, l2, ev2, w2)}
        # steady state (one whole 720-tick episode per measurement, HIP events: what a --steps 20 region cannot show - launch and
        # start-up latency amortised), on every rank: the multi-GPU line carries it too (slowest rank's event time)
        steady = {}
        for m in ("rollout", "step") + (() if injected or n > SERVER_AUTO_MAX_ENVS else ("server",)):
            if (injected and world > 1 and m != args.mode) or (full_cfg and m != "rollout"):
                continue                                 # (the CPU stand-in is slow: the launcher tests keep to the primary mode)
            try:
                w3, ev3, l3, _o = measure(m, EPISODE_TICKS, 0)
                ev_max = sharding.max_over_ranks(ev3 * 1e-3, device=d) * 1e3 if world > 1 else ev3
                steady[m] = {"us_per_tick": ev_max * 1e3 / EPISODE_TICKS, "env_steps_per_s": n * world / (ev_max * 1e-3 / EPISODE_TICKS),
                             "launches": l3, "roofline": roofline(m, EPISODE_TICKS, l3, ev_max, w3)}
            except Exception as ex:   # noqa: BLE001
                steady[m] = {"error": repr(ex)}
        out["steady_state_720_ticks"] = steady
    if rank == 0 and world == 1 and not args.no_secondary and not injected and not full_cfg:
        out["step_kernel_size_sweep"] = size_sweep(dev_index)
        try:
            out["sampler_configs4_shard"] = sampler_block(dev_index)
        except Exception as ex:   # noqa: BLE001 - extra information must not take the contract line down
            out["sampler_configs4_shard"] = {"error": repr(ex)}
    if rank == 0 and not args.no_cpu_baseline and not full_cfg:
        def gpu_check(acts, k):
            """The GPU env on the oracle's own actions (float64 rows) for k ticks from a fresh zero start."""
            chk = DeviceEnv(cfg, device=dev_index)
            rows = [torch.from_numpy(a).to(d) for a in acts]
            for t in range(k):
                chk.step_dev(_lib.ACT_F64_ROWS, rows[t % len(rows)].data_ptr(), 0, _lib.OBS_F32, 0, 0, 0, 0)
            chk.sync()
            st = chk.get_state()
            chk.close()
            return st
        # N > 1: timed on rank 0 AFTER every timed region, while the other ranks wait in the barrier below (their host threads sleep in the
        # collective; nothing of theirs runs on the GPUs) - a shorter sample, without the extras of the N = 1 run
        out["cpu_baseline"] = cpu_baseline(n, ar, budget_s=10.0 if world == 1 else 4.0, gpu_check=None if injected else gpu_check, extras=world == 1)
        if world > 1:
            out["cpu_baseline"]["sample"] += f"; rank 0 of {world}, after the timed regions, the other ranks parked in a barrier"
    elif args.no_cpu_baseline or full_cfg:
        out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
    if rank == 0:
        extra_path = write_extra(out)
        print(contract_line(out, extra_path), flush=True)
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
