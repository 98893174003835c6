"""GPU: bench.py as the driver runs it.  (a) the N=1 contract line: default mode, HIP path (env_impl), roofline object of the default
(register-resident) kernel and of the per-tick kernel; (b) VERDICT r2 item 2: `bench.py --gpus 2` self-launched and OVERSUBSCRIBED
on the one device of this box - both ranks run the same (default) mode on the HIP path, own distinct env_index_base, report their own
host split and placement, nobody falls back."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "Q1_BENCH_ENV_FACTORY", "Q1_BENCH_ALLOW_FAKE"):
        env.pop(k, None)
    env.update(extra_env or {})
    env["Q1_BENCH_EXTRA"] = os.path.join(tempfile.gettempdir(), f"q1_bench_extra_gpu_{os.getpid()}.json")
    r = subprocess.run([sys.executable, "bench.py"] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    # the driver parses the line from the last 8 KB of stdout - it must stay small at any rank count; everything
    # beyond the contract fields lives in the side file the line names.  Returned: the side file (a superset) + the line under "_line".
    assert len(lines[0]) < 4096, len(lines[0])
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "mode", "env_impl", "lib_sha16", "lib_build_id", "extra"):
        assert k in line, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "ticks_per_launch", "pmc_stale")) <= set(line["roofline"])
    with open(line["extra"]) as f:
        full = json.load(f)
    for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup", "mode"):
        assert full[k] == line[k] or abs(full[k] - line[k]) <= 1e-6 * abs(full[k]), k
    full["_line"] = line
    return full


def test_driver_line_single_gpu():
    d = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    assert d["env_impl"] == "q1physrl_amd.device.DeviceEnv" and d["mode"] == "rollout" and d["mode_fallback"] is None
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["config"]["envs_per_gpu"] == 65536 and "written tick-major to HBM" in d["config"]["workload"]
    ro = d["roofline"]
    assert ro["bound"] == "valu_f64" and ro["frac_axis"] == "hbm" and ro["kernel"].startswith("rollout_kernel<float, true, 2, false, 1, false, 1>")
    assert d["_line"]["roofline"]["bound"] == "valu_f64" and d["_line"]["roofline"]["frac_8d_204B"] > 1.0      # 8(d)'s 204 B on a register-resident launch: not a fraction
    # the single-shot region next to 15 repetitions of it
    tr = d["_line"]["timed_region_us"]
    assert tr["reps"] == 15 and 0 < tr["min_us"] <= tr["median_us"] <= tr["max_us"] and tr["min_us"] <= 1.5 * tr["launch_to_signal_seen_us"]
    # a roofline fraction is a fraction: it cannot pass 1 (the 204-B nominal figure may, and is kept aside)
    assert 0.0 < ro["frac"] <= 1.0 and abs(ro["frac"] - ro["achieved"] / 8000.0) < 1e-9
    # which binary ran and whether the counter file describes it
    from q1physrl_amd import _lib, build
    assert d["lib_sha16"] == _lib.lib_sha16() and d["lib_build_id"] == build.sources_sha16() and ro["pmc_stale"] in (True, False)
    assert (ro["traffic"] is None) or not ro["pmc_stale"]
    # the timed region ends in the kernel-written completion signal; what the runtime synchronisation would have
    # added is reported next to it; the device stamps of the timed region fit inside its wall time
    hs = ro["host_split_us"]
    assert hs["completion"].startswith("kernel-written signal") and 0 < hs["device_stamp_us"] <= d["ms_per_step"] * 20 * 1e3
    assert d["ms_per_step_incl_runtime_sync"] >= d["ms_per_step"]
    assert ro["wall_over_event"] >= 1.0
    st = d["per_tick_step"]["roofline"]
    assert st["bound"] == "hbm" and 0.0 < st["frac"] <= 1.0
    sv = d["persistent_server"]
    assert "NOT written to HBM" in sv["workload"] and sv["roofline"]["kernel"].startswith("tick_pair_lds_kernel<true, 1>")
    # the event time of the timed region fits inside its wall time, and 20 ticks of 65 536 envs in well under a millisecond
    assert ro["avg_launch_us"] * 1e-3 <= d["ms_per_step"] * 20 <= 1.0
    assert set(d["steady_state_720_ticks"]) == {"rollout", "step", "server"}


def test_two_ranks_oversubscribed_on_one_device():
    d = _run(["--gpus", "2", "--steps", "20", "--warmup", "5"], extra_env={"Q1_BENCH_OVERSUBSCRIBE": "1"})
    assert d["n_gpus"] == 2 and d["mode"] == "rollout" and d["mode_fallback"] is None and d["env_impl"] == "q1physrl_amd.device.DeviceEnv"
    rows = d["per_rank"]
    assert [r["rank"] for r in rows] == [0, 1]
    assert [r["env_index_base"] for r in rows] == [0, 65536] and all(r["envs"] == 65536 for r in rows)
    assert all(r["mode"] == "rollout" and r["env_impl"] == "q1physrl_amd.device.DeviceEnv" for r in rows)
    assert all(r["host_split_us"] and r["host_split_us"]["enqueue_us"] > 0 for r in rows)
    assert all("placement" in r and r["placement"]["device"] == 0 for r in rows)
    if all(r["placement"].get("pinned") for r in rows) and rows[0]["placement"]["n_cpus"] < os.cpu_count():
        assert rows[0]["placement"]["cpus"] != rows[1]["placement"]["cpus"]          # the two ranks do not share cores
    slowest = max(r["wall_ms"] for r in rows)
    assert abs(d["ms_per_step"] * 20 - slowest) <= 1e-6 * slowest
    # the --steps-independent figures ride along in the multi-rank line too (slowest rank's event time)
    assert set(d["steady_state_720_ticks"]) == {"rollout", "step", "server"} and all("us_per_tick" in v for v in d["steady_state_720_ticks"].values())
    # the CPU baseline is a number in a multi-rank run too: rank 0, after the timed regions, the other rank parked in a barrier -
    # and it still checks the GPU env over the 10 s rollout
    cb = d["cpu_baseline"]
    assert d["config"]["total_envs"] == 131072 and cb["value"] > 1e6 and "rank 0 of 2" in cb["sample"]
    assert cb["parity_vs_gpu_after_719_ticks"]["max_abs_pos_xy_diff"] == 0.0 and d["_line"]["cpu_baseline"]["value"] > 1e6
    tr = d["_line"]["timed_region_us"]
    assert tr["reps"] == 15 and 0 < tr["min_us"] <= tr["median_us"] <= tr["max_us"]


def test_configs2_line_params_yml_with_in_kernel_reset():
    """BASELINE configs[2] as a bench line: params.yml's Config, random starts, finished episodes reset inside the
    rollout kernel - the SPEC instantiation with HAS_RESET = true; 2 000 ticks cross the action tensor's wrap and many episode ends."""
    d = _run(["--gpus", "1", "--steps", "2000", "--warmup", "100", "--envs", "8192", "--config", "params_yml", "--no-secondary"])
    assert d["mode"] == "rollout" and "configs[2]" in d["config"]["workload"] and "reset IN-KERNEL" in d["config"]["workload"]
    ro = d["roofline"]
    assert ro["kernel"].startswith("rollout_kernel<float, true, 2, true, 1, false, 2>") and ro["bound"] == "valu_f64" and 0 < ro["frac"] <= 1.0
    assert d["cpu_baseline"] is None and d["value"] > 1e8
