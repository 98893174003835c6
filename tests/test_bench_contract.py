"""bench.py's output contract.  CPU part: helpers.  GPU part: a short real run must print exactly one JSON line with the
fields the driver and the judge read."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_action_generator_and_traffic_lookup():
    sys.path.insert(0, ROOT)
    import bench
    keys, mouse = bench.make_actions(256, 50, 10.08, seed=3)
    assert keys.shape == (50, 256) and keys.dtype == np.uint8 and keys.max() < 16
    assert mouse.shape == (50, 256) and mouse.dtype == np.float32 and np.abs(mouse).max() <= 10.08
    flips = np.mean((keys[1:] ^ keys[:-1]) != 0)
    assert 0.1 < flips < 0.3                       # ~1 - 0.95^4 of the envs flip at least one key per tick
    pm, _stale = bench.load_pmc("step", 65536)     # profiles/pmc.json: the round's rocprofv3 PMC passes
    if pm is not None:
        t = bench.traffic_per_launch(pm, 65536, 1, resident_state=False)
        assert 9e6 < t < 16e6                      # ~11.7 MB per launch at 65 536 envs (13.4 MB algorithmic; the write-back skips unchanged words)
    pr, _stale = bench.load_pmc("rollout", 65536)
    if pr is not None:                             # register-resident kernel: the 170 B/env state once per launch, the rest per tick
        t720 = bench.traffic_per_launch(pr, 65536, 720, resident_state=True)
        t20 = bench.traffic_per_launch(pr, 65536, 20, resident_state=True)
        assert abs(t720 - (pr["fetch_x2_B"] + pr["write_B"])) < 1.0 and 170 * 65536 < t20 < t720 / 20
        assert 180 < pr["insts_valu"] / pr["waves"] / 720 < 400        # VALU instructions per tick per wave (337 -> 294 in round 3)
    # staleness guard: counters taken from another build of the library are withheld
    ent, stale = bench.load_pmc("rollout", 65536, lib_build_id="0000000000000000")
    if pr is not None:
        assert ent is None and stale is True
    ent, stale = bench.load_pmc("rollout", 12345, lib_build_id="0000000000000000")
    assert ent is None and stale is False              # a size that was never profiled is "absent", not "stale"
    assert bench.B_ALG == 204.0 and bench.EPISODE_TICKS == 720


@pytest.mark.gpu
def test_bench_prints_one_contract_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "288", "--warmup", "72", "--envs", "8192",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, Q1_BENCH_EXTRA=os.path.join(tempfile.gettempdir(), f"q1_bench_extra_contract_{os.getpid()}.json")))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    with open(line["extra"]) as f:          # the side file: the line's fields + every secondary measurement
        d = json.load(f)
    assert abs(d["value"] - line["value"]) <= 1e-6 * d["value"] and set(line["steady_state_us_per_tick"]) == {"rollout", "step", "server"}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 288 and d["warmup"] == 72 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "env-steps/s" and d["value"] > 1e8 and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 8192) / 8192 < 1e-6           # value = envs / time per step
    rf = d["roofline"]
    # default mode = the fused rollout with every tick's outputs in HBM: a register-resident kernel; the roofline object is on the HBM
    # axis (algorithmic bytes over the launch time) and every fraction is a fraction
    assert d["mode"] == "rollout" and d["mode_fallback"] is None and d["env_impl"] == "q1physrl_amd.device.DeviceEnv"
    # (`bound` names the limiter - float64 VALU issue for the register-resident kernels -, `frac` stays on the HBM axis SURVEY 8(d) prescribes,
    #  frac_8d_204B prices the launch at 8(d)'s literal 204 B per env-step: not a fraction, may pass 1)
    assert rf["bound"] == "valu_f64" and rf["frac_axis"] == "hbm" and rf["frac_8d_204B"] > rf["frac"]
    assert "rollout_kernel<float, true, 2, false, 1, false, 2>" in rf["kernel"]            # 288 ticks per launch: the two-ahead form
    assert 0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["peak"] == 8000.0
    assert len(d["lib_sha16"]) == 16 and len(d["lib_build_id"]) == 16
    st = d["per_tick_step"]
    assert st["value"] > 1e8 and st["roofline"]["bound"] == "hbm" and st["roofline"]["unit"] == "GB/s" and 0 < st["roofline"]["frac"] <= 1.0
    assert "tick_pair_lds_kernel" in d["persistent_server"]["roofline"]["kernel"]
    ss = d["steady_state_720_ticks"]
    assert ss["step"]["us_per_tick"] > 0 and ss["server"]["us_per_tick"] > 0 and ss["rollout"]["us_per_tick"] > 0
