"""bench.py's output contract.  CPU part: helpers.  GPU part: a short real run must print exactly one JSON line with the
fields the driver and the judge read."""
import json
import os
import subprocess
import sys

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
    t = bench.load_profiled_traffic("step", 65536)
    assert t is None or 9e6 < t < 16e6             # profiles/traffic.json: ~11.7 MB per launch at 65 536 envs (13.4 MB algorithmic;
                                                   # the write-back skips unchanged state words)
    assert bench.B_ALG == 204.0 and bench.EPISODE_TICKS == 720


@pytest.mark.gpu
def test_bench_prints_one_contract_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "288", "--warmup", "72", "--envs", "8192",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 288 and d["warmup"] == 72 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "env-steps/s" and d["value"] > 1e8 and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 8192) / 8192 < 1e-6           # value = envs / time per step
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert d["fused_rollout"]["value"] > d["value"]
    # the default primary mode is the resident tick server (falling back to per-tick launches, loudly, if it cannot run); the
    # per-tick kernels are then reported next to it, and the steady-state block carries both
    assert d["mode"] in ("server", "step") and (d["mode"] == "server" or d["mode_fallback"])
    if d["mode"] == "server":
        assert d["mode_fallback"] is None and d["per_tick_step"]["value"] > 1e8 and "tick_pair_lds_kernel" in rf["kernel"]
    ss = d["steady_state_720_ticks"]
    assert ss["step"]["us_per_tick"] > 0 and ss["server"]["us_per_tick"] > 0
