"""bench.py's multi-GPU launch path on CPU: `python bench.py --gpus 2` must start its own two workers (no torchrun), rendezvous
on 127.0.0.1, time the region on every rank, reduce MAX over ranks and print ONE JSON line on rank 0.  The per-GPU env handle
is replaced by an oracle-backed stand-in INSIDE THIS TEST ONLY (tests/_bench_fake.py via Q1_BENCH_ENV_FACTORY); everything else
is the code the driver runs on an 8-GPU node.  Also: the same two-rank run under torch.distributed.run, and the single-process
path with world size 1."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = "tests._bench_fake:FakeDeviceEnv"


def _run(cmd, extra_env=None, timeout=300):
    env = dict(os.environ, Q1_BENCH_ENV_FACTORY=FAKE, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _one_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), stdout          # rank 0's stdout is the JSON line and nothing else
    return json.loads(lines[0])


def _check_line(d, world, envs, steps, warmup):
    assert d["n_gpus"] == world and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak"
    assert d["unit"] == "env-steps/s" and d["value"] > 0 and d["config"]["total_envs"] == envs * world
    assert f"x{world}" in d["config"]["parallelism"]
    rows = d["per_rank"]
    assert [r["rank"] for r in rows] == list(range(world))
    slowest = max(r["wall_ms"] for r in rows)
    assert abs(d["ms_per_step"] * steps - slowest) <= 1e-6 * slowest            # MAX over ranks is what `value` is built on
    assert abs(d["value"] - envs * world * steps / (slowest * 1e-3)) <= 1e-6 * d["value"]
    assert d["cpu_baseline"] is None or world == 1


def test_self_launch_two_ranks_without_torchrun():
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs", "48"])
    assert r.returncode == 0, r.stderr[-3000:]
    _check_line(_one_json(r.stdout), 2, 48, 20, 5)


def test_two_ranks_under_torch_distributed_run():
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29731", "bench.py", "--gpus", "2", "--steps", "12", "--warmup", "3", "--envs", "32",
              "--no-secondary"])
    assert r.returncode == 0, r.stderr[-3000:]
    _check_line(_one_json(r.stdout), 2, 32, 12, 3)


def test_world_mismatch_is_refused():
    r = _run([sys.executable, "bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1", "--envs", "8"],
             extra_env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29733"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_episode_boundary_and_config3_label_single_rank():
    # 131072 envs would be too slow for the oracle stand-in; the label logic is exercised through bench's own helper instead
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "30", "--warmup", "700", "--envs", "16", "--no-cpu-baseline",
              "--no-secondary"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json(r.stdout)
    _check_line(d, 1, 16, 30, 700)
    assert "configs[1]" in d["config"]["workload"]


def test_failed_rank_fails_the_launch():
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--envs", "8"],
             extra_env={"Q1_BENCH_ENV_FACTORY": "tests._bench_fake:Missing"})
    assert r.returncode != 0


def test_server_mode_failure_is_collective_and_auto_falls_back():
    """The oracle stand-in has no tick server.  Explicit `--mode server` must fail on BOTH ranks at the same point (nobody is left
    waiting in a barrier: the run ends in seconds, non-zero); the default `auto` mode must fall back to per-tick launches on every
    rank and say so in the JSON line."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--envs", "16", "--mode", "server", "--no-secondary"],
             timeout=120)
    assert r.returncode != 0 and "server mode failed in the dry run" in r.stderr
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--envs", "16", "--no-secondary"],
             extra_env={"Q1_BENCH_FORCE_AUTO_SERVER": "1"}, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json(r.stdout)
    _check_line(d, 2, 16, 6, 2)
    assert d["mode"] == "step" and "server mode failed" in d["mode_fallback"] and "step_kernel" in d["roofline"]["kernel"]


def test_self_launch_eight_ranks():
    """The driver's largest invocation shape, `python bench.py --gpus 8`, on CPU: eight self-launched workers rendezvous, time their
    regions, agree collectively and rank 0 prints the one line with eight per-rank rows (BASELINE configs[3] is 8 x 131 072 envs)."""
    r = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "10", "--warmup", "3", "--envs", "16", "--no-secondary"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json(r.stdout)
    _check_line(d, 8, 16, 10, 3)
    assert len(d["per_rank"]) == 8 and d["config"]["total_envs"] == 128
