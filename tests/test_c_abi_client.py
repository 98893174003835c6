"""The drop-in boundary from plain C: tests/c_abi_client.c includes include/q1env.h, links libq1env.so and nothing of the Python
layer.  CPU: it must compile and link against the header and the library (every symbol it uses resolves) and, without a GPU,
fail loudly with the library's own message (exit status 2) - no CPU fallback.  GPU: it runs its 1 000-env x 300-tick parity
check of q1env_step_host / q1env_get_state_host against the C oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    from q1physrl_amd import build as B
    B.build_lib()
    exe = str(tmp_path / "c_abi_client")
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_client.c"),
           os.path.join(ROOT, "oracle", "q1_oracle.c"), "-o", exe, "-L", os.path.join(ROOT, "q1physrl_amd"), "-lq1env",
           "-Wl,-rpath," + os.path.join(ROOT, "q1physrl_amd"), "-Wl,-rpath,/opt/rocm/lib", "-lm", "-fopenmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_c_client_builds_against_the_header_and_fails_loudly_without_a_gpu(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_client_parity_on_the_gpu(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C_ABI_CLIENT_OK" in r.stdout, r.stdout + r.stderr
