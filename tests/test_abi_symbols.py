"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/q1env.h declares; with no device the entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from q1physrl_amd import build, _lib
    build.build_lib()
    return _lib.load()


def header_functions():
    src = open(os.path.join(ROOT, "include", "q1env.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(q1(?:env|phys)_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    from q1physrl_amd import _lib
    names = header_functions()
    assert len(names) >= 20
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    for n in names:
        assert getattr(lib, n) is not None


def test_abi_version(lib):
    src = open(os.path.join(ROOT, "include", "q1env.h")).read()
    assert int(re.search(r"#define Q1ENV_ABI_VERSION (\d+)", src).group(1)) == lib.q1env_abi_version()


def test_struct_layout_matches_header():
    from q1physrl_amd import _lib
    assert ctypes.sizeof(_lib.Q1Config) == 8 * 4 + 10 * 8 + 8 + 2 * 4      # + legacy_promotion, reserved0 (ABI v2)
    assert _lib.Q1Config.legacy_promotion.offset == 120 and _lib.Q1Config.env_index_base.offset == 112
    assert ctypes.sizeof(_lib.Q1State) == 10 * ctypes.sizeof(ctypes.c_void_p)


def test_code_object_is_gfx950_only():
    from q1physrl_amd import build
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    import subprocess
    out = subprocess.run([objdump, "--offloading", build.OUT], capture_output=True, text=True).stdout
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"}, archs


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from q1physrl_amd import env as E, phys as P, _lib
    with pytest.raises(_lib.Q1EnvError, match="no HIP device|no ROCm"):
        E.VectorPhysEnv(dict(E.Config.get_default().__dict__, num_envs=4))
    with pytest.raises(_lib.Q1EnvError):
        P.apply(P.Inputs(*[np.zeros(2)] * 7), P.PlayerState(np.zeros(2), np.zeros((2, 3), np.float32), np.zeros(2, bool), np.ones(2, bool)))


def test_learner_workspace_sizes_are_host_arithmetic(lib):
    """q1env_learner_workspace_bytes / _adam_state_bytes need no device: pure layout arithmetic (include/q1env.h, ABI v3).  The workspace
    holds, per network, three float16 weight images, four float16 activation arrays of 16 KiB per 32-sample tile (h1, h2 in the forward
    kernel's layout, dZ1, dZ2 in the weight-gradient kernel's; round 4 dropped the second copies of h1 / h2), two small operand
    arrays, 8 KiB of dW1 / db1 products per tile and 8 + 2 (policy) / 2 (value network) KiB of dW3 products (round 6: what the fused forward +
    backward kernel leaves instead of dZ1 and tanh(H2)) and `splits` partial-sum slabs; it must grow linearly in the tile count and in the splits, and refuse nonsense."""
    ws = lib.q1env_learner_workspace_bytes
    b1, b2 = ws(32768, 10, 32), ws(65536, 10, 32)
    tiles = 32768 // 32
    per_tile = 2 * (4 * 16384 + 2 * 2048 + 8192 + 2048) + 8192      # both networks
    assert b2 - b1 == tiles * per_tile + 2 * (32768 * 10 * 4) + 2 * (32768 * 4)
    assert ws(32768, 10, 64) - b1 == 2 * 32 * 89 * 1024 * 4         # 89 products of 32x32 float32 per split and network
    assert ws(1000, 19, 4) > 0 and ws(1000, 19, 4) % 256 == 0      # ragged minibatch: whole tiles, 256-byte aligned pieces
    assert ws(0, 10, 32) == 0 and ws(4096, 33, 32) == 0 and ws(4096, 10, 0) == 0 and ws(4096, 10, 513) == 0
    ad = lib.q1env_learner_adam_state_bytes
    n_pi, n_vf = 65536 + 256 + 1536 + 256 + 10 * 257, 65536 + 256 + 1536 + 256 + 257
    assert ad(10) >= 256 + 2 * 4 * (n_pi + n_vf) and ad(10) < 256 + 2 * 4 * (n_pi + n_vf) + 1024 and ad(0) == 0 and ad(33) == 0
