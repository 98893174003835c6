"""Diagnostic: where a wave of the learner's backward kernel spends its time.  Builds a variant of the library with -DQ1_BWD_STAMPS
(wave 0 of every workgroup stamps a 100 MHz clock after the weight staging, and - for its first tile - before the dZ2 phase, before
and after the 128-MFMA loop, at the tile's end; and at the kernel's end), runs q1env_learner_sgd_step and prints the medians.

    python tools/exp_bwd_stamps.py            (on the GPU box; the product library is not touched)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "q1physrl_amd", "libq1env_bwdstamps.so")
os.environ["Q1ENV_LIB_PATH"] = out               # before the package is imported: _lib reads it at import time
from q1physrl_amd import build

if "--build-only" in sys.argv or not os.path.exists(out):
    build.build_lib(force=True, extra_flags=["-DQ1_BWD_STAMPS=1"], out=out, tag="_bwdstamps")
    if "--build-only" in sys.argv:
        sys.exit(0)
import torch
from q1physrl_amd import policy as P, ppo
from q1physrl_amd.tensor_env import TensorVectorEnv
from q1physrl_amd.env import Config

mb = 32768
env = TensorVectorEnv(Config(**dict(Config.get_default().__dict__, num_envs=256)), device=0, seed=1)
torch.manual_seed(0)
pol = P.Q1Policy().cuda()
total = 4 * mb
g = torch.Generator(device="cuda").manual_seed(2)
obs = torch.randn((total, 6), device="cuda", generator=g)
idx = torch.randperm(total, device="cuda", generator=g)[:mb].contiguous()
nat = ppo.NativeStep(pol, env, mb, splits=32)
full = {"obs": obs, "old_logits": torch.randn((total, 10), device="cuda", generator=g).contiguous(),
        "keys_packed": torch.randint(0, 16, (total,), device="cuda", dtype=torch.uint8),
        "mouse": (torch.rand((total, 1), device="cuda", generator=g) * 20 - 10), "logp": -torch.rand((total,), device="cuda", generator=g) * 5,
        "adv": torch.randn((total,), device="cuda", generator=g), "value": torch.randn((total,), device="cuda", generator=g) * 50,
        "vtarg": torch.randn((total,), device="cuda", generator=g) * 50}
klc = torch.full((1,), 0.2, device="cuda")
for _ in range(5):
    nat.step(full, idx, 0.1, 7500.0, 1.0, 0.01, klc, skip_reduce=True, adam=(3e-5, (0.9, 0.999), 1e-8))
torch.cuda.synchronize()
flat = nat.ws[-2048 * 20:].view(torch.float32).cpu()
st = flat[5120:5120 + 256 * 16].reshape(256, 16)
for label, sel in (("policy network's workgroups", slice(0, 128)), ("value network's workgroups", slice(128, 256))):
    g = st[sel]
    print(label)
    names = ["after staging + barrier", "tile 0: before the dZ2 phase", "tile 0: before the MFMA loop", "tile 0: after the MFMA loop", "tile 0: end"] + \
            [f"  dZ2 phase, after row tile {t}" for t in range(8)] + ["", "", "kernel end (wave 0)"]
    order = [0, 1] + list(range(5, 13)) + [2, 3, 4, 15]
    for k in order:
        print(f"  {names[k]:34s} median {g[:, k].median():7.2f} us   min {g[:, k].min():7.2f}   max {g[:, k].max():7.2f}")
