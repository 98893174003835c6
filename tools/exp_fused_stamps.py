"""Diagnostic: where a wave of the learner's fused forward + backward kernel (csrc/q1learner_fused.hpp) spends its time.  Builds a variant of the
library with -DQ1_FZ_STAMPS (every wave stamps a 100 MHz clock at its phase boundaries and leaves the stamps in the dW1 product array, unused in
step mode "fused"), runs q1env_learner_sgd_step at 32 768 samples and prints medians over the waves of each network.

    python tools/exp_fused_stamps.py            (on the GPU box; the product library is not touched)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = "".join(a[len("-DQ1_FZ_EXP="):] for a in sys.argv[1:] if a.startswith("-DQ1_FZ_EXP="))      # timing experiments (csrc/q1learner_fused.hpp Q1_FZ_EXP)
out = os.path.join(ROOT, "q1physrl_amd", "libq1env_fzstamps%s.so" % VARIANT)
os.environ["Q1ENV_LIB_PATH"] = out               # before the package is imported: _lib reads it at import time
from q1physrl_amd import build

if "--build-only" in sys.argv or not os.path.exists(out):
    build.build_lib(force=True, extra_flags=["-DQ1_FZ_STAMPS=1"] + [a for a in sys.argv[1:] if a.startswith("-D")], out=out, tag="_fzstamps" + VARIANT)
    if "--build-only" in sys.argv:
        sys.exit(0)
import torch
from q1physrl_amd import policy as P, ppo
from q1physrl_amd.tensor_env import TensorVectorEnv
from q1physrl_amd.env import Config

mb = 32768
env = TensorVectorEnv(Config(**dict(Config.get_default().__dict__, num_envs=256)), device=0, seed=1)
MODE = "fused_dw1" if "--dw1" in sys.argv else "fused"
env._dev.learner_set_step_mode(MODE)
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    nat.step(full, idx, 0.1, 7500.0, 1.0, 0.01, klc, skip_reduce=True, adam=(3e-5, (0.9, 0.999), 1e-8))
e1.record()
torch.cuda.synchronize()
print("variant", VARIANT or "0", MODE, "step us", e0.elapsed_time(e1) * 10.0)
tiles = mb // 32
per = tiles * 8192                                  # bytes of one network's dW1 product array: the workspace's last two pieces
names = ["forward image staged + barrier", "forward done", "barrier 1 (all waves forward done)", "backward images staged + barrier", "loss gradient done",
         "[x|1], dY transposed + stored", "dZ2 phase done", "pass 0: 64 MFMAs done", "pass 0: epilogue done", "pass 1: 64 MFMAs done", "pass 1: epilogue done",
         "wave end"]
import numpy as np
for label, k in (("policy network", 0), ("value network", 1)):
    if MODE == "fused":
        a = nat.ws[nat.ws.numel() - (2 - k) * per: nat.ws.numel() - (1 - k) * per].view(torch.float32).reshape(tiles, 2048)[:, :14].cpu()
    else:                                           # the dZ1 array of network k (csrc/q1env_learner.hip carve_ws: images, h1, h2, dZ2, dZ1, [x|1], dY, partial sums)
        act = tiles * 16384
        per_net = 152064 + 135168 + 20480 + 4 * act + 2 * tiles * 2048 + 32 * 89 * 1024 * 4
        o = k * per_net + 307712 + 3 * act
        a = nat.ws[o:o + act].view(torch.float32).reshape(tiles, 4096)[:, :14].cpu()
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r6_fused"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "r6_fused", "stamps%s%s_net%d.npy" % (VARIANT, "_dw1" if MODE != "fused" else "", k)), a.numpy())
    print(label, "first wave start - previous launch's last wave end: min %.2f us, median %.2f" % (a[:, 12].min(), a[:, 12].median()))
    st = a[:, 13].numpy().astype(np.int64)
    st = (st - st.min()) % (1 << 24)
    print(label, "wave starts span %.2f us; last wave end - first wave start = %.2f us" % (st.max() * 0.01, (st * 0.01 + a[:, 11].numpy()).max()))
    print(label, "(us since the wave's start: median / min / max over %d waves)" % tiles)
    for j, nm in enumerate(names):
        print(f"  {nm:40s} {a[:, j].median():7.2f} {a[:, j].min():7.2f} {a[:, j].max():7.2f}")
