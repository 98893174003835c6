"""Large-batch experiment for the per-tick step kernel: whole-wave (ballot-gated) against per-lane
conditional write-back, register-allocation targets and block sizes at 262 144 / 1 M / 4 M envs, next to the known-bytes copy kernel
in the same process.  (Round 3's two-envs-per-lane variants were measured, dropped and removed.)

    python tools/exp_step_large.py build      # (CPU) build the variant libraries into gpurun_scratch/
    python tools/exp_step_large.py run        # (GPU) time every variant, print a table
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"base": [], "wholewave": ["-DQ1_DELTA_PER_LANE=0"], "w6": ["-DQ1_STEP_MINWAVES=6"]}
RUNS = [("base", {}), ("wholewave", {}), ("base", {}), ("wholewave", {}), ("w6", {}), ("base", {"Q1ENV_BLOCK": "256"})]
# (round 4 also ran through this tool, and removed: a software-pipelined streaming form of the kernel - profiles/r4_step_large.txt - and a
# skew between the SoA state arrays - profiles/r4_exp_skew.txt)

CODE = r'''
import sys, json, torch
sys.path.insert(0, %r)
from q1physrl_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from q1physrl_amd.device import DeviceEnv
from q1physrl_amd.env import Config
out = {}
torch.manual_seed(0)        # every variant sees the same actions: equal state hashes = equal results
for n in (262144, 1048576, 4194304):
    cfg = Config(**{**Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    dev = DeviceEnv(cfg, device=0)
    T = 48
    d = torch.device("cuda")
    keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device=d)
    mouse = torch.rand((T, n), device=d) * 20.0 - 10.0
    obs = torch.empty((n, 6), dtype=torch.float32, device=d); rew = torch.empty((n,), dtype=torch.float32, device=d); done = torch.empty((n,), dtype=torch.uint8, device=d)
    torch.cuda.synchronize()
    def go():
        dev.step_many_dev(T, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), 0, True)
    go(); dev.sync()
    best = 1e9
    for _ in range(3):
        dev.timer_start()
        for _ in range(4): go()
        best = min(best, dev.timer_stop() * 1e3 / (4 * T))
    dev.calibrate_traffic(4)
    dev.timer_start(); dev.calibrate_traffic(32); cp = dev.timer_stop() * 1e3 / 32
    st = dev.get_state()
    import hashlib, numpy as np
    h = hashlib.sha256(b"".join(np.ascontiguousarray(st[k]).tobytes() for k in sorted(st))).hexdigest()[:12]
    out[n] = {"step_us": best, "copy_us": cp, "state_hash": h}
    dev.close()
print(json.dumps(out))
''' % ROOT

if sys.argv[1] == "build":
    from q1physrl_amd import build
    os.makedirs(os.path.join(ROOT, "gpurun_scratch"), exist_ok=True)
    for tag, flags in VARIANTS.items():
        print(build.build_lib(force=True, extra_flags=flags, out=os.path.join(ROOT, "gpurun_scratch", f"libq1env_{tag}.so"), tag="_" + tag))
else:
    rows = []
    for tag, env in RUNS:
        so = os.path.join(ROOT, "gpurun_scratch", f"libq1env_{tag}.so")
        r = subprocess.run([sys.executable, "-c", CODE, so], capture_output=True, text=True, env=dict(os.environ, **env))
        if r.returncode != 0:
            print(tag, env, "FAILED", r.stderr[-500:]); continue
        res = json.loads(r.stdout.strip().splitlines()[-1])
        rows.append((tag, env, res))
        print(f"{tag:5s} {str(env):70s} " + "  ".join(f"{int(n)//1024}k: step {v['step_us']:7.2f} us copy {v['copy_us']:7.2f} us ({175.0*int(n)/v['step_us']/1e6:5.2f} / {170.0*int(n)/v['copy_us']/1e6:5.2f} TB/s) {v['state_hash']}" for n, v in res.items()), flush=True)
