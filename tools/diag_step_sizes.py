"""Workload for tools/diag_step_sizes.sh: the per-tick step kernel (48 graph-replayed ticks x 2) and the known-bytes copy kernel
(32 launches) at 262 144 / 1 M / 4 M envs in one process, so that one rocprofv3 --pmc pass yields both kernels at all three sizes.
With `--summarize <dir>` it prints the counters per kernel and grid size instead."""
import csv, glob, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            name = "step_kernel" if "step_kernel" in kn else "step2_kernel" if "step2_kernel" in kn else "calib_copy" if "calib_copy" in kn else None
            if name is None:
                continue
            key = (name, int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for key in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        n = key[1] * (2 if key[0] == "step2_kernel" else 1)
        print(f"== {key[0]} envs={n}  (dispatches averaged: {len(next(iter(acc[key].values())))}; span under the profiler {sum(dur[key]) / len(dur[key]) / 1e3:.1f} us)")
        for k in sorted(c):
            print(f"   {k:40s} {c[k]:16.1f}   per env {c[k] / n:10.4f}")
        d = c
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            print(f"   -> L2 hit rate {d['TCC_HIT_sum'] / max(d['TCC_HIT_sum'] + d['TCC_MISS_sum'], 1):.3f}")
        if "TCP_TCC_READ_REQ_LATENCY_sum" in d and d.get("TCP_TCC_READ_REQ_sum"):
            print(f"   -> mean L1->L2 read latency {d['TCP_TCC_READ_REQ_LATENCY_sum'] / d['TCP_TCC_READ_REQ_sum']:.0f} cycles;"
                  f"  write {d.get('TCP_TCC_WRITE_REQ_LATENCY_sum', 0) / max(d.get('TCP_TCC_WRITE_REQ_sum', 1), 1):.0f}")
        if "TCC_EA0_RDREQ_sum" in d and "TCC_EA0_RDREQ_DRAM_sum" in d:
            print(f"   -> fabric reads {d['TCC_EA0_RDREQ_sum']:.0f}, of which DRAM-routed {d['TCC_EA0_RDREQ_DRAM_sum']:.0f}")
        if "TCP_UTCL1_TRANSLATION_MISS_sum" in d:
            print(f"   -> UTCL1 (L1 TLB) miss rate {d['TCP_UTCL1_TRANSLATION_MISS_sum'] / max(d['TCP_UTCL1_TRANSLATION_MISS_sum'] + d.get('TCP_UTCL1_TRANSLATION_HIT_sum', 0), 1):.4f}")
        if "SQ_WAVE_CYCLES" in d:
            print(f"   -> per wave: cycles {4 * d['SQ_WAVE_CYCLES'] / d['SQ_WAVES']:.0f}, waiting (s_waitcnt) {4 * d.get('SQ_WAIT_ANY', 0) / d['SQ_WAVES']:.0f}, "
                  f"issue-stalled {4 * d.get('SQ_WAIT_INST_ANY', 0) / d['SQ_WAVES']:.0f}, VALU busy {4 * d.get('SQ_ACTIVE_INST_VALU', 0) / d['SQ_WAVES']:.0f}")
    sys.exit(0)

import torch
from q1physrl_amd import _lib
from q1physrl_amd.device import DeviceEnv
from q1physrl_amd.env import Config
torch.manual_seed(0)
for n in (262144, 1048576, 4194304):
    cfg = Config(**{**Config.get_default().__dict__, "num_envs": n, "zero_start_prob": 1.0})
    dev = DeviceEnv(cfg, device=0)
    T = 48
    d = torch.device("cuda")
    keys = torch.randint(0, 16, (T, n), dtype=torch.uint8, device=d)
    mouse = torch.rand((T, n), device=d) * 20.0 - 10.0
    obs = torch.empty((n, 6), dtype=torch.float32, device=d); rew = torch.empty((n,), dtype=torch.float32, device=d); done = torch.empty((n,), dtype=torch.uint8, device=d)
    torch.cuda.synchronize()
    for _ in range(2):
        dev.step_many_dev(T, _lib.ACT_PACKED, keys.data_ptr(), mouse.data_ptr(), _lib.OBS_F32, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), 0, False)
    dev.sync()
    dev.calibrate_traffic(32)
    dev.close()
