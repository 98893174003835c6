#!/usr/bin/env python3
"""Summarise tools/profile_round.sh (round 3: profile_round3.sh): per mode / batch size the primary kernel's rocprofv3 statistics and hardware counters per
launch, as text (stdout) and as <out>/pmc.json - the file bench.py reads as profiles/pmc.json for `roofline.traffic` (HBM bytes) and
the VALU-issue roofline of the register-resident kernels.

    per entry "<mode>_<envs>":
      kernel, ticks_per_launch, launches, waves, avg_ns / median_ns / min_ns (kernel-trace span under the profiler), event_us_unprofiled
      fetch_raw_B, fetch_x2_B (gfx950: FETCH_SIZE counts half of a wide coalesced stream - MI355X_MICROARCH.md HBM; calibrated
      below on calib_copy_kernel's known 85 + 85 B per env), write_B
      insts_valu, insts_salu, valu_busy_cycles = 4 x SQ_ACTIVE_INST_VALU (quad-cycles -> cycles, summed over waves = over SIMDs with
      one wave each), wave_cycles = 4 x SQ_WAVE_CYCLES, grbm_gui_active (max over XCDs ~ kernel cycles), breakdown by type
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
PRIMARY = {"rollout": "rollout_kernel", "rollout_params": "rollout_kernel", "step": "step_kernel", "server": "tick_pair_lds_kernel"}
TAG = os.path.basename(out.rstrip("/")).replace("prof_", "")
build_ids = set()


def first(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


def short(name):
    return name.replace("void ", "").split("(")[0]


def counters(path, want):
    """{counter: (mean per dispatch over the dispatches of the kernel matching `want` with the modal grid size, n)}"""
    acc = defaultdict(list)
    if not path:
        return {}, None
    name = None
    for r in csv.DictReader(open(path)):
        if want in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = short(r["Kernel_Name"])
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}, name


res = {}
try:        # a re-run on a copy whose large per-launch CSVs were deleted (profile_round3.sh does that on the box) keeps those entries
    previous = json.load(open(os.path.join(out, "pmc.json")))
except Exception:   # noqa: BLE001
    previous = {}
for d in sorted(glob.glob(os.path.join(out, "*_*"))):
    if not os.path.isdir(d) or os.path.basename(d) == "calib":
        continue
    key = os.path.basename(d)
    mode, n = key.rsplit("_", 1)
    n = int(n)
    want = PRIMARY.get(mode)
    if not want:
        continue
    bmode = "rollout --config params_yml" if mode == "rollout_params" else mode
    e = {"envs": n, "source": f"profiles/{TAG}_summary.txt (tools/profile_round.sh: rocprofv3 --kernel-trace --stats and separate --pmc passes of `bench.py --mode {bmode} --envs {n} --steps 1440 --warmup 720`)"}
    kt = first(f"{key}/trace/**/*kernel_trace.csv")
    if kt:
        spans = []
        info = {}
        for r in csv.DictReader(open(kt)):
            if want in r["Kernel_Name"]:
                spans.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                info = {k: r[k] for k in ("Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size") if k in r}
                e["kernel"] = short(r["Kernel_Name"])
        if spans:
            s2 = sorted(spans)
            e.update({"launches_traced": len(spans), "avg_ns": sum(spans) / len(spans), "median_ns": s2[len(s2) // 2], "min_ns": s2[0], "dispatch": info})
    e["ticks_per_launch"] = 1 if mode == "step" else 720
    try:
        b = json.load(open(os.path.join(out, f"{key}.bench_unprofiled.json")))
        ro = b["roofline"]          # (round 5: the stdout line is the compact one - us_per_tick = avg_launch_us / ticks_per_launch)
        e["event_us_per_tick_unprofiled"] = ro.get("us_per_tick") or ro["avg_launch_us"] / ro["ticks_per_launch"]
        e["value_unprofiled"] = b["value"]
        if b.get("lib_build_id"):                      # which build of the kernels these passes profiled (bench.py's staleness guard)
            e["build_id"] = b["lib_build_id"]
            e["lib_sha16"] = b.get("lib_sha16")
            build_ids.add(b["lib_build_id"])
    except Exception:   # noqa: BLE001
        pass
    f, _ = counters(first(f"{key}/fetch/**/*counter_collection.csv"), want)
    w, _ = counters(first(f"{key}/write/**/*counter_collection.csv"), want)
    if "FETCH_SIZE" in f:
        e["fetch_raw_B"] = f["FETCH_SIZE"][0] * 1024
        e["fetch_x2_B"] = 2 * e["fetch_raw_B"]
    if "WRITE_SIZE" in w:
        e["write_B"] = w["WRITE_SIZE"][0] * 1024
    sq = {}
    for p in ("sq1", "sq2", "sq3", "grbm"):
        c, _ = counters(first(f"{key}/{p}/**/*counter_collection.csv"), want)
        sq.update({k: v[0] for k, v in c.items()})
    if sq:
        if "SQ_WAVES" in sq:
            e["waves"] = sq["SQ_WAVES"]
        if "SQ_INSTS_VALU" in sq:
            e["insts_valu"] = sq["SQ_INSTS_VALU"]
        if "SQ_INSTS_SALU" in sq:
            e["insts_salu"] = sq["SQ_INSTS_SALU"]
        other = [sq[k] for k in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM") if k in sq]
        if other:
            e["insts_mem"] = sum(other)                                  # vector-memory, LDS and scalar-memory instructions
        if "SQ_ACTIVE_INST_VALU" in sq:
            e["valu_busy_cycles"] = 4 * sq["SQ_ACTIVE_INST_VALU"]
        if "SQ_WAVE_CYCLES" in sq:
            e["wave_cycles"] = 4 * sq["SQ_WAVE_CYCLES"]
        if "GRBM_GUI_ACTIVE" in sq:
            e["grbm_gui_active_sum"] = sq["GRBM_GUI_ACTIVE"]
        e["counters_raw"] = sq
    if "kernel" not in e and "fetch_x2_B" not in e and key in previous:
        e = previous[key]
    res[key] = e

print(f"== {out}: primary kernel per bench.py mode / batch size (per launch; PMC passes are separate runs of the same command)")
for key, e in res.items():
    if key.startswith("_"):
        continue
    n, T = e["envs"], e["ticks_per_launch"]
    print(f"-- {key}: {e.get('kernel', '?')}")
    if "avg_ns" in e:
        print(f"   kernel-trace: launches={e['launches_traced']} avg={e['avg_ns']:.1f} ns median={e['median_ns']} ns min={e['min_ns']} ns  -> {e['avg_ns'] / 1e3 / T:.3f} us per tick under the profiler"
              f"  (un-profiled HIP-event period {e.get('event_us_per_tick_unprofiled', float('nan')):.3f} us per tick)   {e.get('dispatch')}")
    if "fetch_x2_B" in e or "write_B" in e:
        fx, wr = e.get("fetch_x2_B", float("nan")), e.get("write_B", float("nan"))
        print(f"   HBM bytes per launch: fetch_raw={e.get('fetch_raw_B', float('nan')) / 1e6:.3f} MB  fetch_x2={fx / 1e6:.3f} MB  write={wr / 1e6:.3f} MB"
              f"   per env-step: fetch_x2={fx / n / T:.2f} B  write={wr / n / T:.2f} B  total={(fx + wr) / n / T:.2f} B")
        if "avg_ns" in e:
            print(f"   (fetch_x2 + write) / kernel-trace avg = {(fx + wr) / e['avg_ns']:.1f} GB/s = {(fx + wr) / e['avg_ns'] / 8000:.3f} of 8 TB/s"
                  + (f";  / un-profiled period = {(fx + wr) / (e['event_us_per_tick_unprofiled'] * T * 1e3):.1f} GB/s = {(fx + wr) / (e['event_us_per_tick_unprofiled'] * T * 1e3) / 8000:.3f}" if "event_us_per_tick_unprofiled" in e else ""))
    if "insts_valu" in e and "waves" in e:
        wv = e["waves"]
        print(f"   waves={wv:.0f}  VALU instructions per tick per wave={e['insts_valu'] / wv / T:.1f}  SALU={e.get('insts_salu', 0) / wv / T:.1f}"
              f"  VMEM + LDS + SMEM={e.get('insts_mem', 0) / wv / T:.1f}")
        if "valu_busy_cycles" in e and "wave_cycles" in e:
            print(f"   VALU-busy cycles per tick per wave={e['valu_busy_cycles'] / wv / T:.0f} of {e['wave_cycles'] / wv / T:.0f} wave cycles = {e['valu_busy_cycles'] / e['wave_cycles']:.3f}"
                  f"  (cycles per VALU instruction: busy {e['valu_busy_cycles'] / e['insts_valu']:.2f}, elapsed {e['wave_cycles'] / e['insts_valu']:.2f})")
        raw = e.get("counters_raw", {})
        parts = {k.replace("SQ_INSTS_VALU_", ""): v / wv / T for k, v in raw.items() if k.startswith("SQ_INSTS_VALU_")}
        if parts:
            print("   per tick per wave by type: " + "  ".join(f"{k}={v:.1f}" for k, v in sorted(parts.items())) + f"  other={e['insts_valu'] / wv / T - sum(parts.values()):.1f}")
        if "grbm_gui_active_sum" in e and "avg_ns" in e:
            cyc = e["grbm_gui_active_sum"] / 8.0
            print(f"   GRBM_GUI_ACTIVE (sum over 8 XCDs / 8) = {cyc:.0f} cycles per launch -> {cyc / e['avg_ns']:.2f} GHz under the profiler;"
                  f"  VALUBusy = valu_busy_cycles / (1024 SIMDs x those cycles) = {e.get('valu_busy_cycles', 0) / 1024 / cyc:.3f}")

# calibration
cf, _ = counters(first("calib/fetch/**/*counter_collection.csv"), "calib_copy_kernel")
cw, _ = counters(first("calib/write/**/*counter_collection.csv"), "calib_copy_kernel")
cal = {}
for tag, pth, ctr in (("fetch", "calib/fetch/**/*counter_collection.csv", "FETCH_SIZE"), ("write", "calib/write/**/*counter_collection.csv", "WRITE_SIZE")):
    p = first(pth)
    if not p:
        continue
    by = defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "calib_copy_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            by[int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))].append(float(r["Counter_Value"]) * 1024)
    for g, v in by.items():
        cal.setdefault(g, {})[tag] = sum(v) / len(v)
if cal:
    print("== calibration: calib_copy_kernel reads 85 B and writes 85 B per env exactly")
    for g, v in sorted(cal.items()):
        print(f"   grid={g}: fetch_raw={v.get('fetch', float('nan')) / g:.2f} B/env  fetch_x2={2 * v.get('fetch', float('nan')) / g:.2f} B/env  write={v.get('write', float('nan')) / g:.2f} B/env")
    res["_calibration"] = {str(g): {"fetch_x2_B_per_env": 2 * v.get("fetch", float("nan")) / g, "write_B_per_env": v.get("write", float("nan")) / g} for g, v in cal.items()}
for e in res.values():
    e.pop("counters_raw", None) if isinstance(e, dict) and False else None
if len(build_ids) == 1:
    res["_build_id"] = build_ids.pop()
elif build_ids:
    print("WARNING: the passes profiled more than one build of the library:", sorted(build_ids))
json.dump(res, open(os.path.join(out, "pmc.json"), "w"), indent=1)
