#!/bin/bash
# Round 6: HBM traffic of the contract line's OWN launch, measured directly: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only)
# of the driver's bench command; per dispatch of rollout_kernel<..., 1> (bench.py derives the line's `traffic` from the 720-tick launches of the profile
# round: per-tick bytes x 20 + the per-launch state).  Units / gfx950 correction as tools/summarize_pmc.py (counter x 1024 B; FETCH_SIZE x 2).
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_driver_pmc
mkdir -p $O
export Q1_TUNABLEOP=0
python -c "import q1physrl_amd._lib as L, q1physrl_amd.build as B; print('build id', B.sources_sha16(), 'lib sha16', L.lib_sha16())" > $O/build_id.txt 2>&1
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o t -- python bench.py $ARGS > $O/line_fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o t -- python bench.py $ARGS > $O/line_write.json 2> $O/write.err
python - "$O" <<'PY'
import csv, glob, json, statistics, sys
O = sys.argv[1]
KERNEL = "rollout_kernel<float, true, 2, false, 1, false, 1>"
out = open(O + "/summary.txt", "w")
def p(*a):
    print(*a); print(*a, file=out)
p("#", open(O + "/build_id.txt").read().strip(), "- tools/r6_driver_pmc.sh, one MI355X lease")
p("# rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary   (one pass per counter)")
vals = {}
for tag, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = sorted(glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True))
    rows = [r for r in csv.DictReader(open(f[0])) if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == ctr] if f else []
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals[tag] = [float(r["Counter_Value"]) * 1024.0 for r in rows]
    p(f"# {ctr}: {len(rows)} dispatches of {KERNEL}; bytes per dispatch (counter x 1024), in dispatch order:")
    p("   " + " ".join(f"{v / 1e6:.2f}" for v in vals[tag]) + "  MB")
w = vals.get("write", []); f = vals.get("fetch", [])
if w and f and len(w) == len(f):
    thr = (min(w) + max(w)) / 2.0
    big = [i for i, v in enumerate(w) if v > thr]            # the 20-tick launches write 4 x the 5-tick warm-up launches' outputs
    f20 = [2.0 * f[i] for i in big]; w20 = [w[i] for i in big]
    tot = [a + b for a, b in zip(f20, w20)]
    alg = (34 * 20 + 170) * 65536
    p(f"# {len(big)} twenty-tick launches: fetch (x 2, gfx950) mean {statistics.mean(f20) / 1e6:.3f} MB  write mean {statistics.mean(w20) / 1e6:.3f} MB  "
      f"total mean {statistics.mean(tot) / 1e6:.3f} MB (min {min(tot) / 1e6:.3f}, max {max(tot) / 1e6:.3f})")
    p(f"# algorithmic bytes of the launch: (34 B x 20 ticks + 170 B) x 65 536 envs = {alg / 1e6:.3f} MB -> measured / algorithmic = {statistics.mean(tot) / alg:.3f}")
    try:
        d = json.loads(open(O + "/line_write.json").read().strip().splitlines()[-1])
        p(f"# the contract line's roofline.traffic (derived from the 720-tick launches of profiles/pmc.json): {d['roofline']['traffic'] / 1e6:.3f} MB, pmc_stale {d['roofline']['pmc_stale']}")
    except Exception as ex:      # noqa: BLE001
        p(f"# (line unreadable: {ex!r})")
PY
