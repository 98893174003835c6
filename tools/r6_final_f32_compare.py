#!/usr/bin/env python3
"""Compare the logged iterations of a prefix run (tools/r6_final_f32_prefix.sh, final build) with the full run of the same seed under profiles/
(earlier build): every field but wall clocks / checkpoint paths, NaN == NaN.  The prefix run's LAST row carries the end-of-run evaluation, which the full run
makes only on its evaluation schedule: the evaluation fields of that row are compared only if both have them."""
import json, math, sys
SKIP = {"sample_s", "iter_s", "wall_s", "checkpoint", "checkpoint_best"}


def compare(prefix_path, full_path):
    a = json.load(open(prefix_path))["log"]
    b = {r["iter"]: r for r in json.load(open(full_path))["log"]}
    n = bad = 0
    for ra in a:
        rb = b.get(ra["iter"])
        if rb is None:
            continue
        for k, x in ra.items():
            if k in SKIP or k not in rb:
                continue
            y = rb[k]
            if isinstance(x, float) and isinstance(y, float) and math.isnan(x) and math.isnan(y):
                continue
            n += 1
            if x != y:
                bad += 1
                if bad <= 5:
                    print(f"  iteration {ra['iter']} {k}: {x!r} != {y!r}")
    return len(a), n, bad


if __name__ == "__main__":
    rc = 0
    for p, f in zip(sys.argv[1::2], sys.argv[2::2]):
        rows, n, bad = compare(p, f)
        print(f"{p} vs {f}: {rows} logged iterations, {n} values compared, {bad} differ")
        rc |= bad != 0
    sys.exit(rc)
