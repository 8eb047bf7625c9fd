#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output into small JSON files for profiles/.

  profile_summary.py stats <dir>            -> per-kernel count / total / average duration
  profile_summary.py pmc <dir> [<dir> ...]  -> per-kernel mean of every collected counter

Counters are reported raw; the FETCH_SIZE / WRITE_SIZE -> bytes conversion (KB units, FETCH_SIZE
x2 for wide reads on gfx950, see tools/pmc_calibrate.py and MI355X_MICROARCH.md) is applied by the
caller that writes profiles/*_pmc_traffic.json."""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for pre in ("void ", "dsr::"):
        name = name.replace(pre, "")
    return name.split("<")[0].strip()


def stats(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            o = out.setdefault(k, {"calls": 0, "total_ns": 0})
            o["calls"] += int(r["Calls"]); o["total_ns"] += int(float(r["TotalDurationNs"]))
    for o in out.values():
        o["avg_us"] = round(o["total_ns"] / max(1, o["calls"]) / 1e3, 2)
    return dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ns"]))


def pmc(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: {c: {"mean": v[0] / v[1], "launches": v[1]} for c, v in cs.items()} for k, cs in acc.items()}


if __name__ == "__main__":
    mode = sys.argv[1]
    res = stats(sys.argv[2]) if mode == "stats" else pmc(sys.argv[2:])
    json.dump(res, sys.stdout, indent=1)
    print()
