#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output into small JSON files for profiles/.

  profile_summary.py stats <dir> [n]        -> per-kernel count / total / average duration (+ avg of the last n launches)
  profile_summary.py pmc <dir> [<dir> ...]  -> per-kernel mean of every collected counter
  profile_summary.py traffic <fetch_dir> <write_dir> <n>  -> HBM bytes per launch (last n launches)
  profile_summary.py timeline <dir> <anchor-kernel> [k]   -> the kernels between the k-th last and the (k-1)-th last launch of the
                                                             anchor: start / end in us since the anchor's start, queue id

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


def stats(d, last_n=0):
    """per-kernel calls / total / average from *kernel_stats.csv; with last_n > 0 and a
    *kernel_trace.csv next to it, also the average over the LAST last_n launches of every kernel
    (the timed steps of bench.py), which is what bench.py's HIP events measure."""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            o = out.setdefault(k, {"calls": 0, "total_ns": 0})
            o["calls"] += int(r["Calls"]); o["total_ns"] += int(float(r["TotalDurationNs"]))
    for o in out.values():
        o["avg_us"] = round(o["total_ns"] / max(1, o["calls"]) / 1e3, 2)
    if last_n > 0:
        rows = defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                rows[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in rows.items():
            v.sort()
            tail = [x[1] for x in v[-last_n:]]
            if k in out and len(v) >= last_n:
                out[k]["avg_us_last_%d" % last_n] = round(sum(tail) / len(tail) / 1e3, 2)
    return dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ns"]))


def timeline(d, anchor, back=3):
    """One steady-state step as the GPU ran it: every kernel dispatch from the `back`-th last launch of `anchor` up to the next
    one — start and end (us, relative), duration, queue.  Gaps between rows are what the launches of a dependent chain cost."""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if r[2] == anchor]
    if len(idx) < back + 1:
        return {"error": "anchor launched %d times" % len(idx)}
    a, b = idx[-back - 1], idx[-back]
    t0 = rows[a][0]
    return {"anchor": anchor, "step_us": round((rows[b][0] - t0) / 1e3, 2),
            "kernels": [{"name": r[2], "queue": r[3], "start_us": round((r[0] - t0) / 1e3, 2), "end_us": round((r[1] - t0) / 1e3, 2),
                         "us": round((r[1] - r[0]) / 1e3, 2)} for r in rows[a:b]]}


def pmc(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: {c: {"mean": v[0] / v[1], "launches": v[1]} for c, v in cs.items()} for k, cs in acc.items()}


def traffic(fetch_dir, write_dir, last_n, bench_line=None):
    """HBM bytes per launch from separate FETCH_SIZE / WRITE_SIZE passes, averaged over the LAST
    last_n launches of every kernel (the timed steps of bench.py; earlier launches are warm-up).
    Units per MI355X_MICROARCH.md / tools/pmc_calibrate.py: both counters are in KB; FETCH_SIZE
    reports half of the bytes actually read (a 1 GiB device copy reads 524300 "KB")."""
    def per_kernel(d, counter):
        rows = defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    rows[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        out = {}
        for k, v in rows.items():
            v.sort()
            tail = [x[1] for x in v[-last_n:]]
            out[k] = (sum(tail) / len(tail), len(tail))
        return out
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, (0.0, 0))[0] * 1024.0 * 2.0
        w = wr.get(k, (0.0, 0))[0] * 1024.0
        kernels[k] = {"launches": max(fe.get(k, (0, 0))[1], wr.get(k, (0, 0))[1]),
                      "fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes": f + w}
    # k_integrate's traffic is proportional to the visible blocks it walks: normalise with the visible
    # blocks per launch of the SAME frames (bench line of the same command), so that bench.py can
    # report traffic for any --steps/--warmup of the workload
    if bench_line and "k_integrate" in kernels:
        try:
            line = [l for l in open(bench_line) if l.startswith('{"metric"')][-1]
            v = json.loads(line)["roofline"]["visible_blocks_per_launch"]
            kernels["k_integrate"]["visible_blocks_per_launch"] = v
            kernels["k_integrate"]["hbm_bytes_per_visible_block"] = kernels["k_integrate"]["hbm_bytes"] / v
        except Exception as ex:  # keep the per-launch figures
            kernels["k_integrate"]["per_block_error"] = str(ex)
    return {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python bench.py --steps 45 "
                    "--warmup 5 --no-cpu-baseline --no-profile`; KB per launch averaged over the last %d launches "
                    "of each kernel (the timed steps); FETCH_SIZE x2 per MI355X_MICROARCH.md (calibrated with "
                    "tools/pmc_calibrate.py: a 1 GiB device copy reads 524300 KB, writes 1048576 KB)" % last_n,
            "kernels": kernels}


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "timeline":
        res = timeline(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 3)
    elif mode == "traffic":
        res = traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else None)
    else:
        res = stats(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0) if mode == "stats" else pmc(sys.argv[2:])
    json.dump(res, sys.stdout, indent=1)
    print()
