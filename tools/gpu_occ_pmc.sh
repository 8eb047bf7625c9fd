#!/bin/bash
# FETCH_SIZE of k_raycast with the occupancy map on / off (PMC pass only, no tracing)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export DSR_BENCH_NO_POOL=1   # no forked worker pool under the profiler (it hangs there)
for occ in 1 0; do
  O=/tmp/pmc_occ$occ; rm -rf $O
  DSR_OCC=$occ timeout 150 rocprofv3 --pmc FETCH_SIZE -d $O -o p --output-format csv -- python tools/bench_variants.py "" > /tmp/pmc_occ$occ.log 2>&1
  python - $O $occ <<'PY'
import csv, glob, sys, collections
d, occ = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_raycast" in k or "k_integrate" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print("occ=" + occ, k, {c: round(v / n[(k, c)], 1) for c, v in acc[k].items()}, "launches", max(n[(k, c)] for c in acc[k]))
PY
done
