export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof5cm
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --preset 5cm --no-cpu-baseline --no-profile > $O/kt.log 2>&1
python tools/profile_summary.py stats $O/kt > $O/kernel_stats.json
grep '^{"metric"' $O/kt.log | cut -c1-220
rm -rf $O/kt
