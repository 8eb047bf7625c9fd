# round 4, last GPU call: the default bench line of the final code + kernel-trace stats + the two HBM traffic passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r04z}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
timeout -k 5 170 python bench.py > $O/bench_line.json 2> $O/bench.err
export DSR_BENCH_NO_POOL=1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg"
timeout -k 5 80 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.log 2>&1
grep '^{"metric"' $O/kt.log > $O/bench_line_under_rocprof.json
timeout -k 5 70 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B --no-profile > $O/fetch.log 2>&1
timeout -k 5 70 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B --no-profile > $O/write.log 2>&1
python tools/profile_summary.py stats $O/kt 20 > $O/kernel_stats.json
python tools/profile_summary.py traffic $O/fetch $O/write 20 $O/bench_line.json > $O/pmc_traffic.json
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/kt/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
find $O -name "*.csv" -size +1M -delete
rm -rf $O/kt $O/fetch $O/write
ls -la $O; head -c 400 $O/bench_line.json; echo; cat $O/pmc_traffic.json | head -c 600
