# round 4, last GPU call: the whole GPU suite on the final code, smoke, the default bench line, kernel-trace stats and the two
# HBM traffic passes of the same command (each step under its own timeout; the summaries are written as soon as their inputs exist)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r04z}
G=$GRAFT_REPO_ROOT/gpurun_out
O=$G/prof_$T
rm -rf $O; mkdir -p $O
timeout -k 5 560 python -m pytest tests -m gpu -q --timeout 240 > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 4 $G/${T}_gpu_suite.log
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > $G/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $G/${T}_smoke.log; tail -n 1 $G/${T}_smoke.log
timeout -k 5 200 python bench.py > $O/bench_line.json 2> $O/bench.err
head -c 300 $O/bench_line.json; echo
export DSR_BENCH_NO_POOL=1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg"
timeout -k 5 80 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.log 2>&1
grep '^{"metric"' $O/kt.log > $O/bench_line_under_rocprof.json
python tools/profile_summary.py stats $O/kt 20 > $O/kernel_stats.json
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/kt/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/kt
timeout -k 5 70 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B --no-profile > $O/fetch.log 2>&1
timeout -k 5 70 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B --no-profile > $O/write.log 2>&1
python tools/profile_summary.py traffic $O/fetch $O/write 20 $O/bench_line.json > $O/pmc_traffic.json
find $O -name "*.csv" -size +1M -delete
rm -rf $O/fetch $O/write
ls -la $O; head -c 600 $O/pmc_traffic.json
