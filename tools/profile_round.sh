# Round profile: kernel-trace stats of the default bench command + separate PMC passes.
# Run on the GPU box from the repo root: bash tools/profile_round.sh <tag>   (writes gpurun_out/prof_<tag>/)
# PMC passes are never combined with tracing (gpurun refuses that combination).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export DSR_BENCH_NO_POOL=1   # no forked worker pool under the profiler (it has hung there: a 30 GPU-minute call in round 2)
TAG=${1:-r02a}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
B="timeout 200 python bench.py --steps 45 --warmup 5 --no-cpu-baseline --no-through-shim --no-profile"
timeout 300 python bench.py > $O/bench_line.json 2> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --no-cpu-baseline --no-through-shim > $O/kt.log 2>&1
grep '^{"metric"' $O/kt.log > $O/bench_line_under_rocprof.json
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B > $O/write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH -d $O/inst -o p --output-format csv -- $B > $O/inst.log 2>&1
timeout 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $O/cyc -o p --output-format csv -- $B > $O/cyc.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d $O/cyc2 -o p --output-format csv -- $B > $O/cyc2.log 2>&1
python tools/profile_summary.py stats $O/kt 45 > $O/kernel_stats.json
python tools/profile_summary.py traffic $O/fetch $O/write 45 $O/bench_line.json > $O/pmc_traffic.json
python tools/profile_summary.py pmc $O/inst $O/cyc $O/cyc2 > $O/pmc_sq.json
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/kt/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
timeout 200 python bench.py --profile-all --no-cpu-baseline --no-through-shim > $O/bench_line_profile_all.json 2>> $O/bench.err
find $O -name "*.csv" -size +1M -delete
rm -rf $O/kt $O/fetch $O/write $O/inst $O/cyc $O/cyc2
ls -la $O
