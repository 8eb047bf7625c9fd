# Lean PMC set for the dominant kernels: FETCH_SIZE, WRITE_SIZE and the SQ instruction counters, one pass each
# (never combined with tracing), every pass under its own timeout and retried once (rocprofv3 occasionally
# segfaults at start-up on this pool).  bash tools/profile_pmc_lean.sh <tag>  -> gpurun_out/pmc_<tag>/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export DSR_BENCH_NO_POOL=1   # no forked worker pool under the profiler (it has hung there)
TAG=${1:-r02}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-through-shim --no-profile"
pass() {  # name, counters...
  local name=$1; shift
  for attempt in 1; do
    rm -rf $O/$name
    timeout 110 rocprofv3 --pmc "$@" -d $O/$name -o p --output-format csv -- $B > $O/$name.log 2>&1 && break
    echo "pass $name attempt $attempt failed" >> $O/failed.txt
  done
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
if [ "$2" = "full" ]; then
pass inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH
pass cyc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
python tools/profile_summary.py pmc $O/inst $O/cyc > $O/pmc_sq.json
fi
timeout 100 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-through-shim > $O/bench_line.json 2> $O/bench.err
python tools/profile_summary.py traffic $O/fetch $O/write 10 $O/bench_line.json > $O/pmc_traffic.json
find $O -name "*.csv" -delete
rm -rf $O/fetch $O/write $O/inst $O/cyc
ls -la $O
