# Cache / memory-path PMC passes for the bench workload: one small counter group per pass, each under its own
# timeout (a pass with an unknown counter fails alone), no forked worker pool under the profiler.
# bash tools/profile_cache.sh <tag>   -> gpurun_out/cache_<tag>/pmc_cache.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export DSR_BENCH_NO_POOL=1
TAG=${1:-r02}
O=$GRAFT_REPO_ROOT/gpurun_out/cache_$TAG
rm -rf $O; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-through-shim --no-profile"
i=0
for G in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $G -d $O/g$i -o p --output-format csv -- $B > $O/g$i.log 2>&1 || echo "group $i failed: $G" >> $O/failed.txt
done
python tools/profile_summary.py pmc $O/g1 $O/g2 $O/g3 > $O/pmc_cache.json
find $O -name "*.csv" -delete
rm -rf $O/g1 $O/g2 $O/g3
ls -la $O
