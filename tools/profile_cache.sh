# Cache / memory-path PMC passes for the bench workload (one small counter group per pass; a pass with an
# unknown counter name fails alone).  bash tools/profile_cache.sh <tag>   -> gpurun_out/cache_<tag>/pmc_cache.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02}
O=$GRAFT_REPO_ROOT/gpurun_out/cache_$TAG
rm -rf $O; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-profile"
rocprofv3 -L > $O/counters_available.txt 2>&1
i=0
for G in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
         "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
         "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_LATENCY_sum TCC_BUSY_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G -d $O/g$i -o p --output-format csv -- $B > $O/g$i.log 2>&1 || echo "group $i failed: $G" >> $O/failed.txt
done
python tools/profile_summary.py pmc $O/g* > $O/pmc_cache.json
find $O -name "*.csv" -delete
rm -rf $O/g*/
ls -la $O
