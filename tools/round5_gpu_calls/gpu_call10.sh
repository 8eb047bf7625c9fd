# round 5, call 10: previews — the copy-engine path only for a map-sized volume that has the GPU to itself (call 9: the map next to instance drivers must store directly too)
# memory (no conversion scratch, no copy commands) — parity of everything that goes through host buffers, then the C++ host
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05j}
G=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $G
timeout -k 5 400 python -m pytest tests/test_edges.py tests/test_shim.py tests/test_reference_compiles.py tests/test_reference_pipeline.py tests/test_driver_mirror.py -m gpu -q --timeout 240 --maxfail=12 -p no:cacheprovider > $G/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $G/${T}_gpu_subset.log
tail -n 6 $G/${T}_gpu_subset.log | cut -c1-300
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 240 -k "host or preview or published or pipelined" -p no:cacheprovider 2>&1 | tail -n 3
{
for i in 1 2; do
  echo "== configs[2] through the host (run $i)"; timeout -k 5 60 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 2>&1 | tail -n 1
  echo "== configs[1] through the host (run $i)"; timeout -k 5 60 python tools/bench_through_shim.py --steps 20 --warmup 5 2>&1 | tail -n 1
done
echo "== configs[2], pageable preview buffers"; DSR_HOST_PAGEABLE_PREVIEWS=1 timeout -k 5 60 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 2>&1 | tail -n 1
echo "== round 4 library: configs[2] (two-call split not available to this host: expected to fail to start)"; true
} > $G/${T}_through_shim.log 2>&1
cut -c1-330 $G/${T}_through_shim.log
