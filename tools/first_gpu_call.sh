# First GPU call of a round: everything that was written without a GPU gets its first run, then the standing numbers.
# Run on the GPU box from the repo root:   bash tools/first_gpu_call.sh <tag>      (writes gpurun_out/<tag>_*)
# Every step has its own timeout; nothing here combines PMC counters with tracing.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03a}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
# 1. the tests that are xfail-guarded until their first GPU run (remove the markers when they pass)
timeout 600 python -m pytest tests/test_reference_pipeline.py tests/test_reference_edges.py tests/test_gpu_parity.py -m gpu -q -rxX -k "pipeline or special_values or launch_geometry" > $O/${TAG}_new_gpu_tests.log 2>&1
# 2. the whole GPU suite
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_suite.log 2>&1
# 3. the bench line the driver records, and the configs[3] leg on one GPU under torchrun (RCCL path, world 1)
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --volumes 8 --steps 20 --warmup 5 \
  > $O/${TAG}_bench_configs3_torchrun_1rank.json 2>> $O/${TAG}_bench.err
# 4. the reference's own per-frame loop on the HIP engine, timed (host-bound by design: pageable views, CPU silhouette split)
D=/tmp/kitti_like_$TAG; rm -rf $D; mkdir -p $D
timeout 300 python tests/refhost/make_dataset.py $D 16 > /dev/null 2>&1
( cd $D && /usr/bin/time -v timeout 300 $GRAFT_REPO_ROOT/tests/refhost/_build/ref_dynslam_host $D 16 $D/out.bin 0.05 1 0 ) > $O/${TAG}_ref_pipeline_hip.log 2>&1
grep -E "^Timer: (Static map fusion|Instance tracking|Input preprocessing|Map decay)" $O/${TAG}_ref_pipeline_hip.log | sort | uniq -c | sort -rn | head -40 > $O/${TAG}_ref_pipeline_timers.txt
tail -3 $O/${TAG}_new_gpu_tests.log $O/${TAG}_gpu_suite.log
cat $O/${TAG}_bench_line.json | head -c 600
