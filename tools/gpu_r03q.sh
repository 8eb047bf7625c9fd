cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03q
timeout 900 python -m pytest tests/test_multigpu_gloo.py tests/test_bench_contract.py tests/test_gpu_composite.py -m gpu -x -q > $O/${T}_multirank_gpu_tests.log 2>&1; echo "rc=$?" >> $O/${T}_multirank_gpu_tests.log
tail -n 25 $O/${T}_multirank_gpu_tests.log
