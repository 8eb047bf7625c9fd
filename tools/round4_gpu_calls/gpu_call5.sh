# round 4, fifth GPU call: the WHOLE -m gpu suite on the final code (per-test timeout), then the integrate variants once more
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp DSR_BENCH_NO_POOL=1
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=${1:-r04e}
timeout -k 5 540 python -m pytest tests -m gpu -q --timeout 240 > $O/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/${T}_gpu_suite.log
tail -n 8 $O/${T}_gpu_suite.log
timeout -k 5 150 python tools/ab_engine_env.py 'DSR_INTEGRATE_XLDS=0' 'DSR_INTEGRATE_XLDS=1' > $O/${T}_integrate_ab.log 2>&1; tail -n 6 $O/${T}_integrate_ab.log
