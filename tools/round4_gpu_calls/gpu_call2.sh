# round 4, second GPU call: the whole suite, what the two raycast kernels cost each, k_integrate XLDS A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=${1:-r04b}
timeout 1200 python -m pytest tests -m gpu -q > $O/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/${T}_gpu_suite.log
tail -n 12 $O/${T}_gpu_suite.log
timeout 300 python tools/ab_engine_env.py 'DSR_INTEGRATE_XLDS=0' 'DSR_INTEGRATE_XLDS=1' > $O/${T}_integrate_xlds_ab.log 2>&1; echo "xlds rc=$?" >> $O/${T}_integrate_xlds_ab.log
cat $O/${T}_integrate_xlds_ab.log | tail -n 8
timeout 200 python tools/ab_raycast_split.py --splits 64,96,160 > $O/${T}_raycast_split_counts.log 2>&1; tail -n 5 $O/${T}_raycast_split_counts.log
cd /tmp
export DSR_BENCH_NO_POOL=1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_split -o split -- python $GRAFT_REPO_ROOT/tools/ab_raycast_split.py --splits 64 > $O/${T}_prof_split.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_split96 -o split -- python $GRAFT_REPO_ROOT/tools/ab_raycast_split.py --splits 96 > $O/${T}_prof_split96.log 2>&1
cd $GRAFT_REPO_ROOT
for d in $O/${T}_prof_split $O/${T}_prof_split96; do f=$(find $d -name "*kernel_stats.csv" | head -1); echo $f; head -n 8 $f | cut -c1-160; done
find $O/${T}_prof_split* -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O/${T}_prof_split* -name "*.db" -delete
