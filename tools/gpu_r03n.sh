cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03n
timeout 300 python -m pytest tests/test_edges.py tests/test_shim.py -m gpu -x -q > $O/${T}_edges.log 2>&1; tail -n 1 $O/${T}_edges.log
I="timeout 300 python bench.py --instances 4 --steps 20 --warmup 5 --no-cpu-baseline"
$I > $O/${T}_inst4_overlap1.json 2>> $O/${T}.err
DSR_OVERLAP_EXPECTED=0 $I > $O/${T}_inst4_overlap0.json 2>> $O/${T}.err
V="timeout 300 python bench.py --instance-volumes 8 --steps 40 --warmup 10"
$V > $O/${T}_instvol8_overlap1.json 2>> $O/${T}.err
DSR_OVERLAP_EXPECTED=0 $V > $O/${T}_instvol8_overlap0.json 2>> $O/${T}.err
timeout 300 python tools/bench_through_shim.py --instances 4 > $O/${T}_through_shim_configs2.log 2>> $O/${T}.err
timeout 300 python tools/bench_through_shim.py --instances 4 --preset 5cm >> $O/${T}_through_shim_configs2.log 2>> $O/${T}.err
for f in $O/${T}_inst*.json; do echo $f; head -c 230 $f | tail -c 140; echo; done
cat $O/${T}_through_shim_configs2.log
