# r03j: priority of the side stream that computes the range image under the integration.  bash tools/gpu_r03j.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03j
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim"
for i in 1 2; do
$B > $O/${T}_bench_prio_high_$i.json 2>> $O/${T}_bench.err
DSR_SIDE_PRIORITY=0 $B > $O/${T}_bench_prio_same_$i.json 2>> $O/${T}_bench.err
DSR_OVERLAP_EXPECTED=0 $B > $O/${T}_bench_no_overlap_$i.json 2>> $O/${T}_bench.err
done
DSR_VARIANTS_PROFILE_ALL=1 timeout 300 python tools/bench_variants.py "" > $O/${T}_variants.log 2>> $O/${T}_bench.err
for f in $O/${T}_bench_*.json; do echo $f; head -c 230 $f | tail -c 110; echo; done
cut -c1-500 $O/${T}_variants.log
