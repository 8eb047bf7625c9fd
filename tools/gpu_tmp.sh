cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=r03u
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/${T}_parity.log 2>&1; tail -n 1 $O/${T}_parity.log
timeout 300 python tools/bench_variants.py "" "" > $O/${T}_variants.log 2> $O/${T}.err
cat $O/${T}_variants.log
timeout 300 python bench.py --preset 5cm --steps 45 --warmup 5 --no-cpu-baseline --no-through-shim > $O/${T}_5cm.json 2>> $O/${T}.err
head -c 230 $O/${T}_5cm.json | tail -c 140; echo
