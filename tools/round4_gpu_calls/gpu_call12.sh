# the driver's own bench command on a fresh box, then the no-flag default (which read 498 frames/s once, behind the suite)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r04x}
export DSR_BENCH_STEP_TIMES=1
timeout -k 5 150 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_cmd.json 2> $O/${T}_bench.err
head -c 330 $O/${T}_bench_driver_cmd.json | tail -c 250; echo
timeout -k 5 170 python bench.py > $O/${T}_bench_default.json 2>> $O/${T}_bench.err
head -c 330 $O/${T}_bench_default.json | tail -c 250; echo
