# The complete GPU suite on the final code of a round + the standing bench lines.  bash tools/gpu_final.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=${1:-final}
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/${T}_gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${T}_smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_line.json 2> $O/${T}_bench.err
timeout 300 python bench.py --instances 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_inst4.json 2>> $O/${T}_bench.err
timeout 300 python bench.py --instance-volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_instvol8.json 2>> $O/${T}_bench.err
timeout 300 python bench.py --volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_volumes8.json 2>> $O/${T}_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --instance-volumes 8 --volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_torchrun1_both_legs.json 2>> $O/${T}_bench.err
timeout 200 python tools/bench_instance_frame.py > $O/${T}_instance_frame.json 2>> $O/${T}_bench.err
timeout 300 python bench.py --preset 5cm --steps 45 --warmup 5 --no-cpu-baseline > $O/${T}_bench_5cm.json 2>> $O/${T}_bench.err
tail -n 3 $O/${T}_gpu_suite.log; tail -n 2 $O/${T}_smoke.log
for f in $O/${T}_bench_*.json; do echo $f; grep '^{' $f | head -c 260 | tail -c 160; echo; done
