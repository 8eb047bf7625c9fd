# r03a: first GPU call of round 3 — the GPU suite (new tests included), the standing bench line, the new multi-volume modes on one GPU,
# and the instance-frame breakdown.  bash tools/gpu_r03a.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03a
timeout 900 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/${T}_gpu_suite.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_line.json 2> $O/${T}_bench.err
timeout 300 python bench.py --instance-volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_instvol8.json 2>> $O/${T}_bench.err
timeout 300 python bench.py --volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_volumes8.json 2>> $O/${T}_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --instance-volumes 8 --steps 40 --warmup 10 \
  > $O/${T}_bench_instvol8_torchrun1.json 2>> $O/${T}_bench.err
timeout 200 python tools/bench_instance_frame.py > $O/${T}_instance_frame.json 2>> $O/${T}_bench.err
timeout 200 python tools/bench_instance_frame.py --host-masks --two-renders > $O/${T}_instance_frame_r2style.json 2>> $O/${T}_bench.err
tail -n 4 $O/${T}_gpu_suite.log
head -c 400 $O/${T}_bench_line.json; echo
head -c 1500 $O/${T}_instance_frame.json; echo
tail -n 5 $O/${T}_bench.err
