# round 4, sixth GPU call: the shared-stream form of the pipelined view, then the standing lines of the round
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=${1:-r04f}
run_cfg2() { name=$1; shift
  env DSR_BENCH_NO_POOL=1 "$@" timeout -k 5 100 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 > $O/${T}_shim_cfg2_${name}.log 2>&1
  echo "cfg2 $name: $(tail -n 1 $O/${T}_shim_cfg2_${name}.log | cut -c1-130)"; }
run_cfg2 pv0 DSR_PIPELINED_VIEW=0
run_cfg2 pv2 DSR_PIPELINED_VIEW=2
run_cfg2 pv2_prio DSR_PIPELINED_VIEW=2 DSR_STREAM_PRIORITY=1
run_cfg2 pv2_q8 DSR_PIPELINED_VIEW=2 GPU_MAX_HW_QUEUES=8
timeout -k 5 200 python -m pytest -m gpu -q -x --timeout 150 tests/test_edges.py "tests/test_gpu_parity.py::test_host_buffer_frames_pipelined_without_waiting" > $O/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $O/${T}_gpu_subset.log; tail -n 4 $O/${T}_gpu_subset.log
timeout -k 5 300 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_line.json 2> $O/${T}_bench.err; echo "bench rc=$?"
timeout -k 5 200 python bench.py --instances 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_inst4.json 2>> $O/${T}_bench.err
timeout -k 5 200 python bench.py --instance-volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_instvol8.json 2>> $O/${T}_bench.err
timeout -k 5 200 python bench.py --volumes 8 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_volumes8.json 2>> $O/${T}_bench.err
timeout -k 5 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --instance-volumes 8 --volumes 8 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_torchrun1_both_legs.json 2>> $O/${T}_bench.err
timeout -k 5 150 python tools/bench_instance_frame.py > $O/${T}_instance_frame.json 2>> $O/${T}_bench.err
timeout -k 5 200 python bench.py --preset 5cm --steps 45 --warmup 5 --no-cpu-baseline > $O/${T}_bench_5cm.json 2>> $O/${T}_bench.err
for f in $O/${T}_bench_*.json $O/${T}_instance_frame.json; do echo $f; grep '^{' $f | head -c 260 | tail -c 170; echo; done
tail -n 3 $O/${T}_bench.err
