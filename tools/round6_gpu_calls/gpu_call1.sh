# round 6, call 1: where the one-workgroup kernels spend their time (phase clocks), and the round's baselines on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06a
timeout -k 5 150 python tools/small_kernel_clocks.py --frames 64 > $G/${T}_small_kernel_clocks.json 2> $G/${T}_small_kernel_clocks.err; echo "clocks rc=$?"
cat $G/${T}_small_kernel_clocks.json
timeout -k 5 120 python tools/bench_instance_frame.py --share-stream > $G/${T}_instance_frame_shared_stream.json 2> $G/${T}_if.err; echo "if rc=$?"
python -c "
import json
d=json.loads(open('$G/${T}_instance_frame_shared_stream.json').read().strip().splitlines()[-1]); print(d['free_running'], d['sync_per_frame'], d['gpu_kernels'])"
timeout -k 5 260 python bench.py --gpus 1 --steps 20 --warmup 5 > $G/${T}_bench_line.json 2> $G/${T}_bench.err; echo "bench rc=$?"
head -c 1500 $G/${T}_bench_line.json; echo
