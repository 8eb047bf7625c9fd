# round 6, final call: the whole GPU suite on the final code, smoke, the driver's bench command, kernel-trace stats + the two HBM
# traffic passes of the configs[1] workload, kernel-trace stats + one step's timeline of the volume batch, the instance frame, the
# standing lines of the other configurations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r06z}
G=$GRAFT_REPO_ROOT/gpurun_out
O=$G/prof_$T
rm -rf $O; mkdir -p $O
timeout -k 5 700 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 4 $G/${T}_gpu_suite.log
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > $G/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $G/${T}_smoke.log; tail -n 1 $G/${T}_smoke.log
SECONDS=0; timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? wall ${SECONDS}s"
head -c 300 $O/bench_line.json; echo
export DSR_BENCH_NO_POOL=1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg"
timeout -k 5 100 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.log 2>&1
grep '^{"metric"' $O/kt.log > $O/bench_line_under_rocprof.json
python tools/profile_summary.py stats $O/kt 20 > $O/kernel_stats.json
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/kt/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/kt
timeout -k 5 100 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B --no-profile > $O/fetch.log 2>&1
timeout -k 5 100 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B --no-profile > $O/write.log 2>&1
python tools/profile_summary.py traffic $O/fetch $O/write 20 $O/bench_line.json > $O/pmc_traffic.json
rm -rf $O/fetch $O/write
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $O/ktb -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 --no-profile > $O/ktb.log 2>&1
python tools/profile_summary.py stats $O/ktb 20 > $O/batch_kernel_stats.json
python tools/profile_summary.py timeline $O/ktb k_batch_split 35 > $O/batch_step_timeline.json
grep '^{"metric"' $O/ktb.log > $O/bench_instvol8_under_rocprof.json
rm -rf $O/ktb
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $O/kti -o kt --output-format csv -- python tools/bench_instance_frame.py --share-stream --frames 60 > $O/kti.log 2>&1
python tools/profile_summary.py timeline $O/kti k_view_ingest 150 > $O/instance_frame_timeline.json
rm -rf $O/kti
unset DSR_BENCH_NO_POOL
find $O -name "*.csv" -size +1M -delete
B2="python bench.py --steps 40 --warmup 10 --no-cpu-baseline"
timeout -k 5 160 $B2 --instance-volumes 8 > $O/bench_instvol8.json 2>> $O/bench.err
timeout -k 5 160 $B2 --volumes 8 > $O/bench_volumes8.json 2>> $O/bench.err
timeout -k 5 160 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg --instances 4 > $O/bench_inst4.json 2>> $O/bench.err
timeout -k 5 160 python bench.py --steps 45 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg --preset 5cm > $O/bench_5cm.json 2>> $O/bench.err
timeout -k 5 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --instance-volumes 8 --volumes 8 --map-volumes 2 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_torchrun1_legs.json 2>> $O/bench.err
timeout -k 5 120 python tools/bench_instance_frame.py --share-stream > $O/instance_frame_shared_stream.json 2>> $O/bench.err
timeout -k 5 120 python tools/bench_instance_frame.py > $O/instance_frame.json 2>> $O/bench.err
timeout -k 5 120 python tools/bench_instance_frame.py --share-stream --reset-every 16 > $O/instance_frame_shared_stream_allocating.json 2>> $O/bench.err
timeout -k 5 150 python tools/small_kernel_clocks.py --frames 64 > $O/small_kernel_clocks.json 2>> $O/bench.err
timeout -k 5 120 python tools/bench_composite.py > $O/composite_alone.json 2>> $O/bench.err
( timeout -k 5 300 python tools/bench_through_shim.py --preset 5mm --steps 20 --warmup 5 --width 1242 --height 375 2>/dev/null | tail -n 1; timeout -k 5 300 python tools/bench_through_shim.py --preset 5mm --steps 20 --warmup 5 --width 1242 --height 375 --instances 4 2>/dev/null | tail -n 1; timeout -k 5 300 python tools/bench_through_shim.py --preset 5mm --steps 45 --warmup 5 --width 1242 --height 375 --instances 4 2>/dev/null | tail -n 1 ) > $O/through_shim.log
for f in $O/bench_instvol8.json $O/bench_volumes8.json $O/bench_inst4.json $O/bench_5cm.json $O/bench_torchrun1_legs.json; do echo -n "$f "; python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'])" 2>/dev/null || echo failed; done
python -c "
import json
for n in ('instance_frame_shared_stream','instance_frame','instance_frame_shared_stream_allocating'):
    d=json.loads(open('$O/'+n+'.json').read().strip().splitlines()[-1]); print(n, d['free_running']['us_per_frame'], d['sync_per_frame'], d['launches_per_frame'])"
cat $O/through_shim.log | cut -c1-200
ls -la $O | head -40; head -c 700 $O/pmc_traffic.json; echo
for f in $O/*; do mv $f $G/${T}_$(basename $f); done; rmdir $O
