# round 5, final call: the whole GPU suite on the final code, smoke, the driver's bench command, kernel-trace stats + the two HBM
# traffic passes of the configs[1] workload, kernel-trace stats of the volume batch, the 5 cm preset by library (round 3 / 4 / 5 on
# this one box), the standing lines of the other configurations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05z}
G=$GRAFT_REPO_ROOT/gpurun_out
O=$G/prof_$T
rm -rf $O; mkdir -p $O
timeout -k 5 620 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 4 $G/${T}_gpu_suite.log
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > $G/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $G/${T}_smoke.log; tail -n 1 $G/${T}_smoke.log
timeout -k 5 260 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
head -c 300 $O/bench_line.json; echo
export DSR_BENCH_NO_POOL=1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg"
timeout -k 5 80 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.log 2>&1
grep '^{"metric"' $O/kt.log > $O/bench_line_under_rocprof.json
python tools/profile_summary.py stats $O/kt 20 > $O/kernel_stats.json
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/kt/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/kt
timeout -k 5 70 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B --no-profile > $O/fetch.log 2>&1
timeout -k 5 70 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B --no-profile > $O/write.log 2>&1
python tools/profile_summary.py traffic $O/fetch $O/write 20 $O/bench_line.json > $O/pmc_traffic.json
rm -rf $O/fetch $O/write
timeout -k 5 90 rocprofv3 --kernel-trace --stats -d $O/ktb -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 --no-profile > $O/ktb.log 2>&1
python tools/profile_summary.py stats $O/ktb 20 > $O/batch_kernel_stats.json
grep '^{"metric"' $O/ktb.log > $O/bench_instvol8_under_rocprof.json
rm -rf $O/ktb
unset DSR_BENCH_NO_POOL
find $O -name "*.csv" -size +1M -delete
for L in r03 r04; do
  DSR_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/$L/libdsr_hip.so DSR_HIP_LIB_OLDER_ABI=1 timeout -k 5 100 python bench.py --preset 5cm --steps 45 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg > $O/bench_5cm_lib_$L.json 2> $O/bench_5cm_lib_$L.err; echo "5cm $L rc=$?"
done
timeout -k 5 100 python bench.py --preset 5cm --steps 45 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg > $O/bench_5cm_lib_r05.json 2> $O/bench_5cm_lib_r05.err; echo "5cm r05 rc=$?"
DSR_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/r03/libdsr_hip.so DSR_HIP_LIB_OLDER_ABI=1 timeout -k 5 100 python bench.py --preset 5cm --steps 45 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg > $O/bench_5cm_lib_r03_again.json 2>> $O/bench_5cm_lib_r03.err
for f in $O/bench_5cm_lib_*.json; do echo -n "$f "; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null || echo failed; done
B2="python bench.py --steps 40 --warmup 10 --no-cpu-baseline"
timeout -k 5 160 $B2 --instance-volumes 8 > $O/bench_instvol8.json 2>> $O/bench.err
timeout -k 5 160 $B2 --volumes 8 > $O/bench_volumes8.json 2>> $O/bench.err
timeout -k 5 160 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg --instances 4 > $O/bench_inst4.json 2>> $O/bench.err
timeout -k 5 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --instance-volumes 8 --volumes 8 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_torchrun1_both_legs.json 2>> $O/bench.err
timeout -k 5 120 python tools/bench_instance_frame.py --share-stream > $O/instance_frame_shared_stream.json 2>> $O/bench.err
timeout -k 5 120 python tools/bench_instance_frame.py > $O/instance_frame.json 2>> $O/bench.err
for f in $O/bench_instvol8.json $O/bench_volumes8.json $O/bench_inst4.json $O/bench_torchrun1_both_legs.json; do echo -n "$f "; python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'])" 2>/dev/null || echo failed; done
ls -la $O | head -40; head -c 700 $O/pmc_traffic.json; echo; head -c 900 $O/batch_kernel_stats.json
