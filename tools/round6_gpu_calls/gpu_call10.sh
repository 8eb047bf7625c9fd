# round 6, call 10: the allocation mark with its order keys de-duplicated inside a wave: whole GPU suite, then A/B against the
# library as it was before (build_variants/r06_prededup)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06j
timeout -k 5 700 python -m pytest tests -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 12 $G/${T}_gpu_suite.log
OLD=$GRAFT_REPO_ROOT/build_variants/r06_prededup/libdsr_hip.so
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg"
for L in new old new old; do
  if [ $L = old ]; then export DSR_HIP_LIB=$OLD; else unset DSR_HIP_LIB; fi
  timeout -k 5 120 $B > $G/${T}_bench_$L.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['config']['status'])"
done
for L in new old; do
  if [ $L = old ]; then export DSR_HIP_LIB=$OLD; else unset DSR_HIP_LIB; fi
  timeout -k 5 120 $B --profile-all > $G/${T}_bench_profile_all_$L.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_profile_all_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8_$L.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_instvol8_$L.json').read().strip().splitlines()[-1]); c=d['config']; print('$L', d['value'], d['unit'], d['ms_per_step'], c['chain_us_max_rank'], c['composite_us'])"
  timeout -k 5 160 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg --instances 4 > $G/${T}_bench_inst4_$L.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_inst4_$L.json').read().strip().splitlines()[-1]); print('$L configs2', d['value'], d['ms_per_step'])"
done
unset DSR_HIP_LIB
export DSR_BENCH_NO_POOL=1
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $G/ktb -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 --no-profile > $G/${T}_ktb.log 2>&1
python tools/profile_summary.py timeline $G/ktb k_batch_split 35 > $G/${T}_batch_step_timeline.json
rm -rf $G/ktb
python - <<P
import json
d=json.load(open('$G/${T}_batch_step_timeline.json'))
print('batch step', d.get('step_us'))
for k in d.get('kernels', []): print('  %-32s q%-3s %8.1f %8.1f %7.1f' % (k['name'], k['queue'], k['start_us'], k['end_us'], k['us']))
P
