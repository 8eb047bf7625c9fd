# round 5, the final set again on the FINAL code (previews changed after gpu_final.sh): whole GPU suite, smoke, the driver's bench
# command, configs[2] / configs[1] through the C++ host four times each at 45 steps (calls 8-10 read 551-667 for configs[2])
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05zz}
G=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $G
timeout -k 5 620 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 4 $G/${T}_gpu_suite.log
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > $G/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $G/${T}_smoke.log; tail -n 1 $G/${T}_smoke.log
timeout -k 5 260 python bench.py --gpus 1 --steps 20 --warmup 5 > $G/${T}_bench_line.json 2> $G/${T}_bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$G/${T}_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: v.get('avg_us') for k, v in d['kernels'].items()}, 'shim', d['through_shim']['frames_per_s'], d['through_shim']['configs2']['frames_per_s'], 'iv8', d['instance_volumes8_1gpu']['value'])"
{
for i in 1 2 3 4; do
  echo "== configs[2] through the host, 45 steps (run $i)"; timeout -k 5 90 python tools/bench_through_shim.py --steps 45 --warmup 5 --instances 4 2>&1 | tail -n 1
done
for i in 1 2; do
  echo "== configs[1] through the host, 45 steps (run $i)"; timeout -k 5 90 python tools/bench_through_shim.py --steps 45 --warmup 5 2>&1 | tail -n 1
  echo "== configs[2] through the host, 20 steps (run $i)"; timeout -k 5 90 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 2>&1 | tail -n 1
done
} > $G/${T}_through_shim.log 2>&1
grep -o "^==.*\|'frames_per_s': '[0-9.]*'" $G/${T}_through_shim.log | paste - - 
