# round 5, last call: the driver's three round-end commands on the final commit (GPU suite, smoke, bench.py defaults)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05final}
G=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $G
timeout -k 5 620 python -m pytest tests -x -q -m gpu --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 3 $G/${T}_gpu_suite.log
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > $G/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $G/${T}_smoke.log; tail -n 1 $G/${T}_smoke.log
timeout -k 5 300 python bench.py > $G/${T}_bench_line.json 2> $G/${T}_bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$G/${T}_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], 'shim', d['through_shim']['frames_per_s'], d['through_shim']['configs2']['frames_per_s'], 'iv8', d['instance_volumes8_1gpu']['value'])"
