timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "sequence or collisions" 2>&1 | tail -2
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --profile-all 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['avg_us'] for k,v in d['kernels'].items() if k in ('raycast','alloc_mark','integrate')})"; done
timeout 200 python bench.py --no-cpu-baseline --preset 5cm 2>&1 | tail -1 | cut -c1-230
