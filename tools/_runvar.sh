mkdir -p gpurun_out/r01e
python bench.py --instances 4 --no-cpu-baseline > gpurun_out/r01e/bench_instances4.log 2>/dev/null
python bench.py --instances 4 --preset 5cm --no-cpu-baseline > gpurun_out/r01e/bench_instances4_5cm.log 2>/dev/null
python bench.py --preset 4mm --decay --swap --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r01e/bench_cfg5_4mm_decay_swap.log 2>/dev/null
python bench.py --preset 5cm --no-cpu-baseline > gpurun_out/r01e/bench_5cm.log 2>/dev/null
python bench.py --preset 3.5cm --no-cpu-baseline > gpurun_out/r01e/bench_3.5cm.log 2>/dev/null
for f in gpurun_out/r01e/*.log; do python - $f <<PY
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r.get("achieved"), r.get("frac"), d["config"]["status"])
PY
done
