timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for cfg in "DSR_GRID_EXPECTED=128" "DSR_GRID_EXPECTED=96" "DSR_GRID_EXPECTED=160" "DSR_GRID_EXPECTED=192"; do
  env $cfg timeout 200 python bench.py --no-cpu-baseline --profile-all 2>&1 | tail -1 > gpurun_out/v.log
  python - "$cfg" <<PY
import json,sys
d=json.loads(open("gpurun_out/v.log").read())
print(sys.argv[1], d["value"], "expected", d["kernels"]["expected_depth"]["avg_us"])
PY
done
