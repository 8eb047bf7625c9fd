timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_swapping.py tests/test_golden.py -x -q 2>&1 | tail -2
for cfg in "DSR_GRID_DECAY=32768" "DSR_GRID_DECAY=8192"; do
  env $cfg timeout 200 python bench.py --decay --decay-min-age 10 --no-cpu-baseline --profile-all 2>/dev/null | tail -1 > gpurun_out/v.log
  python - "$cfg" <<PY
import json,sys
d=json.loads(open("gpurun_out/v.log").read())
print(sys.argv[1], d["value"], "decay_blocks", d["kernels"]["decay_blocks"]["avg_us"])
PY
done
