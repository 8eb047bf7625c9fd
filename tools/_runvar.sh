timeout 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for cfg in "DSR_RAYCAST_FF=3 DSR_RAYCAST_FFDIST=6" "DSR_RAYCAST_FF=1000000" "DSR_RAYCAST_FF=2 DSR_RAYCAST_FFDIST=4" "DSR_RAYCAST_FF=2 DSR_RAYCAST_FFDIST=10" "DSR_RAYCAST_FF=8 DSR_RAYCAST_FFDIST=6" "DSR_RAYCAST_FF=1 DSR_RAYCAST_FFDIST=3"; do
  env $cfg timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/v.log
  python - "$cfg" <<PY
import json,sys
d=json.loads(open("gpurun_out/v.log").read())
print(sys.argv[1], d["value"], "raycast", d["kernels"]["raycast"]["avg_us"], "integrate", d["kernels"]["integrate"]["avg_us"])
PY
done
