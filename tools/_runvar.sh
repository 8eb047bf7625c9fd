DSR_RAYCAST_TILEMODE=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "sequence or render" 2>&1 | tail -2
for cfg in "DSR_RAYCAST_TILEMODE=0" "DSR_RAYCAST_TILEMODE=1" "DSR_RAYCAST_TILEMODE=0" "DSR_RAYCAST_TILEMODE=1"; do
  env $cfg timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/v.log
  python - "$cfg" <<PY
import json,sys
d=json.loads(open("gpurun_out/v.log").read())
print(sys.argv[1], d["value"], "integrate", d["kernels"]["integrate"]["avg_us"], "raycast", d["kernels"]["raycast"]["avg_us"])
PY
done
