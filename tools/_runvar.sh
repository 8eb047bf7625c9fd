for cfg in "DSR_INTEGRATE_VARIANT=85" "DSR_INTEGRATE_VARIANT=86" "DSR_INTEGRATE_VARIANT=87" "DSR_INTEGRATE_VARIANT=86 DSR_GRID_INTEGRATE=32768"; do
  env $cfg timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/v.log
  python - "$cfg" <<PY
import json,sys
d=json.loads(open("gpurun_out/v.log").read())
print(sys.argv[1], d["value"], "integrate", d["kernels"]["integrate"]["avg_us"], "raycast", d["kernels"]["raycast"]["avg_us"])
PY
done
