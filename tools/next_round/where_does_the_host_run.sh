# NEXT ROUND, first GPU call (~1.5 GPU-min): why the identical host_bench command reads 223-275 frames/s for configs[2] from bench.py's
# process and 439-460 from tools/bench_through_shim.py.  host_bench now prints where its time goes per call site
# (host_ms_*) and which CPU it ran on (cpu=first-last): compare the two parents, then pin the child.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r05a}
L=$O/${T}_where_does_the_host_run.log
: > $L
{
  echo "== topology"; nproc; (command -v numactl >/dev/null && numactl --hardware) || echo "no numactl"
  for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null)"; done
  grep -E "Cpus_allowed_list|Mems_allowed_list" /proc/self/status
} >> $L 2>&1
tool() { timeout -k 5 60 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 2>&1 | tail -n 1; }
echo "== the tool" >> $L; tool >> $L
echo "== the tool, child pinned to the GPU's local CPUs" >> $L
CPUS=$(cat /sys/class/drm/card0/device/local_cpulist 2>/dev/null)
[ -n "$CPUS" ] && (taskset -c $CPUS bash -c "$(declare -f tool); tool") >> $L 2>&1
echo "== from a parent that holds three frame sets (what bench.py's process looked like in round 4)" >> $L
timeout -k 5 120 python - >> $L 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench
from bench_through_shim import run
from dynslam_amd.synth import StreetScene
a = bench.parse_args(["--steps", "20", "--warmup", "5"])
f0, f8, f4 = bench.frames_for(a, 0), bench.frames_for(a, 8), bench.frames_for(a, 4)
intr = StreetScene(1242, 375).intrinsics()
exe = os.path.join(os.getcwd(), "shim", "host_bench")
print("parent cpu", os.sched_getcpu(), "affinity", len(os.sched_getaffinity(0)))
print(run(exe, f4, 1242, 375, intr, bench.settings_kwargs("5mm"), 5, instances=4))
del f8, f0
import gc; gc.collect()
print("after dropping two sets:", run(exe, f4, 1242, 375, intr, bench.settings_kwargs("5mm"), 5, instances=4))
PY
cut -c1-420 $L
