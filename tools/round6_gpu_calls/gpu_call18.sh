# round 6, call 18 (experiment): configs[2] through the C++ host with the instance volumes fusing on the shared view stream
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for PV in 2 3 2 3; do
  export DSR_PIPELINED_VIEW=$PV
  for ST in 20 45; do
    echo -n "pv $PV steps $ST: "; timeout -k 5 300 python tools/bench_through_shim.py --preset 5mm --steps $ST --warmup 5 --width 1242 --height 375 --instances 4 2>/dev/null | tail -n 1 | python -c "
import ast,sys
d=ast.literal_eval(sys.stdin.read()); print(d['frames_per_s'], d['ms_per_frame'], d['host_ms_inst_integrate'], d['host_ms_inst_prepare'], d['host_ms_integrate'], d['host_ms_prepare'], d['composite_hash'])"
  done
done
