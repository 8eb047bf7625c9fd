# round 4: instance-volume frame with the folded scans / fused preview shading / 8-lane range image — A/B on one box, then the
# parity tests that run small volumes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=${1:-r04i}
L=$O/${T}_instance_frame_ab.log
: > $L
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout -k 5 90 python tools/bench_instance_frame.py 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['gpu_kernels']
print(json.dumps({'variant':'$label','free_running_us':d['free_running']['us_per_frame'],'sync_per_frame_us':d['sync_per_frame']['us_per_frame'],'gpu_us':d['gpu_us_per_frame'],'launches':d['launches_per_frame'],'expected_depth_us':k.get('inst:expected_depth',{}).get('us_per_frame'),'alloc_mark_us':k.get('inst:alloc_mark',{}).get('us_per_frame'),'visible_count_us':k.get('inst:visible_count',{}).get('us_per_frame'),'raycast_freeview_us':k.get('inst:raycast_freeview',{}).get('us_per_frame')}))" >> $L
}
run "scans and render as separate launches" DSR_FOLD_SCANS=0 DSR_FUSE_RENDER=0
run "folded scans" DSR_FOLD_SCANS=1 DSR_FUSE_RENDER=0
run "folded scans + fused render (default)" DSR_FOLD_SCANS=1 DSR_FUSE_RENDER=1
run "separate again" DSR_FOLD_SCANS=0 DSR_FUSE_RENDER=0
cat $L
timeout -k 5 120 python bench.py --instance-volumes 8 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_instvol8.json 2> $O/${T}_bench.err
grep '^{' $O/${T}_bench_instvol8.json | head -c 230 | tail -c 140; echo
timeout -k 5 300 python -m pytest tests/test_edges.py tests/test_gpu_parity.py -m gpu -q -x --timeout 120 -k "instance or launch_geometry or freeview or free_view or published" > $O/${T}_subset.log 2>&1; echo "subset rc=$?" >> $O/${T}_subset.log
tail -n 6 $O/${T}_subset.log
