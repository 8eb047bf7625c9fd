# One box, the instance-frame tool at several commits (worktrees under build_variants/, built beforehand): which change moved
# the free-running frame.  bash tools/round4_gpu_calls/gpu_call_bisect.sh <tag> <sha>...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=$1; shift
L=$O/${T}_instance_frame_by_commit.log
: > $L
run() {  # label, dir
  (cd $2 && timeout -k 5 90 python tools/bench_instance_frame.py 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'commit':'$1','free_running_us':d['free_running']['us_per_frame'],'host_enqueue_us':d['free_running']['host_enqueue_us_per_frame'],'sync_per_frame_us':d['sync_per_frame']['us_per_frame'],'gpu_us':d['gpu_us_per_frame'],'host_us_per_call':d['free_running']['host_us_per_call']}))") >> $L
}
run HEAD $GRAFT_REPO_ROOT
for s in "$@"; do run $s $GRAFT_REPO_ROOT/build_variants/wt_$s; done
run HEAD-again $GRAFT_REPO_ROOT
cat $L
