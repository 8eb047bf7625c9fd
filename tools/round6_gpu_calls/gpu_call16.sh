# round 6, call 16: configs[4] sustained on the final code (4541 frames, 4 mm, voxel GC 1 / 200 + host swapping, invariants at 5 checkpoints)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 500 python tools/bench_cfg5_sustained.py > $G/r06_cfg5_sustained_4541frames.json 2> $G/r06_cfg5.err; echo "rc=$?"
python -c "
import json
d=json.loads(open('$G/r06_cfg5_sustained_4541frames.json').read().strip().splitlines()[-1]); print(d['frames_per_s'], d['frames_per_s_first_500'], d['frames_per_s_last_500'], d['peak_hbm_used_GB'], d['pinned_host_GB'], d['status'], d['structure_checks'])"
