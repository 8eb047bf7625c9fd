# round 6, call 17: the integration grid of an instance-sized volume (default 1785 workgroups for 7142 blocks; ~500 visible)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
for GI in 0 128 256 512 1024; do
  if [ $GI = 0 ]; then unset DSR_GRID_INTEGRATE; else export DSR_GRID_INTEGRATE=$GI; fi
  timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 > $G/r06p_bench_instvol8_grid$GI.json 2>> $G/r06p_bench.err
  python -c "
import json
d=json.loads(open('$G/r06p_bench_instvol8_grid$GI.json').read().strip().splitlines()[-1]); c=d['config']; print('grid $GI instvol8', d['value'], d['ms_per_step'], c['chain_us_max_rank'])"
  timeout -k 5 120 python tools/bench_instance_frame.py --share-stream > $G/r06p_instance_frame_grid$GI.json 2>> $G/r06p_bench.err
  python -c "
import json
d=json.loads(open('$G/r06p_instance_frame_grid$GI.json').read().strip().splitlines()[-1]); print('grid $GI instance frame', d['free_running']['us_per_frame'], d['gpu_kernels']['inst:integrate']['us_per_frame'])"
done
