cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 2 17 python -c "
import bench, json
a = bench.parse_args(['--steps','20','--warmup','5'])
r = bench.through_shim(a, True)
print('shim', r['frames_per_s'], 'configs2', r['configs2'].get('frames_per_s'))
" > gpurun_out/r04s_through_shim_via_tool.log 2>&1
cat gpurun_out/r04s_through_shim_via_tool.log | tail -3
