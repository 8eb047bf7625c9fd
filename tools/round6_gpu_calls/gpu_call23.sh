# round 6, call 23: the parity tests with each of this round's fast paths switched OFF (the fall-back paths INTEGRATION.md's table names)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T="tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_edges.py tests/test_gpu_fullsize_golden.py tests/test_swapping.py tests/test_gpu_composite.py tests/test_multigpu_gloo.py"
for SW in DSR_SMALL_LISTS=0 DSR_RAY_BOX=0 DSR_PAIR_RENDER=0 DSR_SMALL_VOLUME=0 DSR_OVERLAP_EXPECTED=0; do
  SECONDS=0
  env $SW timeout -k 5 900 python -m pytest $T -m gpu -q -p no:cacheprovider > $G/r06v_suite_$SW.log 2>&1
  echo "$SW rc=$? ${SECONDS}s: $(tail -n 1 $G/r06v_suite_$SW.log)"
  grep -E "^(FAILED|ERROR)" $G/r06v_suite_$SW.log | head -8
done
