# round 6, call 29: seed 835 of the randomised differential test with each fast path switched off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
for SW in X=0 DSR_SMALL_LISTS=0 DSR_RAY_BOX=0 DSR_PAIR_RENDER=0 DSR_SMALL_VOLUME=0; do
  env $SW DSR_FUZZ_SEEDS=835:836 timeout -k 5 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06y_seed835_$SW.log 2>&1
  echo "$SW rc=$?: $(tail -n 1 $G/r06y_seed835_$SW.log)"; grep -E "^E +(render|visible|hash|[a-z_]+:)" $G/r06y_seed835_$SW.log | cut -c1-300 | tail -2
done
