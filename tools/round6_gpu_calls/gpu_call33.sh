# round 6, call 33: batch fuzz seeds 2, 219, 266 (status words of a blank-only frame) and 412 (raycast result after a reset), 412 with switches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
for S in 2 219 266; do DSR_FUZZ_BATCH_SEEDS=$S:$((S+1)) timeout -k 5 300 python -m pytest tests/test_gpu_fuzz.py -k batch -m gpu -q -p no:cacheprovider > $G/r06y_bseed$S.log 2>&1; echo "seed $S rc=$?: $(tail -n 1 $G/r06y_bseed$S.log)"; grep -E "^E +(volume|render|visible|hash|[a-z_]+:)" $G/r06y_bseed$S.log | cut -c1-500 | tail -2; done
for SW in X=0 DSR_SMALL_LISTS=0 DSR_RAY_BOX=0 DSR_PAIR_RENDER=0; do
  env $SW DSR_FUZZ_BATCH_SEEDS=412:413 timeout -k 5 300 python -m pytest tests/test_gpu_fuzz.py -k batch -m gpu -q -p no:cacheprovider > $G/r06y_bseed412_$SW.log 2>&1
  echo "412 $SW rc=$?: $(tail -n 1 $G/r06y_bseed412_$SW.log)"; grep -E "^E +(volume|render|visible|hash|[a-z_]+:)" $G/r06y_bseed412_$SW.log | cut -c1-600 | tail -2
done
