# round 6, call 26: seed 188 of the randomised differential test, as it is and with each fast path switched off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
for SW in X=0 DSR_SMALL_LISTS=0 DSR_RAY_BOX=0 DSR_PAIR_RENDER=0 DSR_SMALL_VOLUME=0; do
  env $SW DSR_FUZZ_SEEDS=188:189 timeout -k 5 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06x_seed188_$SW.log 2>&1
  echo "$SW rc=$?: $(tail -n 1 $G/r06x_seed188_$SW.log)"; grep -A4 "^E .*calls:" $G/r06x_seed188_$SW.log | cut -c1-600 | tail -4
done
SECONDS=0
DSR_FUZZ_SEEDS=189:400 timeout -k 5 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06x_fuzz_soak_189_400.log 2>&1; echo "soak rc=$? ${SECONDS}s: $(tail -n 1 $G/r06x_fuzz_soak_189_400.log)"
grep -E "^FAILED" $G/r06x_fuzz_soak_189_400.log | head -20
