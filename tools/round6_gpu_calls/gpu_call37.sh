# round 6, call 37: a last soak of the four randomised drivers on the final library (new seeds)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
run() {  # name, env assignment, -k expression, limit
  SECONDS=0
  env $2 timeout -k 5 $4 python -m pytest tests/test_gpu_fuzz.py -k "$3" -m gpu -q -p no:cacheprovider > $G/r06z_fuzz_$1.log 2>&1
  echo "$1 ($2) rc=$? ${SECONDS}s: $(tail -n 1 $G/r06z_fuzz_$1.log)"
  grep -E "^FAILED" $G/r06z_fuzz_$1.log | head -10
  grep -E "^E  " $G/r06z_fuzz_$1.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -12
}
run single DSR_FUZZ_SEEDS=3000:3900 "call_sequences" 1000
run batch DSR_FUZZ_BATCH_SEEDS=1100:1500 "batch" 700
run host DSR_FUZZ_HOST_SEEDS=500:800 "host" 500
run scene DSR_FUZZ_SCENE_SEEDS=400:700 "sharded" 400
