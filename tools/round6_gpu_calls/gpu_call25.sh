# round 6, call 25: the randomised differential test (tests/test_gpu_fuzz.py): the suite's seeds, then a soak over seeds 100..400
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06x_fuzz_suite_seeds.log 2>&1; echo "suite seeds rc=$?: $(tail -n 1 $G/r06x_fuzz_suite_seeds.log)"
SECONDS=0
DSR_FUZZ_SEEDS=100:400 timeout -k 5 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -x > $G/r06x_fuzz_soak_100_400.log 2>&1; echo "soak rc=$? ${SECONDS}s: $(tail -n 1 $G/r06x_fuzz_soak_100_400.log)"
grep -E "^(FAILED|ERROR)|seed [0-9]+ kind" $G/r06x_fuzz_soak_100_400.log | head -5
