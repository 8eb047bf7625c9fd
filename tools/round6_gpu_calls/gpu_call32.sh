# round 6, call 32: the randomised differential test of the volume batch: the suite's seeds, then seeds 100..500
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_fuzz.py -k batch -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_batch_suite_seeds.log 2>&1; echo "suite seeds rc=$?: $(tail -n 1 $G/r06y_fuzz_batch_suite_seeds.log)"
grep -E "^E +" $G/r06y_fuzz_batch_suite_seeds.log | cut -c1-700 | head -12
SECONDS=0
DSR_FUZZ_BATCH_SEEDS=100:500 timeout -k 5 1500 python -m pytest tests/test_gpu_fuzz.py -k batch -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_batch_soak_100_500.log 2>&1; echo "soak rc=$? ${SECONDS}s: $(tail -n 1 $G/r06y_fuzz_batch_soak_100_500.log)"
grep -E "^FAILED" $G/r06y_fuzz_batch_soak_100_500.log | head -20
grep -E "^E +(AssertionError: batch seed|calls|visible|hash|the live|voxel|render|[a-z_]+:)" $G/r06y_fuzz_batch_soak_100_500.log | cut -c1-700 | head -30
