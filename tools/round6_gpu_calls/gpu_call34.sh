# round 6, call 34: batch fuzz after the two fixes (status words of a blank-only frame; a reset keeps the ray box's LAST / POSE record): suite seeds, 412, seeds 500..1100
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_suite_seeds.log 2>&1; echo "suite seeds rc=$?: $(tail -n 1 $G/r06y_fuzz_suite_seeds.log)"
DSR_FUZZ_BATCH_SEEDS=412:413 timeout -k 5 300 python -m pytest tests/test_gpu_fuzz.py -k batch -m gpu -q -p no:cacheprovider > $G/r06y_bseed412.log 2>&1; echo "412 rc=$?: $(tail -n 1 $G/r06y_bseed412.log)"
SECONDS=0
DSR_FUZZ_BATCH_SEEDS=500:1100 timeout -k 5 1500 python -m pytest tests/test_gpu_fuzz.py -k batch -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_batch_soak_500_1100.log 2>&1; echo "soak rc=$? ${SECONDS}s: $(tail -n 1 $G/r06y_fuzz_batch_soak_500_1100.log)"
grep -E "^FAILED" $G/r06y_fuzz_batch_soak_500_1100.log | head -20
grep -E "^E  " $G/r06y_fuzz_batch_soak_500_1100.log | grep -v "Use -v\|^E *$" | cut -c1-700 | head -30
