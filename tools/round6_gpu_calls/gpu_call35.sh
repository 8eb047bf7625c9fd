# round 6, call 35: the randomised differential test of the host's per-engine pipeline (view pipeline forms 0-3): suite seeds, then seeds 100..500
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_fuzz.py -k host -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_host_suite_seeds.log 2>&1; echo "suite seeds rc=$?: $(tail -n 1 $G/r06y_fuzz_host_suite_seeds.log)"
grep -E "^E  " $G/r06y_fuzz_host_suite_seeds.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -12
SECONDS=0
DSR_FUZZ_HOST_SEEDS=100:500 timeout -k 5 1800 python -m pytest tests/test_gpu_fuzz.py -k host -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_host_soak_100_500.log 2>&1; echo "soak rc=$? ${SECONDS}s: $(tail -n 1 $G/r06y_fuzz_host_soak_100_500.log)"
grep -E "^FAILED" $G/r06y_fuzz_host_soak_100_500.log | head -20
grep -E "^E  " $G/r06y_fuzz_host_soak_100_500.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -30
