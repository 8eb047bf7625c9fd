# round 6, call 31: the randomised differential test: the suite's seeds (incl. the three regression seeds), then seeds 1600..2800
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_suite_seeds.log 2>&1; echo "suite seeds rc=$?: $(tail -n 1 $G/r06y_fuzz_suite_seeds.log)"
SECONDS=0
DSR_FUZZ_SEEDS=1600:2800 timeout -k 5 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_soak_1600_2800.log 2>&1; echo "soak rc=$? ${SECONDS}s: $(tail -n 1 $G/r06y_fuzz_soak_1600_2800.log)"
grep -E "^FAILED" $G/r06y_fuzz_soak_1600_2800.log | head -20
grep -E "^E +(AssertionError: seed|calls|visible|hash|the live|voxel|render|[a-z_]+:)" $G/r06y_fuzz_soak_1600_2800.log | cut -c1-600 | head -40
