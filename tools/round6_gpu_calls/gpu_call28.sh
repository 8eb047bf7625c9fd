# round 6, call 28: the list path stays off while the visible list holds an entry without a block: the failing seeds, a soak, the small-volume tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_suite_seeds.log 2>&1; echo "suite seeds rc=$?: $(tail -n 1 $G/r06y_fuzz_suite_seeds.log)"
SECONDS=0
DSR_FUZZ_SEEDS=100:1600 timeout -k 5 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_soak_100_1600.log 2>&1; echo "soak rc=$? ${SECONDS}s: $(tail -n 1 $G/r06y_fuzz_soak_100_1600.log)"
grep -E "^FAILED" $G/r06y_fuzz_soak_100_1600.log | head -20
grep -E "^E +(seed|calls|visible|hash|the live|voxel|render|[a-z_]+:)" $G/r06y_fuzz_soak_100_1600.log | cut -c1-500 | head -40
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_edges.py -m gpu -q -p no:cacheprovider -x > $G/r06y_subset.log 2>&1; echo "subset rc=$?: $(tail -n 1 $G/r06y_subset.log)"
