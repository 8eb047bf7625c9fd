# round 6, call 40: the fuzz file as committed (GC passes on swapping volumes too): its suite seeds, and 120 seeds of swapping volumes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06z_fuzz_suite_final.log 2>&1; echo "suite seeds rc=$?: $(tail -n 1 $G/r06z_fuzz_suite_final.log)"
DSR_FUZZ_KIND=3 DSR_FUZZ_SEEDS=5250:5370 timeout -k 5 400 python -m pytest tests/test_gpu_fuzz.py -k call_sequences -m gpu -q -p no:cacheprovider > $G/r06z_fuzz_swap_gc2.log 2>&1; echo "swap+gc rc=$?: $(tail -n 1 $G/r06z_fuzz_swap_gc2.log)"
grep -E "^E  " $G/r06z_fuzz_suite_final.log $G/r06z_fuzz_swap_gc2.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -12
