# round 6, call 39: single-engine driver, swapping volumes only, with voxel GC passes in the sequences
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
SECONDS=0
DSR_FUZZ_KIND=3 DSR_FUZZ_SWAP_GC=1 DSR_FUZZ_SEEDS=5000:5250 timeout -k 5 700 python -m pytest tests/test_gpu_fuzz.py -k call_sequences -m gpu -q -p no:cacheprovider > $G/r06z_fuzz_swap_gc.log 2>&1; echo "swap+gc rc=$? ${SECONDS}s: $(tail -n 1 $G/r06z_fuzz_swap_gc.log)"
grep -E "^FAILED" $G/r06z_fuzz_swap_gc.log | head -10
grep -E "^E  " $G/r06z_fuzz_swap_gc.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -16
