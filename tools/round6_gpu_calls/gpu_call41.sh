# round 6, call 41: single-engine driver at 640x192, instance-sized volumes (more than kSmallNewMax new entries in a frame, visible lists that overflow)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
for K in 0 1; do
SECONDS=0
DSR_FUZZ_KIND=$K DSR_FUZZ_SIZE=640x192 DSR_FUZZ_SEEDS=$((6000+K*100)):$((6060+K*100)) timeout -k 5 330 python -m pytest tests/test_gpu_fuzz.py -k call_sequences -m gpu -q -p no:cacheprovider > $G/r06z_fuzz_640_kind$K.log 2>&1; echo "kind $K rc=$? ${SECONDS}s: $(tail -n 1 $G/r06z_fuzz_640_kind$K.log)"
grep -E "^FAILED" $G/r06z_fuzz_640_kind$K.log | head -10
grep -E "^E  " $G/r06z_fuzz_640_kind$K.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -12
done
