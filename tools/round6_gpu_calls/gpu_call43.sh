# round 6, call 43 (what is left of the budget): single-engine driver at 640x192, map-sized and swapping volumes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
for K in 2 3; do
SECONDS=0
DSR_FUZZ_KIND=$K DSR_FUZZ_SIZE=640x192 DSR_FUZZ_SEEDS=$((6000+K*100)):$((6030+K*100)) timeout -k 5 150 python -m pytest tests/test_gpu_fuzz.py -k call_sequences -m gpu -q -p no:cacheprovider -x > $G/r06z_fuzz_640_kind$K.log 2>&1; echo "kind $K rc=$? ${SECONDS}s: $(tail -n 1 $G/r06z_fuzz_640_kind$K.log)"
grep -E "^E  " $G/r06z_fuzz_640_kind$K.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -8
done
