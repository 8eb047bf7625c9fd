# round 6, call 38: the single-engine driver with SaveSceneToMesh at the end of every sequence: suite seeds, then seeds 4000..4500
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 300 python -m pytest tests/test_gpu_fuzz.py -k call_sequences -m gpu -q -p no:cacheprovider > $G/r06z_fuzz_mesh_suite.log 2>&1; echo "suite seeds rc=$?: $(tail -n 1 $G/r06z_fuzz_mesh_suite.log)"
grep -E "^E  " $G/r06z_fuzz_mesh_suite.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -8
SECONDS=0
DSR_FUZZ_SEEDS=4000:4500 timeout -k 5 900 python -m pytest tests/test_gpu_fuzz.py -k call_sequences -m gpu -q -p no:cacheprovider > $G/r06z_fuzz_mesh_soak.log 2>&1; echo "soak rc=$? ${SECONDS}s: $(tail -n 1 $G/r06z_fuzz_mesh_soak.log)"
grep -E "^FAILED" $G/r06z_fuzz_mesh_soak.log | head -10
grep -E "^E  " $G/r06z_fuzz_mesh_soak.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -12
