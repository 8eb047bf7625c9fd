# round 6, call 36: the randomised differential test of ShardedScene (batch / loop, exchange slots, composite): suite seeds, then seeds 100..400
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_fuzz.py -k sharded -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_scene_suite_seeds.log 2>&1; echo "suite seeds rc=$?: $(tail -n 1 $G/r06y_fuzz_scene_suite_seeds.log)"
grep -E "^E  " $G/r06y_fuzz_scene_suite_seeds.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -12
SECONDS=0
DSR_FUZZ_SCENE_SEEDS=100:400 timeout -k 5 1800 python -m pytest tests/test_gpu_fuzz.py -k sharded -m gpu -q -p no:cacheprovider > $G/r06y_fuzz_scene_soak_100_400.log 2>&1; echo "soak rc=$? ${SECONDS}s: $(tail -n 1 $G/r06y_fuzz_scene_soak_100_400.log)"
grep -E "^FAILED" $G/r06y_fuzz_scene_soak_100_400.log | head -20
grep -E "^E  " $G/r06y_fuzz_scene_soak_100_400.log | grep -v "Use -v\|^E *$" | cut -c1-900 | head -30
