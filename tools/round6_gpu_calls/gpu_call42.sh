# round 6, call 42: the committed fuzz file's suite seeds and smoke once more (the last GPU call of the round)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $G/r06z_fuzz_suite_last.log 2>&1; echo "fuzz suite seeds rc=$?: $(tail -n 1 $G/r06z_fuzz_suite_last.log)"
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
