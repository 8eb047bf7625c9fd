# round 6, call 12: the two new GPU tests (mask ring growth inside a split; the list path through GC / exhaustion / reset)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_edges.py tests/test_gpu_parity.py -m gpu -q -x --timeout 240 -p no:cacheprovider -k "mask_ring or list_path" 2>&1 | tail -n 15
DSR_SMALL_LISTS=0 timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 240 -p no:cacheprovider -k "list_path" 2>&1 | tail -n 5
DSR_PAIR_RENDER=0 DSR_RAY_BOX=0 timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 240 -p no:cacheprovider -k "list_path or instance_volume" 2>&1 | tail -n 5
