# round 6, call 4: k_composite alone (tools/bench_composite.py): 4 vs 2 pixels per lane, aligned / unaligned planes, cover
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06d
timeout -k 5 200 python -m pytest tests/test_gpu_composite.py tests/test_reference_edges.py -m gpu -q -x --timeout 240 -p no:cacheprovider 2>&1 | tail -n 3
for PX in 4 2; do
  DSR_COMPOSITE_PX=$PX timeout -k 5 120 python tools/bench_composite.py > $G/${T}_composite_px$PX.json 2> $G/${T}_composite.err; cat $G/${T}_composite_px$PX.json; echo
done
