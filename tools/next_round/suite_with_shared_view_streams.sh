# NEXT ROUND (~6-9 GPU-min): the whole GPU suite with DSR_PIPELINED_VIEW=2 as the process-wide setting — the evidence needed to make
# the shared view / small-volume streams the default for host-driven engines (configs[2] through the host: 439-459 -> 485-507).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r05b}
DSR_PIPELINED_VIEW=2 timeout -k 5 700 python -m pytest tests -m gpu -q --timeout 240 > $O/${T}_gpu_suite_pv2.log 2>&1; echo "suite rc=$?" >> $O/${T}_gpu_suite_pv2.log
tail -n 6 $O/${T}_gpu_suite_pv2.log
