# round 6, call 15: phase clocks of the one-workgroup kernels in frames that allocate (the instance volume reset every 16 frames)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
for R in 0 16; do
timeout -k 5 150 python tools/small_kernel_clocks.py --frames 64 --reset-every $R > $G/r06o_small_kernel_clocks_reset$R.json 2> $G/r06o_clk.err; cat $G/r06o_small_kernel_clocks_reset$R.json; echo
done
