export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmcvar
rm -rf $O; mkdir -p $O
B="python bench.py --steps 45 --warmup 5 --no-cpu-baseline --no-profile"
for v in 48 87; do
DSR_INTEGRATE_VARIANT=$v timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/i$v -o p --output-format csv -- $B > $O/i$v.log 2>&1
DSR_INTEGRATE_VARIANT=$v timeout 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/c$v -o p --output-format csv -- $B > $O/c$v.log 2>&1
python tools/profile_summary.py pmc $O/i$v $O/c$v > $O/pmc$v.json
rm -rf $O/i$v $O/c$v
done
