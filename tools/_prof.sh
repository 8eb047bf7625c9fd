set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r01d
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1 || true
B="python bench.py --no-cpu-baseline --no-profile"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py > $O/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B > $O/write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/inst -o p --output-format csv -- $B > $O/inst.log 2>&1
timeout 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU -d $O/cyc -o p --output-format csv -- $B > $O/cyc.log 2>&1
python tools/profile_summary.py stats $O/kt > $O/stats.json
python tools/profile_summary.py pmc $O/fetch $O/write $O/inst $O/cyc > $O/pmc.json
tail -3 $O/*.log
# keep the merged output small
find $O -name "*.csv" -size +2M -delete
du -sh $O
