cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03m
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim"
$B > $O/${T}_bench_lds_1.json 2>> $O/${T}_bench.err
DSR_OVERLAP_KERNEL=1 $B > $O/${T}_bench_glob_1.json 2>> $O/${T}_bench.err
$B > $O/${T}_bench_lds_2.json 2>> $O/${T}_bench.err
DSR_OVERLAP_KERNEL=1 $B > $O/${T}_bench_glob_2.json 2>> $O/${T}_bench.err
DSR_OVERLAP_KERNEL=1 $B --profile-all > $O/${T}_bench_glob_profile_all.json 2>> $O/${T}_bench.err
DSR_OVERLAP_KERNEL=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sequence or pose_or_list or geometry or odd" > $O/${T}_parity.log 2>&1
tail -n 1 $O/${T}_parity.log
for f in $O/${T}_bench_*.json; do echo $f; head -c 230 $f | tail -c 110; echo; done
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03m_bench_glob_profile_all.json"))
print({k:v["avg_us"] for k,v in d["kernels"].items()})
PY
