# round 6, call 27: replay of the failing seeds of the randomised differential test with a look at the entries that differ
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for S in 398 188; do echo "== seed $S"; timeout -k 5 200 python tests/study/fuzz_seed_debug.py $S 2>&1 | grep -v amdgpu.ids | tail -n 30; done
