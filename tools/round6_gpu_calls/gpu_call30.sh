# round 6, call 30: replay of seed 835 (range image differs after a frame without visible blocks?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 5 200 python tests/study/fuzz_seed_debug.py 835 2>&1 | grep -v amdgpu.ids | tail -n 30
