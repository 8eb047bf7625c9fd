#!/bin/bash
# tools/gpu_call.sh <timeout-seconds> <script> [args...] — rebuild whatever is stale (the snapshot ships the built .so files: a stale
# library means a wasted call), then run the script on the GPU box
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build_hip(); g.build_oracle(); g.build_hosts()" 2>&1 | grep -v "hip-link" || true
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "bash $*"
