# The complete GPU set on the final code of a round.  Round 5's form: tools/gpu_call.sh 1500 tools/round5_gpu_calls/gpu_final.sh <tag>
# (rebuilds whatever is stale, then the whole GPU suite, smoke, the driver's bench command, rocprofv3 kernel-trace stats and the two
# HBM traffic passes of configs[1], kernel-trace stats of the volume batch, the 5 cm preset by library, the standing lines).
exec bash "$(dirname "$0")/round5_gpu_calls/gpu_final.sh" "$@"
