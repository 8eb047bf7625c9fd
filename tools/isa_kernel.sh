#!/bin/bash
# registers / loads / waits / scratch of one kernel of the library: tools/isa_kernel.sh <mangled-name-substring>
cd "$(dirname "$0")/../dynslam_amd/csrc" || exit 1
# (all translation units of the library, concatenated: tools/isa_diff.py compares two such dumps)
: > /tmp/eng.s
for tu in dsr_engine dsr_view dsr_exchange dsr_hostio dsr_profile; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fno-fast-math -Wno-unused-function --cuda-device-only -S -o /tmp/eng_$tu.s $tu.hip 2>&1 | grep -E "error" | head
  cat /tmp/eng_$tu.s >> /tmp/eng.s
done
L0=$(grep -n "^_ZN3dsr[0-9]*$1.*:" /tmp/eng.s | head -1 | cut -d: -f1); L1=$(grep -n "amdhsa_kernel _ZN3dsr[0-9]*$1" /tmp/eng.s | head -1 | cut -d: -f1)
sed -n "${L1},$((L1+60))p" /tmp/eng.s | grep -E "next_free_vgpr|private_segment_fixed"
echo "loads $(sed -n "${L0},${L1}p" /tmp/eng.s | grep -c global_load) waits $(sed -n "${L0},${L1}p" /tmp/eng.s | grep -c 's_waitcnt vmcnt') scratch $(sed -n "${L0},${L1}p" /tmp/eng.s | grep -c scratch_) lines $((L1-L0))"
