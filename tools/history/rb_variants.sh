#!/bin/bash
# ffn_probe against the shipped library and the -DS2T_RB_DBG variants built by tools/rb_dbg_build.sh
cd "$(dirname "$0")/.."
echo "== shipped"; python tools/ffn_probe.py 2>&1 | grep -v amdgpu.ids
for d in s2t_amd/lib/rbdbg*; do
  echo "== $d"; S2T_HIP_LIB=$PWD/$d/libs2t_hip.so python tools/ffn_probe.py 2>&1 | grep -v amdgpu.ids
done
