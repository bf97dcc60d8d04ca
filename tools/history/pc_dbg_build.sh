#!/bin/bash
# Build experiment variants of libs2t_hip.so with -DS2T_PC_DBG=<n> for ffn_pc.hip into s2t_amd/lib/pcdbg<n>/
# (use with S2T_HIP_LIB=...).  Extra hipcc flags after "--".
set -e
cd "$(dirname "$0")/.."
extra=""
ns=()
for a in "$@"; do if [ "$a" = "--" ]; then shift; extra="$*"; break; fi; ns+=("$a"); shift; done
for n in "${ns[@]}"; do
  d=s2t_amd/lib/pcdbg$n; mkdir -p $d; objs=""
  for f in s2t_amd/csrc/*.hip; do
    o=s2t_amd/lib/obj/$(basename ${f%.hip}).o
    if [ "$(basename $f)" = ffn_pc.hip ]; then o=$d/ffn_pc.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DS2T_PC_DBG=${n%%_*} $extra -c $f -o $o; fi
    objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libs2t_hip.so $objs
  echo built $d/libs2t_hip.so
done
