#!/bin/bash
# Build experiment variants of libs2t_hip.so with -DS2T_DBG_EPI=<n> into s2t_amd/lib/dbg<n>/ (use with S2T_HIP_LIB=...).
set -e
cd "$(dirname "$0")/.."
for n in "$@"; do
  d=s2t_amd/lib/dbg$n; mkdir -p $d
  for f in s2t_amd/csrc/*.hip; do
    o=s2t_amd/lib/obj/$(basename ${f%.hip}).o
    if [ "$(basename $f)" = gemm.hip ]; then o=$d/gemm.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DS2T_DBG_EPI=$n -c $f -o $o; fi
    objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libs2t_hip.so $objs; objs=""
  echo built $d/libs2t_hip.so
done
