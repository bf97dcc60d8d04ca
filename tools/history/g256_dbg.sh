#!/bin/bash
# Experiment variants of gemm256.hip (-DS2T_G256_DBG=<n> [-DS2T_G256_VAR=<v>]) into s2t_amd/lib/g256_<tag>/ (use with S2T_HIP_LIB=...).
# usage: tools/g256_dbg.sh tag "-DS2T_G256_DBG=1" [tag2 "flags2" ...]
set -e
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  d=s2t_amd/lib/g256_$tag; mkdir -p $d
  objs=""
  for f in s2t_amd/csrc/*.hip; do
    o=s2t_amd/lib/obj/$(basename ${f%.hip}).o
    if [ "$(basename $f)" = gemm256.hip ]; then o=$d/gemm256.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $f -o $o; fi
    objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libs2t_hip.so $objs
  echo built $d/libs2t_hip.so
done
