#!/bin/bash
# Build an experiment variant of libs2t_hip.so: one source recompiled with extra defines, the other objects reused.
# usage: tools/dbg_variant.sh <name> <source.hip> <hipcc flags...>   ->  s2t_amd/lib/var_<name>/libs2t_hip.so  (S2T_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
d=s2t_amd/lib/var_$name; mkdir -p $d; objs=""
for f in s2t_amd/csrc/*.hip; do
  o=s2t_amd/lib/obj/$(basename ${f%.hip}).o
  if [ "$(basename $f)" = "$src" ]; then o=$d/${src%.hip}.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude "$@" -c $f -o $o; fi
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libs2t_hip.so $objs
echo built $d/libs2t_hip.so
