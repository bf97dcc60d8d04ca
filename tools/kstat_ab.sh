#!/bin/bash
# usage (GPU box): tools/kstat_ab.sh <variant> <name-substring>... — kernel averages of the default bench with the shipped
# library and with s2t_amd/lib/var_<variant>/libs2t_hip.so, one box
v=$1; shift
tools/kstat_one.sh "$@"
export S2T_HIP_LIB=$PWD/s2t_amd/lib/var_$v/libs2t_hip.so
echo "--- variant $v"
tools/kstat_one.sh "$@"
