#!/bin/bash
# usage: tools/load_chains.sh <source.hip> [kernel-name-substring] [hipcc flags...] — per kernel: how many times the instruction
# stream REQUESTS vector-memory loads and then WAITS for them (s_waitcnt vmcnt) before requesting more: each such group is a
# memory round trip the wave sits through.  Loops count once.  Prologues that read per-row operands pass by pass, or a seed /
# mask entry / statistics value where it is first used, show up as long chains.  Runs on the CPU (hipcc -S).
src=$1; pat=${2:-.}; shift 2
out=/tmp/chains_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I$(dirname $0)/../include "$@" s2t_amd/csrc/$src -o $out 2>/dev/null || exit 1
awk -v pat="$pat" '
  /^_Z.*:/ { if (k != "" && k ~ pat) printf "%4d round trips, %4d loads  %s\n", rt[k], nl[k], k; k=$1; pend=0 }
  /(global|buffer|scratch)_load/ && !/ lds/ { pend++; nl[k]++ }
  /s_waitcnt.*vmcnt/ { if (pend > 0) { rt[k]++; pend=0 } }
  END { if (k ~ pat) printf "%4d round trips, %4d loads  %s\n", rt[k], nl[k], k }' $out | c++filt | sed 's/(anonymous namespace):://g' | sort -rn
rm -f $out
