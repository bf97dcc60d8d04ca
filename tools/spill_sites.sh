#!/bin/bash
# usage: tools/spill_sites.sh <source.hip> [kernel-name-substring] [hipcc flags...] — scratch (spill) instructions of each
# kernel by basic block, loop headers marked: a spill inside a chunk loop that counts its own s_waitcnt vmcnt is a stall
# (the compiler's wait for the scratch load also waits for the LDS-DMA stream).  Runs on the CPU (hipcc -S).
src=$1; pat=${2:-.}; shift 2
out=/tmp/spill_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I$(dirname $0)/../include "$@" s2t_amd/csrc/$src -o $out 2>/dev/null || exit 1
awk -v pat="$pat" '
  /^_Z.*:/ { k=$1; inl=0 }
  /^\.LBB/ { cur=$1; inl = ($0 ~ /Loop Header|=>This/) ? 1 : 0; if (inl) loops[k SUBSEP cur]=1 }
  /scratch_(load|store)/ { if (k ~ pat) n[k SUBSEP cur]++ }
  END { for (x in n) { split(x, a, SUBSEP); printf "%s %s %d%s\n", a[1], a[2], n[x], ((x in loops) ? "  <-- LOOP" : "") } }' $out | sort | c++filt | sed 's/(anonymous namespace):://g'
rm -f $out
