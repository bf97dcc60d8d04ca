#!/bin/bash
# tools/g256_time.py with experiment builds of the large-tile GEMM (s2t_amd/lib/g256_<tag>/, tools/g256_dbg.sh) on one box; "shipped" = the library as built
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  if [ $t = shipped ]; then timeout -k 10 120 python3 tools/g256_time.py || exit 1
  else S2T_HIP_LIB=$GRAFT_REPO_ROOT/s2t_amd/lib/g256_$t/libs2t_hip.so timeout -k 10 120 python3 tools/g256_time.py || exit 1; fi
done
