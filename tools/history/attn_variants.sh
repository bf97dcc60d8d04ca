#!/bin/bash
# usage (GPU box): tools/attn_variants.sh v1 v2 ...  -> rocprof kernel averages of tools/attn_probe.py per variant library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=s2t_amd/lib/libs2t_hip.so; [ "$v" != base ] && lib=s2t_amd/lib/var_$v/libs2t_hip.so
  S2T_HIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/av_$v -- python3 tools/attn_probe.py > gpurun_out/av_$v.log 2>&1
  f=$(ls gpurun_out/av_$v/*/*kernel_stats.csv | head -1)
  echo "== $v"; python3 - $f <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn' in r['Name']: print("  %8.1f us x%4s  %s" % (float(r['AverageNs'])/1e3, r['Calls'], r['Name'][:60]))
PY
done
