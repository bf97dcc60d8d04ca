#!/bin/bash
# usage (GPU box): tools/wg_nt_ab.sh "<nt>:<stagger> ..." — weight-gradient kernel duration in the replayed step per policy
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in ${1:-0:0 1:0}; do
  export S2T_WG_NT=${c%%:*} S2T_WG_STAG=${c##*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wgnt -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/wgnt.json 2> gpurun_out/wgnt.err || exit 1
  python3 - $(ls gpurun_out/wgnt/*/*kernel_stats.csv | head -1) $c <<PY
import csv,sys,json
r=[x for x in csv.DictReader(open(sys.argv[1])) if x["Name"].startswith("(anonymous namespace)::wgrad256_kernel")][0]
print("nt:stagger %s  %.3f ms/step  wgrad256 avg %.1f us min %.1f (%s calls)"%(sys.argv[2],json.load(open("gpurun_out/wgnt.json"))["ms_per_step"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,r["Calls"]))
PY
  rm -rf gpurun_out/wgnt
done
