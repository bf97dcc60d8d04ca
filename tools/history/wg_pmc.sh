#!/bin/bash
# usage (GPU box): tools/wg_pmc.sh "<nt>:<stagger> ..." — bytes the weight-gradient kernel fetches past L2 per launch (FETCH_SIZE x2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in ${1:-1:0}; do
  export S2T_WG_NT=${c%%:*} S2T_WG_STAG=${c##*:}
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/wgpmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/wgpmc.log 2>&1 || exit 1
  python3 - $(ls gpurun_out/wgpmc/*/*counter_collection.csv | head -1) $c <<PY
import csv,sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"]=="FETCH_SIZE" and "wgrad256_kernel" in r["Kernel_Name"]]
print("nt:stagger %s  wgrad256 fetch %.2f GB per launch (%d launches, min %.2f max %.2f)"%(sys.argv[2],2*sum(v)/len(v)*1024/1e9,len(v),2*min(v)*1024/1e9,2*max(v)*1024/1e9))
PY
  rm -rf gpurun_out/wgpmc
done
