#!/bin/bash
# usage (on the GPU box): tools/kstats.sh <outdir> <nsteps_total> -- bench args...
# rocprofv3 --kernel-trace --stats of `python bench.py <args>`; prints the per-step kernel table.
out=$1; n=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out -- python bench.py "$@" > gpurun_out/${out}_bench.log 2>&1
f=$(ls gpurun_out/$out/*/*kernel_stats.csv | head -1)
python - $f $n <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); n=int(sys.argv[2])
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step %.2f" % (tot/n/1e6))
for r in rows[:28]:
    nm=r["Name"].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","")[:84]
    print("%6.2f%% %7.2f ms/step  calls/step %5.0f  avg %8.1f us  %s"%(float(r["Percentage"]), int(r["TotalDurationNs"])/n/1e6, int(r["Calls"])/n, float(r["AverageNs"])/1e3, nm))
PY
