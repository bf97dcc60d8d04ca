#!/bin/bash
# usage (on the GPU box): tools/kstats2.sh <outdir> <npasses> <script.py> [args...]
# rocprofv3 --kernel-trace --stats of `python <script.py> <args>`; prints the per-pass kernel table.
out=$1; n=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out -- python "$@" > gpurun_out/${out}_run.log 2>&1
f=$(ls gpurun_out/$out/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/${out}_kernel_stats.csv
python - $f $n <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); n=int(sys.argv[2])
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/pass %.3f   launches/pass %.0f" % (tot/n/1e6, sum(int(r["Calls"]) for r in rows)/n))
for r in rows[:40]:
    nm=r["Name"].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","")[:90]
    print("%6.2f%% %7.3f ms/pass  calls/pass %5.1f  avg %8.1f us  %s"%(float(r["Percentage"]), int(r["TotalDurationNs"])/n/1e6, int(r["Calls"])/n, float(r["AverageNs"])/1e3, nm))
PY
