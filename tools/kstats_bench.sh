#!/bin/bash
# usage (GPU box): tools/kstats_bench.sh [top-n] [bench args] — per-kernel totals of bench.py under rocprofv3 (one JSON line first)
N=${1:-40}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py "$@" > gpurun_out/kb_plain.log 2> gpurun_out/kb_plain.err || exit 1
cat gpurun_out/kb_plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kb -- python3 bench.py --steps 10 --warmup 3 > gpurun_out/kb.log 2> gpurun_out/kb.err || exit 1
cp $(ls gpurun_out/kb/*/*kernel_stats.csv | head -1) gpurun_out/kb_kernel_stats.csv; rm -rf gpurun_out/kb
python3 - $N <<PY
import csv,sys
rows=list(csv.DictReader(open("gpurun_out/kb_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms"%(tot/1e6))
for r in rows[:int(sys.argv[1])]:
    print("%5.1f%% %9.1f us avg x%-6s %s"%(float(r["Percentage"]),float(r["AverageNs"])/1e3,r["Calls"],r["Name"].replace("(anonymous namespace)::","")[:110]))
PY
