#!/bin/bash
# usage (GPU box): tools/kstats_cfg.sh <config> [top-n] — per-kernel totals of tools/run_configs.py <config> under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kc -- python3 tools/run_configs.py $1 > gpurun_out/kc_$1.log 2> gpurun_out/kc_$1.err || exit 1
cp $(ls gpurun_out/kc/*/*kernel_stats.csv | head -1) gpurun_out/kc_$1_kernel_stats.csv; rm -rf gpurun_out/kc
cat gpurun_out/kc_$1.log
python3 - $1 ${2:-40} <<PY
import csv,sys
rows=list(csv.DictReader(open("gpurun_out/kc_%s_kernel_stats.csv"%sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms"%(tot/1e6))
for r in rows[:int(sys.argv[2])]:
    print("%5.1f%% %9.1f us avg x%-6s %s"%(float(r["Percentage"]),float(r["AverageNs"])/1e3,r["Calls"],r["Name"].replace("(anonymous namespace)::","")[:110]))
PY
