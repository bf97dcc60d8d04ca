#!/bin/bash
# usage (GPU box): tools/kstat_one.sh <name-substring> ... — average duration of the matching kernels in the default bench run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k1 -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/k1_bench.json 2> gpurun_out/k1.err || exit 1
cp $(ls gpurun_out/k1/*/*kernel_stats.csv | head -1) gpurun_out/k1_kernel_stats.csv; rm -rf gpurun_out/k1
python3 - "$@" <<PY
import csv,sys,json
print("ms/step %.3f"%json.load(open("gpurun_out/k1_bench.json"))["ms_per_step"])
for r in csv.DictReader(open("gpurun_out/k1_kernel_stats.csv")):
    if any(s in r["Name"] for s in sys.argv[1:]):
        print("%8.1f us avg %8.1f min  x%-5s %s"%(float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,r["Calls"],r["Name"].replace("(anonymous namespace)::","")[:90]))
PY
