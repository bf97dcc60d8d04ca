#!/bin/bash
# usage (on the GPU box): tools/pmc_gemm.sh <kind>   -> gpurun_out/pmc_<kind>.txt  (SQ counters of the GEMM kernel, one pass)
kind=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_$kind -- python tools/pmc_gemm.py $kind > gpurun_out/pmc_${kind}_run.log 2>&1
python - $kind <<'PY'
import csv, glob, sys, collections
kind = sys.argv[1]
f = glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % kind)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gemm_kernel" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k[:100])
    wc = d.get("SQ_WAVE_CYCLES", 1)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f  %6.1f%% of wave cycles" % (c, v, 100 * v / wc))
PY
