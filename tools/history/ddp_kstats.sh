#!/bin/bash
# usage (GPU box): tools/ddp_kstats.sh  — kernel table of the single-rank data-parallel rehearsal (S2T_FORCE_DDP=1) next to the plain step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in 0 1; do
  export S2T_FORCE_DDP=$mode
  out=gpurun_out/ddpk$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline > gpurun_out/ddpk${mode}_bench.log 2>&1
  cp $(ls $out/*/*kernel_stats.csv | head -1) gpurun_out/ddpk${mode}_stats.csv
  rm -rf $out
done
python3 - <<PY
import csv
def load(f):
    return {r["Name"]:(int(r["Calls"]),int(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a=load("gpurun_out/ddpk0_stats.csv"); b=load("gpurun_out/ddpk1_stats.csv")
rows=[]
for k in set(a)|set(b):
    ca,ta=a.get(k,(0,0)); cb,tb=b.get(k,(0,0))
    rows.append(((tb-ta)/30/1e3,k,ca,cb,ta,tb))
rows.sort(reverse=True)
print("total kernel us/step: plain %.0f  ddp %.0f"%(sum(v[1] for v in a.values())/30/1e3,sum(v[1] for v in b.values())/30/1e3))
for d,k,ca,cb,ta,tb in rows[:14]+rows[-6:]:
    print("%+8.1f us/step  calls %5d -> %5d   %s"%(d,ca,cb,k.replace("(anonymous namespace)::","")[:90]))
PY
