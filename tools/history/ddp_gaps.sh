#!/bin/bash
# usage (GPU box): tools/ddp_gaps.sh [0|1] — idle gaps of one replayed step of the single-rank data-parallel rehearsal (kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export S2T_FORCE_DDP=${1:-1}
out=gpurun_out/ddpgap
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 3 > gpurun_out/ddpgap_bench.log 2>&1
python3 - $(ls $out/*/*kernel_trace.csv | head -1) $(ls $out/*/*memory_copy_trace.csv | head -1) <<PY
import csv,sys
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:70]) for r in csv.DictReader(open(sys.argv[1]))]
try:
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")) for r in csv.DictReader(open(sys.argv[2]))]
except Exception as e: print("no copies",e)
rows.sort()
# steps end with adam_kernel: take the interval between the 3rd-last and 2nd-last adam launches (a timed replay)
ad=[i for i,r in enumerate(rows) if r[2].startswith("adam_kernel")]
i0,i1=ad[-5]+1,ad[-4]+1
seg=rows[i0:i1]
t0=seg[0][0]; busy_end=seg[0][1]; gaps=[]
for k in range(1,len(seg)):
    s,e,n=seg[k]
    if s>busy_end: gaps.append((s-busy_end, seg[k-1][2], n, (s-t0)/1e3))
    busy_end=max(busy_end,e)
print("step span %.1f us, %d launches, idle %.1f us in %d gaps"%((busy_end-t0)/1e3,len(seg),sum(g[0] for g in gaps)/1e3,len(gaps)))
for g in sorted(gaps,reverse=True)[:12]:
    print("%7.1f us at %8.1f  after %-50s before %s"%(g[0]/1e3,g[3],g[1][:50],g[2][:50]))
big=max(gaps)
print("--- launches around the largest gap")
for s_,e_,n_ in seg:
    if abs((s_-t0)/1e3-big[3])<250: print("%9.1f %7.1f  %s"%((s_-t0)/1e3,(e_-s_)/1e3,n_[:64]))
PY
rm -rf $out
