#!/bin/bash
# usage (GPU box): tools/step_trace.sh — every launch of one replayed step in time order -> gpurun_out/step_trace.txt
# (start offset us, duration us, gap to the previous launch's end us, kernel, grid, workgroup)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/steptrace
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 3 --rotate 1 > gpurun_out/steptrace_bench.log 2>&1 || exit 1
python3 - $(ls $out/*/*kernel_trace.csv | head -1) <<PY
import csv,sys
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:90],r.get("Grid_Size_X",""),r.get("Workgroup_Size_X","")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
ad=[i for i,r in enumerate(rows) if r[2].startswith("adam_kernel")]
spans=[(rows[ad[k+1]][1]-rows[ad[k]][1],k) for k in range(len(ad)-1)]
print("adam launches %d; step spans ms: %s"%(len(ad)," ".join("%.2f"%(s_/1e6) for s_,_ in spans)))
k=min(spans)[1]  # a replayed step (the eager warm-up steps are several times longer)
i0,i1=ad[k]+1,ad[k+1]+1
seg=rows[i0:i1]
t0=seg[0][0]; be=t0; out=[]
for s,e,n,g,w in seg:
    out.append("%9.1f %7.1f %6.1f  %-90s %s/%s"%((s-t0)/1e3,(e-s)/1e3,(s-be)/1e3,n,g,w))
    be=max(be,e)
out.append("span %.1f us, %d launches"%((be-t0)/1e3,len(seg)))
open("gpurun_out/step_trace.txt","w").write("\n".join(out)+"\n")
print(out[-1])
PY
rm -rf $out
