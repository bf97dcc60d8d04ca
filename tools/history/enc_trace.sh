cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/enctrace -- python3 tools/enc_fwd_profile.py 3 > gpurun_out/enctrace.log 2>&1
f=$(ls gpurun_out/enctrace/*/*kernel_trace.csv | head -1)
python3 - $f <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)
out=[]
for r in rows[-260:]:
    nm=r["Kernel_Name"].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","")[:60]
    out.append("%10d %8.1f %s grid=%s wg=%s"%(int(r["Start_Timestamp"])-int(rows[-260]["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, nm, r.get("Grid_Size_X",""), r.get("Workgroup_Size_X","")))
open("gpurun_out/enctrace_tail.txt","w").write("\n".join(out))
PY
ls gpurun_out/enctrace/*/ 
