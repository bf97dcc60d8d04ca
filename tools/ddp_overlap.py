#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `S2T_FORCE_DDP=1 bench.py`: for the last replayed step, every RCCL kernel with its start /
end and the compute kernels that ran while it was in flight — the evidence that the bucketed all-reduce overlaps backward."""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo = adam[-2] + 1 if len(adam) > 1 else 0
step = rows[lo:adam[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
def short(n):
    return re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:60]
cc = [r for r in step if "nccl" in r["Kernel_Name"].lower() or "AllReduce" in r["Kernel_Name"]]
print("step: %d kernels, %.3f ms; RCCL kernels: %d" % (len(step), (int(step[-1]["End_Timestamp"]) - t0) / 1e6, len(cc)))
tot_cc = tot_ov = 0.0
for r in cc:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    over = [q for q in step if q is not r and q not in cc and int(q["Start_Timestamp"]) < e and int(q["End_Timestamp"]) > s]
    ov = sum(min(e, int(q["End_Timestamp"])) - max(s, int(q["Start_Timestamp"])) for q in over)
    tot_cc += e - s
    tot_ov += min(ov, e - s)
    names = sorted({short(q["Kernel_Name"]) for q in over})
    print("  %-44s %9.1f..%9.1f us (%7.1f us)  concurrent compute %7.1f us in %3d kernels: %s" % (
        short(r["Kernel_Name"]), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, ov / 1e3, len(over), ", ".join(names[:4])))
if cc:
    print("RCCL kernel time %.1f us per step, of which %.1f us (%.0f %%) under compute kernels of the same step" % (
        tot_cc / 1e3, tot_ov / 1e3, 100.0 * tot_ov / max(tot_cc, 1)))
