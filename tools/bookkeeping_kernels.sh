#!/bin/bash
# usage (GPU box): tools/bookkeeping_kernels.sh — what the per-batch bookkeeping outside the captured step launches: kernel
# call counts of the default bench with 4 rotated batches minus the same run with one batch (no load_batch), per timed step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bk$r -- python3 bench.py --no-cpu-baseline --no-roofline --blocks 1 --steps 40 --warmup 4 --rotate $r > gpurun_out/bk${r}_bench.json 2> gpurun_out/bk$r.err || exit 1
  cp $(ls gpurun_out/bk$r/*/*kernel_stats.csv | head -1) gpurun_out/bk${r}_kernel_stats.csv; rm -rf gpurun_out/bk$r
done
python3 - <<PY
import csv
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
a, b = load("gpurun_out/bk1_kernel_stats.csv"), load("gpurun_out/bk4_kernel_stats.csv")
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    if cb != ca:
        rows.append(((tb - ta) / 43.0 / 1e3, (cb - ca) / 43.0, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows); n = sum(r[1] for r in rows)
print("per step: %.1f us in %.1f launches" % (tot, n))
for t, c, k in rows:
    print("%7.1f us  %5.1f x  %s" % (t, c, k.replace("(anonymous namespace)::", "")[:150]))
PY
