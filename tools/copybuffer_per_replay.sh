#!/bin/bash
# usage (GPU box): tools/copybuffer_per_replay.sh [outdir]
# How many `__amd_rocclr_copyBuffer` launches (and ATen kernels) ONE captured replay of the headline step costs: two kernel traces of
# bench.py without its instrumented legs, 10 and 40 timed steps; the difference divided by 30 is the per-replay count (capture,
# warm-up and model construction cancel).  tools/copy_sites.py says which Python lines they are.
out=${1:-gpurun_out/copybuf}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 10 40; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/k$n -- python3 bench.py --steps $n --warmup 2 --blocks 1 --no-cpu-baseline --no-roofline > $out/b$n.json 2> $out/b$n.err || exit 1
  cp $(ls $out/k$n/*/*kernel_stats.csv | head -1) $out/stats$n.csv
done
python3 - $out <<'PY'
import csv, sys
out = sys.argv[1]
def load(n):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f"{out}/stats{n}.csv"))}
a, b = load(10), load(40)
rows = []
for k in b:
    dc = (b[k][0] - a.get(k, (0, 0))[0]) / 30.0
    dt = (b[k][1] - a.get(k, (0, 0.0))[1]) / 30.0 / 1e3
    rows.append((dt, dc, k))
tot = sum(r[0] for r in rows)
print("per replay: %.1f launches, %.1f us of kernel time" % (sum(r[1] for r in rows), tot))
sel = [r for r in rows if "copyBuffer" in r[2] or r[2].startswith("void at::") or "at::native" in r[2] or "fillBuffer" in r[2]]
print("of which copyBuffer / fillBuffer / ATen kernels: %.1f launches, %.1f us" % (sum(r[1] for r in sel), sum(r[0] for r in sel)))
for dt, dc, k in sorted(sel, reverse=True):
    if dc > 0:
        print("  %6.2f/replay %8.2f us  %s" % (dc, dt, k[:150]))
PY
rm -rf $out/k10 $out/k40
