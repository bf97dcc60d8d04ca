#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE counter_collection.csv files).

FETCH_SIZE / WRITE_SIZE are in KiB... (rocprofv3: kilobytes at the L2's memory-side interface).  gfx950 correction
(/opt/skills/guides/MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming
reads -> doubled here; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.
usage: pmc_traffic.py <fetch.csv> <write.csv> [name-substring ...] [--json out.json]"""
import csv, json, sys
from collections import defaultdict

def load(path, counter):
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if out_json: args.remove(out_json)
fetch, write = load(args[0], "FETCH_SIZE"), load(args[1], "WRITE_SIZE")
subs = args[2:]
rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[0] * 2 + write.get(k, [0, 0])[0])):
    if subs and not any(s in k for s in subs):
        continue
    f, nf = fetch.get(k, [0.0, 1]); w, nw = write.get(k, [0.0, 1])
    rd = 2.0 * f * 1024 / max(nf, 1); wr = w * 1024 / max(nw, 1)
    rows.append({"kernel": k, "launches": nf, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                 "hbm_bytes_per_launch": rd + wr})
for r in rows[:40]:
    print("%9.2f MB rd %9.2f MB wr  x%5d  %s" % (r["read_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6,
                                                  r["launches"], r["kernel"][:110]))
if out_json:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_hash  # the table is only valid for the kernel sources it was collected from
    # per-update total: the profiled command runs full training updates only (bench.py --no-roofline), one adam_kernel each
    updates = max([r["launches"] for r in rows if "adam_kernel" in r["kernel"]] or [0])
    total = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows)
    json.dump({"unit": "bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)", "csrc_sha256": csrc_hash(),
               "updates_profiled": updates, "hbm_bytes_per_step": (total / updates) if updates else None, "kernels": rows},
              open(out_json, "w"), indent=1)
