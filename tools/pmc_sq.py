#!/usr/bin/env python3
"""Per-kernel SQ counters of a profiled training run -> MFMA-pipe / vector / stall figures of the kernels that take the time.

usage: pmc_sq.py <counter_collection.csv> [--json out.json] [--top N]
Counters (one rocprofv3 --pmc pass, /opt/skills/guides/MI355X_MICROARCH.md "SQ" row): SQ_WAVE_CYCLES, SQ_WAIT_ANY (waves parked
on s_waitcnt / barriers), SQ_WAIT_INST_ANY (issue stalls), SQ_ACTIVE_INST_ANY, SQ_ACTIVE_INST_VALU, SQ_VALU_MFMA_BUSY_CYCLES,
SQ_BUSY_CYCLES.  WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles, summed over waves); MFMA busy cycles are
per SIMD and summed over the chip: divided by SQ_BUSY_CYCLES x the SIMDs per SQ-counter instance it is the matrix pipes' duty cycle
while the kernel runs (reported as mfma_busy_frac; mfma_per_wave_frac = busy cycles / (4 x wave quad-cycles) is the share of a
wave's life its SIMD's matrix pipe was busy)."""
import csv, json, sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
if out_json:
    args.remove(out_json)
if "--top" in sys.argv:
    args.remove(str(top))
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(float)
for r in csv.DictReader(open(args[0])):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
    if r.get("Start_Timestamp") and r.get("End_Timestamp") and r["Counter_Name"] == "SQ_WAVE_CYCLES":
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = []
for k, d in acc.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    n = max(cnt[k].values())
    row = {"kernel": k, "launches": n, "wave_quad_cycles_per_launch": wc / n,
           "parked_frac": d.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stall_frac": d.get("SQ_WAIT_INST_ANY", 0.0) / wc,
           "active_frac": d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "valu_active_frac": d.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
           "mfma_busy_cycles_per_launch": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n,
           "mfma_per_wave_frac": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * wc),
           "sq_busy_cycles_per_launch": d.get("SQ_BUSY_CYCLES", 0.0) / n}
    if dur[k] > 0:
        row["avg_duration_us_under_pmc"] = dur[k] / n / 1e3
        # 1024 SIMDs (256 CUs x 4): the matrix pipes' duty cycle over the launch, at the clock the launch ran at (unknown here:
        # 2.1 GHz nominal under load)
        row["mfma_busy_frac_at_2p1GHz"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (dur[k] * 2.1 * 1024)
    rows.append(row)
rows.sort(key=lambda r: -r["wave_quad_cycles_per_launch"] * r["launches"])
for r in rows[:top]:
    print("%-78s x%4d parked %4.1f%% stall %4.1f%% active %4.1f%% (valu %4.1f%%) mfma/wave %4.1f%%%s" % (
        r["kernel"].replace("(anonymous namespace)::", "")[:78], r["launches"], 100 * r["parked_frac"], 100 * r["issue_stall_frac"],
        100 * r["active_frac"], 100 * r["valu_active_frac"], 100 * r["mfma_per_wave_frac"],
        ("  mfma duty %4.1f%%  %.1f us" % (100 * r["mfma_busy_frac_at_2p1GHz"], r["avg_duration_us_under_pmc"])) if "mfma_busy_frac_at_2p1GHz" in r else ""))
if out_json:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_hash
    json.dump({"unit": "fractions of SQ_WAVE_CYCLES unless named otherwise; see tools/pmc_sq.py", "csrc_sha256": csrc_hash(),
               "kernels": rows[:max(top, 24)]}, open(out_json, "w"), indent=1)
