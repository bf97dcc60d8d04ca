timeout -k 10 600 python3 bench.py > gpurun_out/r4n_bench_plain.json 2> gpurun_out/r4n.err || tail -20 gpurun_out/r4n.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4n_bench_plain.json")); r=d["roofline"]
print(d["ms_per_step"], d["value"], r["kernel"], r["bound"], round(r["frac"],4), round(r["avg_launch_us"],1), r["traffic"], r.get("hbm_bytes_per_step"), r.get("hbm_frac"))
for k,v in r["ffn_flavours"].items(): print(k, round(v["avg_launch_us"],1), v["bound"], round(v["frac"],3), v["traffic"], v["traffic_over_algorithmic"])
print(r["encoder_fwd"]); print(d["cpu_baseline"])
PY
