timeout -k 10 1100 python3 -m pytest tests -q -m gpu > gpurun_out/r4s_all.log 2>&1; echo "rc=$?" >> gpurun_out/r4s_all.log; tail -5 gpurun_out/r4s_all.log
timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4s_bench.json 2> gpurun_out/r4s.err || tail -20 gpurun_out/r4s.err
S2T_PACKED=0 timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4s_bench0.json 2> gpurun_out/r4s0.err
timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4s_bench2.json 2> gpurun_out/r4s2.err
python3 - <<'PY'
import json
for n in ("bench","bench0","bench2"):
    d=json.load(open("gpurun_out/r4s_%s.json"%n)); print(n, round(d["ms_per_step"],3), int(d["value"]), d["config"]["timed_blocks_ms_per_step"])
PY
