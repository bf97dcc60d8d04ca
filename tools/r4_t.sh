timeout -k 10 900 python3 -m pytest tests/test_packed_rows_gpu.py tests/test_attn_fused_gpu.py -q -k "packed or row_map or glue or edge or captured or eval_outputs or training_step or pds or long or 5a" > gpurun_out/r4t.log 2>&1; echo "rc=$?" >> gpurun_out/r4t.log; tail -4 gpurun_out/r4t.log
timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4t_bench.json 2> gpurun_out/r4t.err || tail -20 gpurun_out/r4t.err
timeout -k 10 600 python3 bench.py --no-cpu-baseline --rotate 1 > gpurun_out/r4t_bench_r1.json 2> gpurun_out/r4t1.err
S2T_PACKED=0 timeout -k 10 600 python3 bench.py --no-cpu-baseline --rotate 1 > gpurun_out/r4t_bench0_r1.json 2> gpurun_out/r4t01.err
python3 - <<'PY'
import json
for n in ("bench","bench_r1","bench0_r1"):
    d=json.load(open("gpurun_out/r4t_%s.json"%n)); print(n, round(d["ms_per_step"],3), int(d["value"]), d["config"]["timed_blocks_ms_per_step"])
PY
