timeout -k 10 600 python3 -m pytest tests/test_packed_rows_gpu.py -q > gpurun_out/r4d_packed.log 2>&1; echo "rc=$?" >> gpurun_out/r4d_packed.log; tail -3 gpurun_out/r4d_packed.log
S2T_PACKED=0 timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4d_padded.json 2> gpurun_out/r4d_padded.err || tail -20 gpurun_out/r4d_padded.err
timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4d_packed.json 2> gpurun_out/r4d_packed.err || tail -20 gpurun_out/r4d_packed.err
S2T_PACKED=0 timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4d_padded2.json 2> gpurun_out/r4d_padded2.err
timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4d_packed2.json 2> gpurun_out/r4d_packed2.err
python3 - <<'PY'
import json
for n in ("padded","packed","padded2","packed2"):
    try:
        d=json.load(open("gpurun_out/r4d_%s.json"%n)); print(n, d["ms_per_step"], d["value"], d["roofline"].get("encoder_fwd"))
    except Exception as e: print(n, "ERR", e)
PY
