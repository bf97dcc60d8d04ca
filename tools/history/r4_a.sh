set -x
python3 bench.py --no-cpu-baseline > gpurun_out/r4a_b64.json 2> gpurun_out/r4a_b64.err &&
python3 bench.py --no-cpu-baseline --batch 52 > gpurun_out/r4a_b52.json 2> gpurun_out/r4a_b52.err &&
python3 bench.py --no-cpu-baseline --batch 54 > gpurun_out/r4a_b54.json 2> gpurun_out/r4a_b54.err &&
python3 bench.py --no-cpu-baseline --batch 56 > gpurun_out/r4a_b56.json 2> gpurun_out/r4a_b56.err &&
bash tools/step_trace.sh && cp gpurun_out/step_trace.txt gpurun_out/r4a_step_trace_b64.txt
