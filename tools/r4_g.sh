timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > gpurun_out/r4g_all.log 2>&1; echo "rc=$?" >> gpurun_out/r4g_all.log; tail -6 gpurun_out/r4g_all.log
timeout -k 10 600 python3 bench.py > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err || tail -20 gpurun_out/r4g_bench.err
cat gpurun_out/r4g_bench.json | head -c 6000
