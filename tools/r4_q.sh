timeout -k 10 800 python3 -m pytest tests/test_packed_rows_gpu.py -q -k "edge or long" > gpurun_out/r4q.log 2>&1; echo "rc=$?" >> gpurun_out/r4q.log; tail -30 gpurun_out/r4q.log
