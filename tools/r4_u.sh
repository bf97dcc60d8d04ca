timeout -k 10 900 python3 -m pytest tests/test_packed_rows_gpu.py -q -k "beam" > gpurun_out/r4u.log 2>&1; echo "rc=$?" >> gpurun_out/r4u.log; tail -30 gpurun_out/r4u.log
