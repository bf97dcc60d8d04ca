timeout -k 10 900 python3 -m pytest tests/test_packed_rows_gpu.py -q > gpurun_out/r4r.log 2>&1; echo "rc=$?" >> gpurun_out/r4r.log; tail -30 gpurun_out/r4r.log
