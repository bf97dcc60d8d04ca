timeout -k 10 600 python3 -m pytest tests/test_packed_rows_gpu.py tests/test_fullsize_properties_gpu.py -q > gpurun_out/r4j.log 2>&1; echo "rc=$?" >> gpurun_out/r4j.log; tail -8 gpurun_out/r4j.log
