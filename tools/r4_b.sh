timeout -k 10 900 python3 -m pytest tests/test_packed_rows_gpu.py -q > gpurun_out/r4b_packed.log 2>&1; echo "rc=$?" >> gpurun_out/r4b_packed.log; tail -5 gpurun_out/r4b_packed.log
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x --deselect tests/test_packed_rows_gpu.py > gpurun_out/r4b_all.log 2>&1; echo "rc=$?" >> gpurun_out/r4b_all.log; tail -15 gpurun_out/r4b_all.log
