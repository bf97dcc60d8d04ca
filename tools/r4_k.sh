timeout -k 10 600 python3 -m pytest tests/test_packed_rows_gpu.py -q -k pds > gpurun_out/r4k.log 2>&1; echo "rc=$?" >> gpurun_out/r4k.log; tail -30 gpurun_out/r4k.log
timeout -k 10 600 python3 tools/run_configs.py 3 > gpurun_out/r4k_cfg3.log 2>&1; grep "train" gpurun_out/r4k_cfg3.log
