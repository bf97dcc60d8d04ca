timeout -k 10 900 python3 -m pytest tests/test_attn_fused_gpu.py -q -k "packed_rows_equals" > gpurun_out/r4v.log 2>&1; echo "rc=$?" >> gpurun_out/r4v.log; tail -40 gpurun_out/r4v.log
