timeout -k 10 600 python3 -m pytest tests/test_attn_fused_gpu.py -q -k "glue" > gpurun_out/r4o.log 2>&1; echo "rc=$?" >> gpurun_out/r4o.log; tail -15 gpurun_out/r4o.log
