timeout -k 10 900 python3 -m pytest tests/test_packed_kernels_gpu.py -q > gpurun_out/r4w.log 2>&1; echo "rc=$?" >> gpurun_out/r4w.log; tail -40 gpurun_out/r4w.log
