timeout -k 10 800 python3 -m pytest tests/test_trainer_gpu.py -q > gpurun_out/r4p.log 2>&1; echo "rc=$?" >> gpurun_out/r4p.log; tail -15 gpurun_out/r4p.log
