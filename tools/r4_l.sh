timeout -k 10 1100 python3 -m pytest tests -q -m gpu > gpurun_out/r4l_all.log 2>&1; echo "rc=$?" >> gpurun_out/r4l_all.log; tail -6 gpurun_out/r4l_all.log
