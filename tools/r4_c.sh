timeout -k 10 600 python3 tools/packed_dbg2.py > gpurun_out/r4c_dbg.log 2>&1
tail -20 gpurun_out/r4c_dbg.log
