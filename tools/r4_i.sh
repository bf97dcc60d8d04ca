timeout -k 10 900 python3 tools/run_configs.py 1 2 2p 3 4 5a > gpurun_out/r4i_cfg.log 2>&1; echo "rc=$?" >> gpurun_out/r4i_cfg.log
S2T_PACKED=0 timeout -k 10 600 python3 tools/run_configs.py 2 4 5a > gpurun_out/r4i_cfg_padded.log 2>&1; echo "rc=$?" >> gpurun_out/r4i_cfg_padded.log
grep -v "amdgpu.ids" gpurun_out/r4i_cfg.log | tail -12; grep -v "amdgpu.ids" gpurun_out/r4i_cfg_padded.log | tail -6
