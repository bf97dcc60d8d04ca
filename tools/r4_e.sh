bash tools/step_trace.sh > gpurun_out/r4e_trace.log 2>&1 && cp gpurun_out/step_trace.txt gpurun_out/r4e_step_trace_packed.txt
tail -3 gpurun_out/r4e_trace.log
