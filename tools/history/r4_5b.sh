#!/bin/bash
# kernel statistics of configuration 5b's greedy pass (the d = 512 NAST stack)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_5b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 tools/run_configs.py 5bg > $O/log.txt 2>&1
tail -5 $O/log.txt
