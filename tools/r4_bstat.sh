#!/bin/bash
# kernel statistics of the default bench command (no roofline leg)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_bstat
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --no-roofline > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-200
