#!/bin/bash
# kernel statistics of configuration 5b's captured training step
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_5bt
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 tools/try_capture_5b.py > $O/log.txt 2>&1
grep "captured\|eager" $O/log.txt
