#!/bin/bash
# kernel statistics of configuration 3 (PDS, 64 x 2000) and 4 (SATE): captured training steps
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in 3 4; do
  O=$GRAFT_REPO_ROOT/gpurun_out/r4_c$c; mkdir -p $O
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 tools/run_configs.py $c > $O/log.txt 2>&1 || exit 1
  grep "ms/step" $O/log.txt
done
