#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <outdir> <COUNTER> -- bench args...   (one counter set per pass, own run)
out=$1; ctr=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/$out -- python bench.py "$@" > gpurun_out/${out}_bench.log 2>&1
ls gpurun_out/$out/*/ | head
