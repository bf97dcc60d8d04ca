#!/bin/bash
# usage (GPU box): tools/ab_env.sh "<VAR=value ...>" [rounds] — the default bench without and with the given environment, alternated
n=${2:-2}
for i in $(seq $n); do
  python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base  %.3f' % d['ms_per_step'])"
  env $1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with  %.3f   ($1)' % d['ms_per_step'])"
done
