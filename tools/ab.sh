#!/bin/bash
# usage (GPU box): tools/ab.sh VAR val_a val_b [rounds]  — bench.py ms_per_step with VAR=val_a / VAR=val_b, alternating, same box
var=$1; a=$2; b=$3; n=${4:-3}
for i in $(seq $n); do
  for v in $a $b; do
    ms=$(env $var=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$var=$v  $ms ms"
  done
done
