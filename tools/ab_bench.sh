#!/bin/bash
# usage (GPU box): tools/ab_bench.sh <variant-name> [rounds]  — the default bench with the shipped library and with
# s2t_amd/lib/var_<name>/libs2t_hip.so (tools/dbg_variant.sh), alternated on one box
n=${2:-2}
for i in $(seq $n); do
  python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base      %.3f' % d['ms_per_step'])"
  S2T_HIP_LIB=$PWD/s2t_amd/lib/var_$1/libs2t_hip.so python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('var %-6s %.3f' % ('$1', d['ms_per_step']))"
done
