#!/bin/bash
# configurations 3, 4, 5b (greedy) and the bench with s2t_gemm's large-tile path off / automatic (same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_cfg; mkdir -p $O
for m in 0 1; do
  S2T_GEMM256=$m timeout -k 10 300 python3 tools/run_configs.py 3 4 5a 5bg > $O/cfg_$m.txt 2>&1 || exit 1
  S2T_GEMM256=$m timeout -k 10 200 python3 bench.py --no-roofline > $O/bench_$m.json 2> $O/bench_$m.err || exit 1
done
grep -h "ms/" $O/cfg_0.txt $O/cfg_1.txt
python3 - <<'P'
import json
for m in (0,1):
    d=json.loads(open('gpurun_out/r4_cfg/bench_%d.json'%m).read().strip().splitlines()[-1])
    print(m, d['ms_per_step'], d['value'])
P
