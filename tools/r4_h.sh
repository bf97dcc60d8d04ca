timeout -k 10 900 python3 -m pytest tests -q -m gpu -x -k "fullsize_properties or packed" > gpurun_out/r4h_t.log 2>&1; echo "rc=$?" >> gpurun_out/r4h_t.log; tail -4 gpurun_out/r4h_t.log
S2T_FORCE_DDP=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4h_ddp.json 2> gpurun_out/r4h_ddp.err || tail -20 gpurun_out/r4h_ddp.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4h_ddp.json')); print('ddp1', d['ms_per_step'], d['config']['grad_allreduce'], d['config']['final_loss'])"
timeout -k 10 600 python3 bench.py --no-cpu-baseline --arch transformer > gpurun_out/r4h_tr.json 2> gpurun_out/r4h_tr.err || tail -20 gpurun_out/r4h_tr.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4h_tr.json')); print('transformer', d['ms_per_step'], d['value'])"
S2T_PACKED=0 timeout -k 10 600 python3 bench.py --no-cpu-baseline --arch transformer > gpurun_out/r4h_tr0.json 2> gpurun_out/r4h_tr0.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4h_tr0.json')); print('transformer padded', d['ms_per_step'], d['value'])"
