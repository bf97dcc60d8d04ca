#!/bin/bash
# usage (GPU box): tools/collect_profiles.sh <tag>    -> gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/)
#   1. the default command under rocprofv3 --kernel-trace --stats           -> <tag>_kernel_stats_default_bench.csv + bench line
#   2. encoder forward only, 10 eager passes                                 -> <tag>_encoder_fwd_kernel_stats.csv
#   3. two --pmc passes (FETCH_SIZE, WRITE_SIZE) of the hipGraph kernel mix  -> <tag>_pmc_traffic.json
#   4. one --pmc pass of SQ counters (MFMA busy, vector active, stalls)      -> <tag>_pmc_sq.json
#   5. tools/gemm256_probe.py + SQ counters of the large-tile GEMM           -> <tag>_gemm256_probe.txt, <tag>_gemm256_pmc_sq.json
#   6. configuration 5b greedy under --kernel-trace --stats                  -> <tag>_config5b_greedy_kernel_stats.csv
#   7. configuration 3 (PDS Conformer, 64 x 2000) training step, same        -> <tag>_config3_kernel_stats.csv
# (A single-rank communicator launches no RCCL kernel — the library short-circuits a one-rank all-reduce — so the overlap of
#  the bucketed all-reduce with backward can only be traced on a multi-GPU node: tools/ddp_overlap.py reads such a trace.)
# Every profiler run puts the program itself after "--" and keeps counters apart from traces.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/k -- python3 bench.py > $out/${tag}_bench.json 2> $out/bench.err || exit 1
cp $(ls $out/k/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_default_bench.csv
echo "step 1 done" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/e -- python3 tools/enc_fwd_profile.py 10 > $out/enc.log 2>&1 || exit 1
cp $(ls $out/e/*/*kernel_stats.csv | head -1) $out/${tag}_encoder_fwd_kernel_stats.csv
echo "step 2 done" >&2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -- python3 bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --no-roofline > $out/pf.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -- python3 bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --no-roofline > $out/pw.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $(ls $out/pf/*/*counter_collection.csv | head -1) $(ls $out/pw/*/*counter_collection.csv | head -1) --json $out/${tag}_pmc_traffic.json > $out/${tag}_pmc_traffic.txt
echo "step 3 done" >&2
# 4. SQ counters of the same kernel mix (MFMA pipe busy / vector active / parked / issue-stalled) -> <tag>_pmc_sq.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/ps -- python3 bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --no-roofline > $out/ps.log 2>&1 || exit 1
python3 tools/pmc_sq.py $(ls $out/ps/*/*counter_collection.csv | head -1) --json $out/${tag}_pmc_sq.json > $out/${tag}_pmc_sq.txt
echo "step 4 done" >&2
# 5. s2t_gemm's two tile paths side by side (equality + interleaved timings), and the large-tile kernel's SQ counters
python3 tools/gemm256_probe.py > $out/${tag}_gemm256_probe.txt 2> $out/g5.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pg -- python3 tools/g256_time.py > $out/pg.log 2>&1 || exit 1
python3 tools/pmc_sq.py $(ls $out/pg/*/*counter_collection.csv | head -1) --json $out/${tag}_gemm256_pmc_sq.json > $out/${tag}_gemm256_pmc_sq.txt
echo "step 5 done" >&2
# 6. configuration 5b (the d = 512 NAST stack), greedy pass: kernel statistics
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -- python3 tools/run_configs.py 5bg > $out/${tag}_config5b_greedy.log 2>&1 || exit 1
cp $(ls $out/c5/*/*kernel_stats.csv | head -1) $out/${tag}_config5b_greedy_kernel_stats.csv
echo "step 6 done" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3 -- python3 tools/run_configs.py 3 > $out/${tag}_config3.log 2>&1 || exit 1
cp $(ls $out/c3/*/*kernel_stats.csv | head -1) $out/${tag}_config3_kernel_stats.csv
echo "step 7 done" >&2
# 8. encoder forward at 4 x the batch (256 x 1000): the same kernels with 3.2 rounds of workgroups (speed-of-light table, tools/sol_table.py)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/e4 -- python3 tools/enc_fwd_profile.py 5 256 > $out/enc256.log 2>&1 || exit 1
cp $(ls $out/e4/*/*kernel_stats.csv | head -1) $out/${tag}_encoder_fwd_b256_kernel_stats.csv
python3 tools/sol_table.py $out/${tag}_encoder_fwd_kernel_stats.csv $(grep -o "([0-9]* frames)" $out/enc.log | tr -dc 0-9) \
    $out/${tag}_encoder_fwd_b256_kernel_stats.csv $(grep -o "([0-9]* frames)" $out/enc256.log | tr -dc 0-9) > $out/${tag}_sol_table.md 2> $out/sol.err
echo "step 8 done" >&2
# 9. launches per captured replay that are not this library's kernels (copyBuffer / fillBuffer / ATen), by differencing two traces
tools/copybuffer_per_replay.sh $out/copybuf > $out/${tag}_aten_per_replay.txt 2>&1 || exit 1
echo "step 9 done" >&2
# 10. the row-panel projection prototype (tools/ubench/rowpanel_proj.hip) beside the shipped kernel, every configuration's timing
timeout -k 10 200 tools/ubench/rowpanel_proj > $out/${tag}_rowpanel_proto.txt 2>&1
python3 tools/rb_proj_times.py >> $out/${tag}_rowpanel_proto.txt 2>&1
python3 tools/run_configs.py 1 2 2p 3 4 5a 5b > $out/${tag}_configs.log 2>&1
echo "step 10 done" >&2
rm -rf $out/k $out/e $out/pf $out/pw $out/ps $out/d $out/pg $out/c5 $out/c3 $out/e4 $out/copybuf
ls -la $out
