#!/bin/bash
# Round-end measurement campaign (run on the GPU box from the repo root); everything lands under gpurun_out/r06_*; the summaries that are
# cited in DESIGN.md are copied to profiles/ by hand afterwards.   usage: tools/final_profiles.sh [quick]
# (Run it in pieces under gpurun: one rocprofv3 --pmc pass that aborts can sit in its signal handler until the outer limit -- every pass runs under
# `timeout -k 10`, and the campaign as a whole cost 40 GPU-minutes the one time a pass did that without -k.)
o=gpurun_out
tools/stats_step.sh $o/r06_bench_b256 --steps 20 --no-ladder
tools/stats_step.sh $o/r06_bench_b32 --steps 20 --batch 32 --hip-graph on
tools/stats_step.sh $o/r06_bench_b64 --steps 20 --batch 64 --hip-graph on
tools/stats_step.sh $o/r06_bench_b128 --steps 20 --batch 128 --hip-graph on
tools/stats_step.sh $o/r06_real_b64 --real-only
tools/pmc_gemm2.sh $o/r06_pmc_nn nn 18432 1024 1024 > /dev/null 2>&1
python tools/pmc_gemm_summary.py $o/r06_pmc_nn/summary.json nn 18432 1024 1024 $o/r06_pmc_gemm_split_nn_18432x1024x1024.json 3 > /dev/null
tools/pmc_gat_all.sh $o/r06_pmc_gat_n36 512 36 36 1024 4 > /dev/null 2>&1; cp $o/r06_pmc_gat_n36/summary.json $o/r06_pmc_gat_n36.json
tools/pmc_gat_all.sh $o/r06_pmc_gat_n52 128 52 52 1024 4 > /dev/null 2>&1; cp $o/r06_pmc_gat_n52/summary.json $o/r06_pmc_gat_n52.json
python tools/f16x2_bench.py > $o/r06_f16x2_bench.txt 2>&1
python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 5 --gemm-shapes > /dev/null 2> $o/r06_gemm_shapes_b256.txt
python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 5 --batch 32 --gemm-shapes > /dev/null 2> $o/r06_gemm_shapes_b32.txt
python bench.py > $o/r06_bench_default_line.json 2> $o/r06_bench_default.err
rm -rf $o/r06_pmc_nn/g* $o/r06_pmc_gat_n36/g* $o/r06_pmc_gat_n52/g*
ls $o | grep r06_
