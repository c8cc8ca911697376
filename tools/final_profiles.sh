#!/bin/bash
# Round-end measurement campaign (run on the GPU box from the repo root); everything lands under gpurun_out/r05_*; the summaries that are
# cited in DESIGN.md are copied to profiles/ by hand afterwards.   usage: tools/final_profiles.sh [quick]
o=gpurun_out
tools/stats_step.sh $o/r05_bench_b256 --steps 20 --no-ladder
tools/stats_step.sh $o/r05_bench_b32 --steps 20 --batch 32 --hip-graph on
tools/stats_step.sh $o/r05_bench_b64 --steps 20 --batch 64 --hip-graph on
tools/stats_step.sh $o/r05_bench_b128 --steps 20 --batch 128 --hip-graph on
tools/stats_step.sh $o/r05_real_b64 --real-only
tools/pmc_gemm2.sh $o/r05_pmc_nn nn 18432 1024 1024 > /dev/null 2>&1
python tools/pmc_gemm_summary.py $o/r05_pmc_nn/summary.json nn 18432 1024 1024 $o/r05_pmc_gemm_split_nn_18432x1024x1024.json > /dev/null
python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 5 --gemm-shapes > /dev/null 2> $o/r05_gemm_shapes_b256.txt
python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 5 --batch 32 --gemm-shapes > /dev/null 2> $o/r05_gemm_shapes_b32.txt
python bench.py > $o/r05_bench_default_line.json 2> $o/r05_bench_default.err
rm -rf $o/r05_pmc_nn/g*
ls $o | grep r05_
