#!/bin/bash
# Round-end measurement campaign (run on the GPU box from the repo root); everything lands under gpurun_out/r03_*.
o=gpurun_out
tools/stats_step.sh $o/r03_bench_b256 --steps 20
tools/stats_step.sh $o/r03_bench_b32 --steps 20 --batch 32 --hip-graph on
tools/stats_step.sh $o/r03_real_b64 --real-only
tools/pmc_gemm2.sh $o/r03_pmc_nn nn 18432 1024 1024 > /dev/null 2>&1
python tools/pmc_gemm_summary.py $o/r03_pmc_nn/summary.json nn 18432 1024 1024 $o/r03_pmc_gemm_split_nn_18432x1024x1024.json > /dev/null
tools/pmc_step.sh $o/r03_pmc_step > /dev/null 2>&1
python tools/pmc_step_read.py $o/r03_pmc_step $o/r03_pmc_step_b256.json 27 > /dev/null
for shape in "512 36 36 1024 4" "512 52 52 1024 4" "128 52 52 1024 4" "64 36 36 1024 4"; do
  echo "== this tree, shape $shape"; python tools/gat_bench.py $shape 2>&1 | grep -E "^forward|^backward"
  echo "== round-2 tree, shape $shape"; (cd _base && python tools/gat_bench.py $shape 2>&1 | grep -E "^forward")
done > $o/r03_gat_microbench.txt
tools/ab.sh $o/r03_ab256 --steps 40 --warmup 5 > $o/r03_ab_vs_round2.txt 2>&1
tools/ab.sh $o/r03_ab32 --steps 40 --warmup 5 --batch 32 --hip-graph on >> $o/r03_ab_vs_round2.txt 2>&1
python bench.py > $o/r03_bench_default_line.json 2> $o/r03_bench_default.err
rm -rf $o/r03_pmc_step/g* $o/r03_pmc_step/trace
ls $o | grep r03_
