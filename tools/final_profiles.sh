#!/bin/bash
# Round-end measurement campaign (run on the GPU box from the repo root); everything lands under gpurun_out/r04_*; the summaries that are
# cited in DESIGN.md are copied to profiles/ by hand afterwards.
o=gpurun_out
tools/stats_step.sh $o/r04_bench_b256 --steps 20
tools/stats_step.sh $o/r04_bench_b32 --steps 20 --batch 32 --hip-graph on
tools/stats_step.sh $o/r04_real_b64 --real-only
tools/pmc_gemm2.sh $o/r04_pmc_nn nn 18432 1024 1024 > /dev/null 2>&1
python tools/pmc_gemm_summary.py $o/r04_pmc_nn/summary.json nn 18432 1024 1024 $o/r04_pmc_gemm_split_nn_18432x1024x1024.json > /dev/null
tools/pmc_step.sh $o/r04_pmc_step > /dev/null 2>&1
python tools/pmc_step_read.py $o/r04_pmc_step $o/r04_pmc_step_b256.json 27 > /dev/null
{
  echo "decoder sequence kernels alone (tools/seq_bench.py): step-wise chain vs persistent kernel, us per step, 40 steps; max relative error over every saved tensor / backward output"
  python tools/seq_bench.py --batch 7 32 64 128 256 --steps 40 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('forward  B=%3d: step-wise %6.1f  persistent %6.1f  status %s  max rel err %.2e' % (d['batch'], d['stepwise_us_per_step'], d['persistent_us_per_step'], d['status'], d['max_rel_err']))"
  python tools/seq_bench.py --batch 7 32 64 128 256 --steps 40 --bwd 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('backward B=%3d: step-wise %6.1f  persistent %6.1f  status %s  max rel err %.2e' % (d['batch'], d['stepwise_us_per_step'], d['persistent_us_per_step'], d['status'], d['max_rel_err']))"
  echo "in-kernel phase stamps (EKAID_PERSIST_TRACE=1), 64 rows: us per span, median over workgroups (max)"
  EKAID_PERSIST_TRACE=1 python tools/seq_bench.py --batch 64 --steps 40 2>&1 | grep -A1 "trace B"
  EKAID_PERSIST_TRACE=1 python tools/seq_bench.py --batch 64 --steps 40 --bwd 2>&1 | grep -A1 "trace B"
  echo "question GRU (tools/gru_bench.py)"
  python tools/gru_bench.py 2>/dev/null | tail -12
  echo "greedy decoding, 64 rows, 90 steps (tools/greedy_time.py)"
  python tools/greedy_time.py 2>/dev/null | grep -v amdgpu
} > $o/r04_seq_microbench.txt 2>&1
tools/ab_round.sh $o/r04_ab_r03_vs_r04.txt
python bench.py > $o/r04_bench_default_line.json 2> $o/r04_bench_default.err
rm -rf $o/r04_pmc_step/g* $o/r04_pmc_step/trace $o/r04_pmc_nn/g* 
ls $o | grep r04_
