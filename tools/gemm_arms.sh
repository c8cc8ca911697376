#!/bin/bash
# 128x128-tile GEMM arms on the dominant shapes, interleaved rounds of separate processes on one box (the arm is read once per
# process from EKAID_GEMM_BIG): pipe = shipped software-pipelined 32x32x16, plain = 32-deep loop 32x32x16, m16 = 16x16x32.
# usage: tools/gemm_arms.sh <out file>
out=$1
: > $out
for round in 1 2 3; do
  for arm in pipe plain m16; do
    echo "== round $round arm $arm" >> $out
    EKAID_GEMM_BIG=$arm GEMM_BENCH_SET=big GEMM_BENCH_CHECK=1 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids >> $out
  done
done
EKAID_GEMM_BIG=m16 python -m pytest tests/test_hip_ops.py -q -m gpu -k "gemm" 2>&1 | tail -3 >> $out
