#!/bin/bash
# GAT forward kernel on one box: round-2 tree (_base/), EKAID_GAT_V=1 (edge terms in LDS, no chunk prefetch), default (both)
out=$1
: > $out
for round in 1 2; do
  for shape in "512 36 36 1024 4" "512 52 52 1024 4" "128 52 52 1024 4" "64 36 36 1024 4"; do
    echo "== round $round round-2 kernel, shape $shape" >> $out
    (cd _base && python tools/gat_bench.py $shape 2>&1 | grep -E "^forward" >> ../$out)
    echo "== round $round EKAID_GAT_V=1 shape $shape" >> $out
    EKAID_GAT_V=1 python tools/gat_bench.py $shape 2>&1 | grep -E "^forward" >> $out
    echo "== round $round default shape $shape" >> $out
    python tools/gat_bench.py $shape 2>&1 | grep -E "^forward" >> $out
  done
done
python -m pytest tests/test_hip_ops.py -q -m gpu -k "gat" 2>&1 | tail -3 >> $out
