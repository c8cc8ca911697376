#!/bin/bash
# Tile / split sweep of the dense products at the row counts of the 2 / 4 / 8-GPU operating points (M = 9216 / 4608 / 2304):
# per-dispatch durations from the rocprofv3 kernel trace.   usage: tools/gemm_mid_rows.sh <out file>
out=$1
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
SHAPES="nt,2304,1024,1024 ntp,2304,1024,1024 nn,2304,1024,1024 nnp,2304,1024,1024 tn,1024,1024,2304 tnp,1024,1024,2304 nt,2304,1024,4096 nn,2304,4096,1024 tn,1024,4096,2304 nt,4608,1024,1024 ntp,4608,1024,1024 nn,4608,1024,1024 tn,1024,1024,4608 nt,9216,1024,1024 ntp,9216,1024,1024 nn,9216,1024,1024 tn,1024,1024,9216"
: > $out
run() {
  d=$(mktemp -d /tmp/gt.XXXX)
  echo "=== $1" >> $out
  env $1 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 tools/gemm_trace.py $SHAPES > /dev/null 2>&1
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python3 tools/gemm_trace_read.py $f >> $out 2>&1
  rm -rf $d
}
run "EKAID_X=0"
run "EKAID_GEMM_SMALL_UNSPLIT=0"
run "EKAID_GEMM_SMALL_UNSPLIT=0 EKAID_GEMM_TILE=3"
run "EKAID_GEMM_SMALL_UNSPLIT=0 EKAID_GEMM_TILE=4"
run "EKAID_GEMM_SMALL_UNSPLIT=4096"
