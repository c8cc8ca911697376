#!/bin/bash
# rocprofv3 --kernel-trace of `bench.py <args>` -> <out>_kernel_trace.csv (every dispatch with start / end stamps; read it with tools/step_trace_read.py)
# usage: tools/step_trace.sh <out prefix under gpurun_out/> [bench args...]
out=$1; shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
d=$(mktemp -d /tmp/trace.XXXX)
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 bench.py --warmup 3 --no-cpu-baseline --no-extra "$@" > ${out}.log 2>&1
f=$(find $d -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && cp $f ${out}_kernel_trace.csv
rm -rf $d
