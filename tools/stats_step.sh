#!/bin/bash
# rocprofv3 --kernel-trace --stats of `bench.py <args>` -> <out>_kernel_stats.csv (summary only)
# usage: tools/stats_step.sh <out prefix under gpurun_out/> [bench args...]
out=$1; shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
d=$(mktemp -d /tmp/stats.XXXX)
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --warmup 3 --no-cpu-baseline --no-extra "$@" > ${out}.log 2>&1
f=$(find $d -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f ${out}_kernel_stats.csv
rm -rf $d
