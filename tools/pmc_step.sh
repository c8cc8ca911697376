#!/bin/bash
# rocprofv3 evidence for one bench configuration (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of `bench.py <args>`            -> $out/trace
#   2. three separate --pmc passes over a short run of the same command (MFMA/SQ group, FETCH_SIZE, WRITE_SIZE)
# usage: tools/pmc_step.sh <outdir> [bench args...]      (the program itself follows `--`: no env/bash hop)
out=$1; shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra "$@" > $out/trace.log 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra "$@" > $out/g$i.log 2>&1
done
# keep the summaries only: the per-launch traces are tens of MB and gpurun copies back at most 64 MiB
rm -f $out/trace/*kernel_trace.csv
