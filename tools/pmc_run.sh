#!/bin/bash
# PMC passes over one GEMM shape (tuning aid).  usage: tools/pmc_run.sh <outdir> <mode> <M> <N> <K>
# (a group the hardware cannot collect in one pass makes rocprofv3 abort and hang: every pass runs under timeout)
# Each counter group is its own rocprofv3 run (kernel-trace + pmc only), as the pool requires.
out=$1; shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
i=0
first=${PMC_FIRST:-1}
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  [ $i -lt $first ] && continue
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o p -- python3 tools/pmc_gemm.py "$@" > $out.g$i.log 2>&1
done
