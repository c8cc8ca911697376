#!/bin/bash
# MFMA-busy and clock of one GEMM shape (two PMC passes).  usage: tools/pmc_quick.sh <outdir> <nt|nn|tn> M N K
out=$1; shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o p -- python3 tools/pmc_gemm.py "$@" > $out/g$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
res = {}
for g in sorted(glob.glob(os.path.join(out, 'g*'))):
    if not os.path.isdir(g): continue
    cc = glob.glob(os.path.join(g, '**', '*counter_collection.csv'), recursive=True)
    kt = glob.glob(os.path.join(g, '**', '*kernel_trace.csv'), recursive=True)
    if not cc: continue
    rows = [r for r in csv.DictReader(open(cc[0])) if 'gemm_' in r['Kernel_Name'] and 'reduce' not in r['Kernel_Name']]
    last = max(int(r['Dispatch_Id']) for r in rows)
    for r in rows:
        if int(r['Dispatch_Id']) == last:
            res[r['Counter_Name']] = float(r['Counter_Value']); res['kernel'] = r['Kernel_Name'][:60]
    if kt:
        k = [r for r in csv.DictReader(open(kt[0])) if 'gemm_' in r['Kernel_Name'] and 'reduce' not in r['Kernel_Name']]
        res['dur_us'] = (int(k[-1]['End_Timestamp']) - int(k[-1]['Start_Timestamp'])) / 1e3
cyc = res['GRBM_GUI_ACTIVE'] / 8
print(res['kernel'], 'dur %.1f us  clock %.3f GHz  MFMA busy %.3f  coexec/busy %.2f  wait_any %.2f  wait_inst %.2f  valu/mfma %.2f' % (
    res['dur_us'], cyc / res['dur_us'] / 1e3, res['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), res['SQ_VALU_MFMA_COEXEC_CYCLES'] / res['SQ_VALU_MFMA_BUSY_CYCLES'],
    res['SQ_WAIT_ANY'] / res['SQ_WAVE_CYCLES'], res['SQ_WAIT_INST_ANY'] / res['SQ_WAVE_CYCLES'], res['SQ_INSTS_VALU'] / res['SQ_INSTS_MFMA']))
PY
find $out -name "*kernel_trace.csv" -delete
