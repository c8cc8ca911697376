#!/bin/bash
# Detailed PMC passes over one GEMM shape (tuning aid).  usage: tools/pmc_gemm2.sh <outdir> <nt|nn|tn> M N K
out=$1; shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o p -- python3 tools/pmc_gemm.py "$@" > $out/g$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for g in sorted(glob.glob(os.path.join(out, 'g*'))):
    if not os.path.isdir(g): continue
    cc = glob.glob(os.path.join(g, '**', '*counter_collection.csv'), recursive=True)
    if not cc: continue
    rows = [r for r in csv.DictReader(open(cc[0])) if 'gemm_' in r['Kernel_Name'] and 'split_planes' not in r['Kernel_Name'] and 'reduce' not in r['Kernel_Name']]
    if not rows: continue
    last = max(int(r['Dispatch_Id']) for r in rows)
    for r in rows:
        if int(r['Dispatch_Id']) == last:
            res[r['Counter_Name']] = float(r['Counter_Value'])
            res['_kernel'] = r['Kernel_Name'][:90]
    kt = glob.glob(os.path.join(g, '**', '*kernel_trace.csv'), recursive=True)
    if kt and '_dur_us' not in res:
        d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt[0])) if 'gemm_' in r['Kernel_Name'] and 'split_planes' not in r['Kernel_Name'] and 'reduce' not in r['Kernel_Name']]
        res['_dur_us'] = sorted(d)[len(d) // 2]
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find $out -name "*.csv" -delete; find $out -name "*.db" -delete
