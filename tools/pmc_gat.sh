#!/bin/bash
# PMC passes over the fused graph-attention forward kernel alone (tools/gat_bench.py --kernel: 35 launches through the C ABI; the
# counters of the last one).  usage: tools/pmc_gat.sh <outdir> R N M A H      -> <outdir>/summary.json
out=$1; shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o p -- python3 tools/gat_bench.py --kernel ekaid_amd/lib/libekaid_hip.so "$@" > $out/g$i.log 2>&1
done
python3 - "$out" "$@" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
R, N, M, A, H = [int(x) for x in sys.argv[2:7]]
res = {}
for g in sorted(glob.glob(os.path.join(out, 'g*'))):
    if not os.path.isdir(g): continue
    cc = glob.glob(os.path.join(g, '**', '*counter_collection.csv'), recursive=True)
    if not cc: continue
    rows = [r for r in csv.DictReader(open(cc[0])) if 'gat_fused_fwd' in r['Kernel_Name']]
    if not rows: continue
    last = max(int(r['Dispatch_Id']) for r in rows)
    for r in rows:
        if int(r['Dispatch_Id']) == last:
            res[r['Counter_Name']] = float(r['Counter_Value'])
            res['_kernel'] = r['Kernel_Name'][:90]
            for k in ('VGPR_Count', 'LDS_Block_Size', 'Workgroup_Size', 'Grid_Size'):
                if k in r: res['_' + k] = r[k]
    kt = glob.glob(os.path.join(g, '**', '*kernel_trace.csv'), recursive=True)
    if kt and '_dur_us' not in res:
        d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt[0])) if 'gat_fused_fwd' in r['Kernel_Name']]
        res['_dur_us'] = sorted(d)[len(d) // 2]
c = res
dur = c['_dur_us']
clock = c['GRBM_GUI_ACTIVE'] / 8 / dur / 1e3
simd_cycles = 1024 * dur * 1e-6 * clock * 1e9
alg = R * (3 * N * A * 4 + N * H * A * 4 + N * H * M * 4)
flop = 2.0 * R * (N * M * A + N * H * M * A)
d = {'shape': dict(R=R, N=N, M=M, A=A, H=H), 'median_duration_us': dur, 'clock_GHz': round(clock, 3),
     'valu_busy_frac_of_simd_cycles': round(c['SQ_ACTIVE_INST_VALU'] * 4 / simd_cycles / 4, 4) if False else round(c['SQ_ACTIVE_INST_VALU'] / simd_cycles, 4),
     'valu_insts_per_wave': round(c['SQ_INSTS_VALU'] / c['SQ_WAVES'], 1), 'lds_insts_per_wave': round(c['SQ_INSTS_LDS'] / c['SQ_WAVES'], 1),
     'vmem_insts_per_wave': round(c['SQ_INSTS_VMEM'] / c['SQ_WAVES'], 1),
     'wave_cycles_waiting_frac': round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 4),
     'wave_cycles_waiting_on_issue_frac': round(c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], 4),
     'wave_cycles_waiting_on_lds_frac': round(c['SQ_WAIT_INST_LDS'] / c['SQ_WAVE_CYCLES'], 4),
     'lds_array_busy_frac_per_cu': round(c['SQ_LDS_IDX_ACTIVE'] / (256 * dur * 1e-6 * clock * 1e9), 4),
     'lds_bank_conflict_frac_of_lds_active': round(c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), 4),
     'waves_resident_avg_per_simd': round(c['SQ_WAVE_CYCLES'] / simd_cycles, 3),
     'f32_fma_tflops': round(flop / (dur * 1e-6) / 1e12, 1), 'f32_vector_peak_tflops': 157.3,
     'hbm_fetch_bytes_corrected': int(c['FETCH_SIZE'] * 1024 * 2), 'hbm_write_bytes': int(c.get('WRITE_SIZE', 0) * 1024),
     'algorithmic_bytes': alg, 'algorithmic_TBps': round(alg / dur / 1e6, 2),
     'l2_hit_rate': round(c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']), 4)}
json.dump({'what': 'tools/pmc_gat.sh: rocprofv3 --kernel-trace --pmc <group> -- python3 tools/gat_bench.py --kernel ... (separate passes); counters of the last launch of ' + c.get('_kernel', '?'),
           'counters': {k: v for k, v in c.items()}, 'derived': d}, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
print(json.dumps(d, indent=1))
PY
find $out -name "*.csv" -delete; find $out -name "*.db" -delete
