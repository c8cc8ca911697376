#!/bin/bash
# PMC passes over the graph-attention core as the step runs it (tools/gat_bench.py: forward through ops.gat_attention, backward through
# autograd: ekaid_gat_bwd_core from 256 slices on, else batched products + masked_softmax_bwd; then gat_dqk): the counters of the LAST launch of each of its own kernels.
# usage: tools/pmc_gat_all.sh <outdir> R N M A H      -> <outdir>/summary.json      (separate --pmc passes, kernel trace only)
out=$1; shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o p -- python3 tools/gat_bench.py "$@" > $out/g$i.log 2>&1
done
python3 - "$out" "$@" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
R, N, M, A, H = [int(x) for x in sys.argv[2:7]]
KERNELS = {'gat_fused_fwd': R * (3 * N * A * 4 + N * H * A * 4 + N * H * M * 4),             # q, k, S in; outT, P out
           'gat_dqk': R * (2 * N * H * M * 4 + 2 * N * A * 4 + 2 * N * A * 4),              # daff (+P) in, q, k in; dq, dk out
           'masked_softmax_bwd': R * (3 * N * H * M * 4),                                   # P, dP in; daff out (+ small label / geometry sums)
           'gat_bwd_core': R * (N * H * A * 4 + 2 * N * A * 4 + 2 * N * H * M * 4 + 2 * N * N * 4)}   # d(outT), S, P, adjacency in; dS, daff, label-bias gradient out
res = {k: {} for k in KERNELS}
for g in sorted(glob.glob(os.path.join(out, 'g*'))):
    if not os.path.isdir(g):
        continue
    cc = glob.glob(os.path.join(g, '**', '*counter_collection.csv'), recursive=True)
    kt = glob.glob(os.path.join(g, '**', '*kernel_trace.csv'), recursive=True)
    for name in KERNELS:
        if cc:
            rows = [r for r in csv.DictReader(open(cc[0])) if name in r['Kernel_Name']]
            if rows:
                last = max(int(r['Dispatch_Id']) for r in rows)
                for r in rows:
                    if int(r['Dispatch_Id']) == last:
                        res[name][r['Counter_Name']] = float(r['Counter_Value'])
                        res[name]['_kernel'] = r['Kernel_Name'][:100]
                        for k in ('VGPR_Count', 'LDS_Block_Size', 'Workgroup_Size', 'Grid_Size'):
                            if k in r:
                                res[name]['_' + k] = r[k]
        if kt and '_dur_us' not in res[name]:
            d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt[0])) if name in r['Kernel_Name']]
            if d:
                res[name]['_dur_us'] = sorted(d)[len(d) // 2]
                res[name]['_launches_traced'] = len(d)
summary = {'what': 'tools/pmc_gat_all.sh: rocprofv3 --kernel-trace --pmc <group> -- python3 tools/gat_bench.py R N M A H (one pass per counter group); '
                   'per kernel the counters of its last launch and the median duration over the traced launches. FETCH_SIZE is in KiB and under-counts '
                   'by 2 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section): hbm_fetch_bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024.',
           'shape': dict(R=R, N=N, M=M, A=A, H=H), 'kernels': {}}
for name, alg in KERNELS.items():
    c = res[name]
    if '_dur_us' not in c or 'GRBM_GUI_ACTIVE' not in c:
        summary['kernels'][name] = {'error': 'not seen', 'counters': c}
        continue
    dur = c['_dur_us']
    clock = c['GRBM_GUI_ACTIVE'] / 8 / dur / 1e3
    simd_cycles = 1024 * dur * 1e-6 * clock * 1e9
    fetch, write = c.get('FETCH_SIZE', 0) * 1024 * 2, c.get('WRITE_SIZE', 0) * 1024
    d = {'median_duration_us': round(dur, 1), 'clock_GHz': round(clock, 3), 'algorithmic_bytes': alg, 'algorithmic_TBps': round(alg / dur / 1e6, 2),
         'frac_of_8TBps': round(alg / dur / 1e6 / 8.0, 3),
         'hbm_fetch_bytes_corrected': int(fetch), 'hbm_write_bytes': int(write), 'hbm_traffic_over_algorithmic': round((fetch + write) / alg, 2),
         'hbm_TBps_moved': round((fetch + write) / dur / 1e6, 2),
         'l2_hit_rate': round(c['TCC_HIT_sum'] / max(c['TCC_HIT_sum'] + c['TCC_MISS_sum'], 1), 4) if 'TCC_HIT_sum' in c else None,
         'valu_busy_frac_of_simd_cycles': round(c.get('SQ_ACTIVE_INST_VALU', 0) / simd_cycles, 4),
         'wave_cycles_waiting_frac': round(c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1), 4),
         'waves_resident_avg_per_simd': round(c.get('SQ_WAVE_CYCLES', 0) / simd_cycles, 3),
         'lds_bank_conflict_frac_of_lds_active': round(c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1), 4),
         'valu_insts_per_wave': round(c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_WAVES', 1), 1), 1),
         'vmem_insts_per_wave': round(c.get('SQ_INSTS_VMEM', 0) / max(c.get('SQ_WAVES', 1), 1), 1)}
    summary['kernels'][name] = {'derived': d, 'counters': c}
json.dump(summary, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
print(json.dumps({k: v.get('derived', v) for k, v in summary['kernels'].items()}, indent=1))
PY
find $out -name "*.csv" -delete; find $out -name "*.db" -delete
