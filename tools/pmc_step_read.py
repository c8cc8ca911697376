#!/usr/bin/env python3
"""Summarise the output of tools/pmc_step.sh: per-kernel time table of the traced run and, for the dominant kernel
(ekaid::gemm_split_kernel<..., 2, 2, 32, 1, ...>), the PMC counters averaged per launch over every launch of the short
counter runs.  usage: python tools/pmc_step_read.py <outdir> <summary.json> [steps_in_trace]"""
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True)
    return hits[0] if hits else None


def main():
    out, dst = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 28          # 20 timed + 3 warm-up + 1 + 1 + 2 + (capture: 0)
    summary = {'what': 'tools/pmc_step.sh: rocprofv3 around bench.py (one process, B=256 train step unless stated otherwise)'}
    stats = find(os.path.join(out, 'trace'), 'kernel_stats.csv')
    if stats:
        rows = list(csv.DictReader(open(stats)))
        tot = sum(float(r['TotalDurationNs']) for r in rows)
        summary['kernel_time_ms_per_step'] = round(tot / steps / 1e6, 3)
        summary['launches_per_step'] = round(sum(int(r['Calls']) for r in rows) / steps, 1)
        summary['top_kernels'] = [{'name': r['Name'][:110], 'calls_per_step': round(int(r['Calls']) / steps, 1),
                                   'ms_per_step': round(float(r['TotalDurationNs']) / steps / 1e6, 3),
                                   'avg_us': round(float(r['AverageNs']) / 1e3, 2)} for r in rows[:25]]
        summary['aten_kernels_ms_per_step'] = round(sum(float(r['TotalDurationNs']) for r in rows if 'at::native' in r['Name']) / steps / 1e6, 3)
    key = 'gemm_split_kernel<'
    counters, durs = {}, []
    for g in sorted(glob.glob(os.path.join(out, 'g*'))):
        if not os.path.isdir(g):
            continue
        cc = find(g, 'counter_collection.csv')
        if not cc:
            continue
        acc, n = {}, {}
        for r in csv.DictReader(open(cc)):
            name = r.get('Kernel_Name', '')
            if key not in name or ', 2, 2, 32, 1, ' not in name:          # every 128x128-tile instantiation (pipelined and plain loop)
                continue
            c = r['Counter_Name']
            acc[c] = acc.get(c, 0.0) + float(r['Counter_Value'])
            n[c] = n.get(c, 0) + 1
        for c in acc:
            counters[c] = acc[c] / n[c]
            counters.setdefault('_launches', n[c])
        kt = find(g, 'kernel_trace.csv')
        if kt and not durs:
            for r in csv.DictReader(open(kt)):
                if key in r['Kernel_Name'] and ', 2, 2, 32, 1, ' in r['Kernel_Name']:
                    durs.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    summary['dominant_kernel'] = 'ekaid::gemm_split_kernel<.,.,2,2,32,1,.,3> (all 128x128-tile instantiations, pipelined and plain K loop, every launch of the counter runs)'
    summary['counters_avg_per_launch'] = counters
    if durs:
        durs.sort()
        summary['avg_launch_us_in_pmc_pass'] = round(sum(durs) / len(durs), 2)
    d = {}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in counters and 'GRBM_GUI_ACTIVE' in counters:
        cyc = counters['GRBM_GUI_ACTIVE'] / 8.0                       # per XCD
        d['mfma_busy_frac_of_simd_cycles'] = round(counters['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 256 * 4), 4)
        if durs:
            d['clock_GHz'] = round(cyc / (sum(durs) / len(durs)) / 1e3, 3)
    if 'SQ_INSTS_VALU' in counters and 'SQ_INSTS_MFMA' in counters:
        d['valu_insts_per_mfma'] = round(counters['SQ_INSTS_VALU'] / counters['SQ_INSTS_MFMA'], 2)
    if 'FETCH_SIZE' in counters:
        d['hbm_fetch_bytes_per_launch_corrected'] = int(counters['FETCH_SIZE'] * 1024 * 2)     # gfx950: FETCH_SIZE reads half (guide, HBM section)
    if 'WRITE_SIZE' in counters:
        d['hbm_write_bytes_per_launch'] = int(counters['WRITE_SIZE'] * 1024)
    summary['derived'] = d
    json.dump(summary, open(dst, 'w'), indent=1)
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == '__main__':
    main()
