"""Summarise a rocprofv3 kernel-trace CSV produced around tools/gemm_trace.py: per consecutive run of identical
(kernel, grid) dispatches print count / median duration."""
import csv, sys, statistics, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
runs = []
for r in rows:
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('ekaid::', '')
    key = (name, r['Grid_Size_X'], r['Grid_Size_Z'], r.get('LDS_Block_Size', ''))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if runs and runs[-1][0] == key:
        runs[-1][1].append(d)
    elif len(runs) > 1 and runs[-2][0] == key:      # alternating gemm / reduce pairs
        runs[-2][1].append(d)
    else:
        runs.append((key, [d]))
for key, ds in runs:
    print('%-58s grid %8s z %3s  n %3d  median %8.2f us  min %8.2f' % (key[0][:58], key[1], key[2], len(ds), statistics.median(ds), min(ds)))
