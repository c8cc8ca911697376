import csv, glob, collections
f = glob.glob('gpurun_out/amax_trace/**/*kernel_trace.csv', recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'amax_rc' not in r['Kernel_Name']:
        continue
    k = (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']) if 'Grid_Size_X' in r else (r.get('Grid_Size'),)
    d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    print(k, 'n=%d median %.1f us min %.1f' % (len(v), v[len(v)//2], v[0]))
