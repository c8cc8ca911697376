#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv of bench.py (tools/step_trace.sh): cuts it into steps at the Adam kernel and prints, for the
last replayed steps, wall time, busy time (sum of kernel durations), the idle time between consecutive kernels (count and sum of gaps by
size class) and the kernels in front of the largest gaps.   usage: step_trace_read.py <kernel_trace.csv> [steps to average, default 5]"""
import csv
import re
import sys
from collections import Counter, defaultdict


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'\(.*', '', n)[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    nst = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows]
    cuts = [i for i, e in enumerate(ev) if 'adam_dev_kernel' in e[2]]
    steps = [ev[cuts[i] + 1:cuts[i + 1] + 1] for i in range(len(cuts) - 1)]
    steps = [s for s in steps if len(s) > 50]
    mode = Counter(len(s) for s in steps).most_common(1)[0][0]       # the replayed steps all have the same number of kernels
    same = [s for s in steps if len(s) == mode]
    print('%d steps found, %d of them with %d kernels (the replayed / steady ones); averaging the last %d of those' % (len(steps), len(same), mode, nst))
    use = same[-nst:]
    wall = busy = 0.0
    gaps = Counter()
    gapsum = defaultdict(float)
    before = defaultdict(lambda: [0, 0.0])
    per = defaultdict(lambda: [0, 0.0])
    for s in use:
        wall += (s[-1][1] - s[0][0]) / 1e3
        prev_end = None
        for a, b, n in s:
            busy += (b - a) / 1e3
            per[n][0] += 1
            per[n][1] += (b - a) / 1e3
            if prev_end is not None:
                g = (a - prev_end) / 1e3
                cls = '<0 (overlap)' if g < 0 else '0-1 us' if g < 1 else '1-2 us' if g < 2 else '2-4 us' if g < 4 else '4-8 us' if g < 8 else '>8 us'
                gaps[cls] += 1
                gapsum[cls] += g
                if g >= 2:
                    before[(prev_name, n)][0] += 1
                    before[(prev_name, n)][1] += g
            prev_end, prev_name = max(b, prev_end or 0), n
    k = float(len(use))
    print('per step: wall %.1f us, busy %.1f us, idle %.1f us' % (wall / k, busy / k, (wall - busy) / k))
    for cls in ('<0 (overlap)', '0-1 us', '1-2 us', '2-4 us', '4-8 us', '>8 us'):
        print('  gaps %-13s %7.1f per step, %8.1f us per step' % (cls, gaps[cls] / k, gapsum[cls] / k))
    print('largest idle in front of (previous kernel -> kernel): count per step, us per step')
    for (p, n), (c, t) in sorted(before.items(), key=lambda kv: -kv[1][1])[:14]:
        print('  %6.1f %8.1f   %s -> %s' % (c / k, t / k, p[:48], n[:48]))
    print('kernels by time: launches per step, us per step, average us')
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
        print('  %6.1f %8.1f %7.1f   %s' % (c / k, t / k, t / c, n))


if __name__ == '__main__':
    main()
