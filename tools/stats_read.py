#!/usr/bin/env python3
"""Per-step view of a rocprofv3 kernel_stats.csv of `bench.py --steps K --warmup W` (+2 profiled eager steps, +1 issue-time
step [+1 eager]): launches and microseconds per step, grouped by kernel family.   usage: stats_read.py <csv> <steps run> [top]"""
import csv
import re
import sys


def fam(name):
    name = re.sub(r'^void ', '', name)
    m = re.match(r'(?:ekaid::)?([A-Za-z0-9_]+)(<[^>]*>)?', name)
    base = m.group(1) if m else name[:40]
    if base == 'gemm_split_kernel':
        t = re.search(r'<(\w+), (\w+), (\d), (\d)', name)
        return 'gemm_split %sx%s %s%s' % (64 * int(t.group(3)), 64 * int(t.group(4)), 'k' if t.group(1) == 'true' else 'm', 'k' if t.group(2) == 'true' else 'n')
    if base == 'gemm_split_pair_kernel':
        t = re.search(r'<(\w+), (\w+), (\w+)', name)          # <A_KC, B_KC, BIG, ...>: BIG = 128x128 tiles
        return 'gemm_pair %s %s%s' % ('128x128' if t.group(3) == 'true' else '64x64', 'k' if t.group(1) == 'true' else 'm', 'k' if t.group(2) == 'true' else 'n')
    if base == 'at':
        return 'aten:' + name[:60]
    return base


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    agg = {}
    for r in rows:
        k = fam(r['Name'])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r['Calls'])
        a[1] += float(r['TotalDurationNs'])
    tot_l = sum(a[0] for a in agg.values()) / steps
    tot_t = sum(a[1] for a in agg.values()) / steps / 1e6
    print('%-44s %9s %10s %9s' % ('kernel family', 'launches', 'ms/step', 'avg us'))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-44s %9.1f %10.3f %9.1f' % (k[:44], a[0] / steps, a[1] / steps / 1e6, a[1] / a[0] / 1e3))
    print('%-44s %9.1f %10.3f' % ('TOTAL', tot_l, tot_t))


if __name__ == '__main__':
    main()
