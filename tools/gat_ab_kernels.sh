#!/bin/bash
# Sum of the backward's kernel time (rocprofv3 --kernel-trace --stats of tools/gat_bench.py) with and without the one-launch core, per slice count.
# usage: tools/gat_ab_kernels.sh "R N" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for shape in "$@"; do
  set -- $shape
  for v in 0 1; do
    rm -rf /tmp/gt; EKAID_GAT_BWD_CORE=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gt -o t -- python tools/gat_bench.py $1 $2 $2 1024 4 > /dev/null 2>&1
    f=$(find /tmp/gt -name "*kernel_stats.csv" | head -1)
    python - "$f" "$1" "$2" "$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = max(int(r['Calls']) for r in rows if 'gat_fused_fwd' in r['Name'])
fwdk = ('gat_fused_fwd',)
tot = 0.0; parts = []
for r in rows:
    per = float(r['TotalDurationNs']) / calls / 1e3
    nm = r['Name'].replace('void ', '').replace('ekaid::', '')[:40]
    if any(k in r['Name'] for k in fwdk) or 'Fill' in r['Name'] or 'elementwise' in r['Name']:
        continue
    parts.append('%s %.1f' % (nm.split('(')[0][:28], per)); tot += per
print('R=%s N=%s core=%s: non-forward kernels %.1f us per fwd+bwd  [%s]' % (sys.argv[2], sys.argv[3], sys.argv[4], tot, '; '.join(parts[:7])))
PY
  done
done
