#!/bin/bash
# Same-box sweep of the bounds-pass thresholds over the 128 / 64 / 32-pair steps (bench.py --ladder-only).
# usage: tools/ladder_sweep.sh <out file> "<min_work>:<min_work_shared>" ...
out=$1; shift
: > $out
for kv in "$@" "$1"; do
  mw=${kv%%:*}; ms=${kv##*:}
  EKAID_F16X2_MIN_WORK=$mw EKAID_F16X2_MIN_WORK_SHARED=$ms python bench.py --ladder-only --no-cpu-baseline > /tmp/ladder.json 2> /tmp/ladder.err
  python - "$mw" "$ms" >> $out <<'PY'
import json, sys
try:
    d = json.loads([l for l in open('/tmp/ladder.json') if l.startswith('{')][-1])
    lad = d['extra']['per_gpu_ladder']['rungs']
    print('min_work %s shared %s: 256: %.3f ms |  ' % (sys.argv[1], sys.argv[2], d['ms_per_step']) + '  '.join('%s: %.3f ms @%s MHz (%s launches)' % (
        r['per_gpu_batch'], r['ms_per_step'], (r.get('clocks_under_load') or {}).get('sclk_mhz'), r.get('launches_per_step')) for r in lad))
except Exception as e:
    print('min_work %s: failed %r' % (sys.argv[1], e), open('/tmp/ladder.err').read()[-800:])
PY
done
cat $out
