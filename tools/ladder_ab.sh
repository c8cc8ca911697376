#!/bin/bash
# Same-box A/B of one environment switch over the 256-pair line and the 128 / 64 / 32-pair rungs (bench.py --ladder-only), interleaved.
# usage: tools/ladder_ab.sh <out file> VAR=VALUE [VAR2=VALUE2 ...]
out=$1; shift
: > $out
for tag in off on off on; do
  if [ $tag = on ]; then env "$@" python bench.py --ladder-only --no-cpu-baseline > /tmp/ladder.json 2> /tmp/ladder.err
  else python bench.py --ladder-only --no-cpu-baseline > /tmp/ladder.json 2> /tmp/ladder.err; fi
  python - "$tag" "$*" >> $out <<'PY'
import json, sys
try:
    d = json.loads([l for l in open('/tmp/ladder.json') if l.startswith('{')][-1])
    lad = d['extra']['per_gpu_ladder']['rungs']
    print('%-3s %s: 256: %.3f ms @%s |  ' % (sys.argv[1], sys.argv[2] if sys.argv[1] == 'on' else '(default)', d['ms_per_step'], (d.get('clocks') or {}).get('sclk_mhz')) + '  '.join(
        '%s: %.3f ms @%s MHz (%s launches)' % (r['per_gpu_batch'], r['ms_per_step'], (r.get('clocks_under_load') or {}).get('sclk_mhz'), r.get('launches_per_step')) for r in lad))
except Exception as e:
    print('%s: failed %r' % (sys.argv[1], e), open('/tmp/ladder.err').read()[-800:])
PY
done
cat $out
