#!/bin/bash
# Same-box A/B of one environment switch: bench.py with and without `VAR=VALUE`, interleaved, two rounds.
# usage: tools/ab_env.sh <out prefix> VAR=VALUE [bench args...]
out=$1; kv=$2; shift; shift
for i in 1 2; do
  python bench.py --no-cpu-baseline "$@" > ${out}_off$i.json 2> ${out}_off$i.err
  env "$kv" python bench.py --no-cpu-baseline "$@" > ${out}_on$i.json 2> ${out}_on$i.err
done
python - "$out" <<'PY'
import json, sys
o = sys.argv[1]
for k in ('off1', 'on1', 'off2', 'on2'):
    try:
        d = json.load(open('%s_%s.json' % (o, k)))
        ex = d.get('extra') or {}
        print(k, 'ms/step', d['ms_per_step'], 'big', d['roofline']['kernel_ms_per_step'], 'family', d['roofline']['gemm_family']['ms_per_step'],
              'T90', (ex.get('train_T90') or {}).get('ms_per_step'), 'fwd64', (ex.get('forward_only_b64') or {}).get('ms'))
    except Exception as e:
        print(k, 'failed', e)
PY
