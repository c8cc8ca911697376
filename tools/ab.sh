#!/bin/bash
# Same-box A/B: the round-2 tree under _base/ (calibration) and the current tree, interleaved, same flags.
# usage: tools/ab.sh <out prefix> [bench args...]   -> <prefix>_base.json / <prefix>_new.json (two runs each)
out=$1; shift
for i in 1 2; do
  (cd _base && python bench.py --no-cpu-baseline --no-extra "$@" > ../${out}_base$i.json 2> ../${out}_base$i.err)
  python bench.py --no-cpu-baseline --no-extra "$@" > ${out}_new$i.json 2> ${out}_new$i.err
done
python - "$out" <<'PY'
import json, sys
o = sys.argv[1]
for k in ('base1', 'new1', 'base2', 'new2'):
    try:
        d = json.load(open('%s_%s.json' % (o, k)))
        print(k, d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['gemm_family']['ms_per_step'])
    except Exception as e:
        print(k, 'failed', e)
PY
