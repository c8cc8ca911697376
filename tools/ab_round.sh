#!/bin/bash
# Same-box A/B of two trees: _base/ (a checkout of the previous round's final commit with its own built library; git-ignored, travels with
# gpurun) against this tree, interleaved, two rounds: 256-pair headline with the extras (T = 90, forward-only 64, greedy decode), 32 pairs
# replayed from a hipGraph (the 8-GPU strong-scaling point), the real-data loop.   usage: tools/ab_round.sh <out file>
out=$1
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$REPO"
t=$(mktemp -d /tmp/abround.XXXX)
for i in 1 2; do
  for w in base new; do
    d=$REPO; [ $w = base ] && d=$REPO/_base
    (cd $d && python bench.py --no-cpu-baseline --steps 60 --warmup 5 --no-real > $t/b256_$w$i.json 2> $t/b256_$w$i.err)
    (cd $d && python bench.py --no-cpu-baseline --no-extra --steps 60 --warmup 5 --batch 32 --hip-graph on > $t/b32_$w$i.json 2> $t/b32_$w$i.err)
    (cd $d && python bench.py --real-only > $t/real_$w$i.json 2> $t/real_$w$i.err)
  done
done
python - "$t" > "$out" <<'PY'
import json, sys
t = sys.argv[1]
print('same-box A/B: base = previous round\'s final tree (_base/), new = this tree; two interleaved rounds')
for i in (1, 2):
    for w in ('base', 'new'):
        try:
            a = json.load(open('%s/b256_%s%d.json' % (t, w, i)))
            b = json.load(open('%s/b32_%s%d.json' % (t, w, i)))
            r = json.load(open('%s/real_%s%d.json' % (t, w, i)))['real_b64']
            ex = a.get('extra') or {}
            print('round %d %-4s  B=256 %.3f ms (%.0f pairs/s)  B=32 graph %.3f ms  T90 %.2f ms  fwd64 %.2f ms  greedy %.4f ms/step  real_b64 %.2f ms/iter (%.0f pairs/s, loader wait %.4f)' % (
                i, w, a['ms_per_step'], a['value'], b['ms_per_step'],
                (ex.get('train_T90') or {}).get('ms_per_step', 0), (ex.get('forward_only_b64') or {}).get('ms', 0),
                (ex.get('greedy_decode_b64') or {}).get('ms_per_decode_step', 0), r['ms_per_iteration'], r['qa_pairs_per_s'], r['loader_wait_fraction']))
        except Exception as e:
            print('round %d %s failed: %r' % (i, w, e))
PY
rm -rf $t
