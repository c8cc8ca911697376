#!/bin/bash
# Same-box A/B of the sequence mode (persistent decoder kernels vs the step-wise chains): the SAME library, EKAID_SEQ_MODE switched,
# interleaved, two rounds; B = 256 eager (headline), 32 pairs hipGraph (8-GPU strong-scaling point), T = 90, real-data loop.
# usage: tools/ab_seq.sh <out file>
out=$1
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$REPO"
t=$(mktemp -d /tmp/abseq.XXXX)
for i in 1 2; do
  for m in stepwise persistent; do
    EKAID_SEQ_MODE=$m python bench.py --no-cpu-baseline --steps 60 --warmup 5 --no-real > $t/b256_$m$i.json 2> $t/b256_$m$i.err
    EKAID_SEQ_MODE=$m python bench.py --no-cpu-baseline --no-extra --steps 60 --warmup 5 --batch 32 --hip-graph on > $t/b32_$m$i.json 2> $t/b32_$m$i.err
    EKAID_SEQ_MODE=$m python bench.py --real-only > $t/real_$m$i.json 2> $t/real_$m$i.err
  done
done
python - "$t" > "$out" <<'PY'
import json, sys
t = sys.argv[1]
print('same-box A/B, EKAID_SEQ_MODE = stepwise | persistent (two interleaved rounds)')
for i in (1, 2):
    for m in ('stepwise', 'persistent'):
        try:
            a = json.load(open('%s/b256_%s%d.json' % (t, m, i)))
            b = json.load(open('%s/b32_%s%d.json' % (t, m, i)))
            r = json.load(open('%s/real_%s%d.json' % (t, m, i)))['real_b64']
            ex = a.get('extra') or {}
            print('round %d %-10s  B=256 %.3f ms (%.0f pairs/s, speaker fwd %.2f, backward %.2f)  B=32 graph %.3f ms  T90 %.2f ms  fwd64 %.2f ms  greedy %.4f ms/step  real_b64 %.2f ms/iter (%.0f pairs/s, loader wait %.3f)' % (
                i, m, a['ms_per_step'], a['value'], a['phases_ms'].get('speaker_fwd_loss', 0), a['phases_ms'].get('backward', 0), b['ms_per_step'],
                (ex.get('train_T90') or {}).get('ms_per_step', 0), (ex.get('forward_only_b64') or {}).get('ms', 0),
                (ex.get('greedy_decode_b64') or {}).get('ms_per_decode_step', 0), r['ms_per_iteration'], r['qa_pairs_per_s'], r['loader_wait_fraction']))
        except Exception as e:
            print('round', i, m, 'failed', e)
PY
rm -rf $t
cat "$out"
