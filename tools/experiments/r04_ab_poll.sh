#!/bin/bash
# Same-box A/B of the hand-off protocol of the persistent forward kernel: EKAID_PERSIST_POLL=0 (counter seams everywhere) | 1 (the three
# fragment-major hand-offs of a step polled on their own data).  usage: tools/ab_poll.sh <out file>
out=$1
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$REPO"
{
for i in 1 2; do
  for p in 0 1; do
    echo "== round $i EKAID_PERSIST_POLL=$p"
    EKAID_PERSIST_POLL=$p python tools/seq_bench.py --batch 7 32 64 128 256 --steps 40 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('fwd B=%d: step-wise %.1f us/step, persistent %.1f, status %s, max rel err %.2e' % (d['batch'], d['stepwise_us_per_step'], d['persistent_us_per_step'], d['status'], d['max_rel_err']))"
    EKAID_PERSIST_POLL=$p python tools/greedy_time.py 2>/dev/null | grep "mode 1 poll 90"
    EKAID_PERSIST_POLL=$p python bench.py --real-only 2>/dev/null | grep -o '"ms_per_iteration": [0-9.]*'
  done
done
} > "$out" 2>&1
