for p in 0 1; do
  echo "== EKAID_PERSIST_POLL=$p"
  EKAID_PERSIST_POLL=$p python tools/seq_bench.py --batch 7 64 256 --steps 40 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('fwd B=%d: step-wise %.1f us/step, persistent %.1f, status %s, max rel err %.2e' % (d['batch'], d['stepwise_us_per_step'], d['persistent_us_per_step'], d['status'], d['max_rel_err']))"
  EKAID_PERSIST_POLL=$p python tools/greedy_time.py 2>/dev/null | grep "mode 1 poll 90"
done
