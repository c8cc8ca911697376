for i in 1 2; do for m in stepwise persistent; do
  EKAID_SEQ_MODE=$m python bench.py --no-cpu-baseline --no-extra --steps 80 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'], d['phases_ms'])"
done; done
