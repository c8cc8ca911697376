for i in 1 2; do for v in vbase vnt; do
  export EKAID_HIP_LIB=$GRAFT_REPO_ROOT/ekaid_amd/lib/libekaid_$v.so
  python bench.py --no-cpu-baseline --no-extra --steps 80 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v B=256', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
  python bench.py --no-cpu-baseline --no-extra --steps 80 --warmup 5 --batch 32 --hip-graph on 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v B=32 graph', d['ms_per_step'])"
  python bench.py --real-only 2>/dev/null | grep -o '"ms_per_iteration": [0-9.]*' | head -1
done; done
