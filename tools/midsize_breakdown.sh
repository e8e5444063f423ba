for B in 8192 12288 18432 24576 65536; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --batch $B --no-full-solves --no-dense-blocks --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print(json.dumps({'B': $B, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'partitions': d.get('solve', {}).get('partitions'), 'kernel': r['kernel'], 'facts': r['factorizations_per_launch'], 'k_ms': r['kernel_ms_per_iteration'], 'launches': r.get('launches_per_iteration')}))
"
done
