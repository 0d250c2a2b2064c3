# A/B of kernel generations on one box: MDEMOD_KERNEL="" (default: v3 where it applies) against v2, headline (c1), OQPSK (c3), wide (c4)
for k in "" v2; do
  for c in ${CFGS:-c1 c3}; do
    MDEMOD_KERNEL=$k python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline ${CHECK:---no-check} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${k:-default}', '$c', d['value'], 'MS/s', d['roofline']['kernel_ms'], 'ms', d['roofline'].get('kernel'), d.get('spot_check'))"
  done
done
