# A/B of experimental builds on one box: headline kernel (c1) and OQPSK (c3), no CPU leg
for lib in "" $(ls gpurun_exp/*.so 2>/dev/null); do
  for c in ${CFGS:-c1 c3}; do
    MDEMOD_LIB_PATH=$lib python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-default}', '$c', d['value'], 'MS/s', d['roofline']['kernel_ms'], 'ms')"
  done
done
