# A/B of latency-kernel builds on one box: per-stream rate at 1 and 1024 streams (tools/lat_bench.py) for the product library and every gpurun_exp/*.so
for lib in "" $(ls gpurun_exp/*.so 2>/dev/null); do
  echo "lib=${lib:-default}"
  MDEMOD_LIB_PATH=$lib python3 tools/lat_bench.py ${CFGS:-c1 c3} 2>&1 | grep -E "streams +(1|1024) x" | cut -c1-200
done
