# A/B of experimental builds of the latency kernel (gpurun_exp/*.so), one box
for lib in "" $(ls gpurun_exp/*.so 2>/dev/null); do
  echo "== ${lib:-default}"
  MDEMOD_LIB_PATH=$lib python tools/lat_bench.py c1 c3 c4 2>&1 | grep -E "streams +(1|1024) "
done
