for lib in "" $(ls gpurun_exp/*.so 2>/dev/null); do echo "lib=${lib:-default}"; MDEMOD_LIB_PATH=$lib python3 tools/ab_f32.py "" 2>&1 | grep "GS/s" | head -3; done
