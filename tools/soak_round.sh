mkdir -p gpurun_out/r06c
python -m pytest tests/test_gpu_parity.py -q -k "pinning_that_covers or pinned" > gpurun_out/r06c/pin.log 2>&1; tail -3 gpurun_out/r06c/pin.log
# N > 1 control flow at real sizes: 4 ranks on one GPU at the full shape, 8 at a quarter of the tiles
python bench.py --gpus 4 --oversubscribe --steps 3 --warmup 1 > gpurun_out/r06c/n4.out 2> gpurun_out/r06c/n4.err; echo "rc=$?" >> gpurun_out/r06c/n4.err
python bench.py --gpus 8 --oversubscribe --tiles 98304 --steps 3 --warmup 1 > gpurun_out/r06c/n8.out 2> gpurun_out/r06c/n8.err; echo "rc=$?" >> gpurun_out/r06c/n8.err
MDEMOD_BENCH_FAULT=fanin_hang@3 MDEMOD_BENCH_POST_DEADLINE_S=60 python bench.py --gpus 4 --oversubscribe --tiles 98304 --steps 2 --warmup 1 > gpurun_out/r06c/n4hang.out 2> gpurun_out/r06c/n4hang.err; echo "rc=$?" >> gpurun_out/r06c/n4hang.err
wc -c gpurun_out/r06c/*.out; tail -c 300 gpurun_out/r06c/n4.err
# soaks on the round's build, fresh seeds
S=$(date +%s)
python tools/config_fuzz.py 1500 $S > gpurun_out/r06c/config_fuzz.log 2>&1; tail -2 gpurun_out/r06c/config_fuzz.log
FUZZ_HYB=1 python tools/config_fuzz.py 400 $((S+1)) > gpurun_out/r06c/config_fuzz_hyb.log 2>&1; tail -2 gpurun_out/r06c/config_fuzz_hyb.log
python tools/api_fuzz.py 1500 $((S+2)) > gpurun_out/r06c/api_fuzz.log 2>&1; tail -2 gpurun_out/r06c/api_fuzz.log
python tools/recording_fuzz.py 200 $((S+3)) > gpurun_out/r06c/recording_fuzz.log 2>&1; tail -3 gpurun_out/r06c/recording_fuzz.log
python tools/cli_fuzz.py 250 $((S+4)) > gpurun_out/r06c/cli_fuzz.log 2>&1; tail -2 gpurun_out/r06c/cli_fuzz.log
python tools/recording_opts_fuzz.py 100 $((S+5)) > gpurun_out/r06c/recording_opts_fuzz.log 2>&1; tail -2 gpurun_out/r06c/recording_opts_fuzz.log
