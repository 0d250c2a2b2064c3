# a longer soak of the round's last build (fresh seeds): tools/soak_long.sh  ->  gpurun_out/r06_soak2/*.log
O=gpurun_out/${1:-r06_soak2}; mkdir -p $O
S=$(date +%s)
for k in 1 2 3; do python tools/config_fuzz.py 1500 $((S+k)) > $O/config_fuzz_$k.log 2>&1; tail -1 $O/config_fuzz_$k.log | cut -c1-80; done
FUZZ_HYB=1 python tools/config_fuzz.py 600 $((S+10)) > $O/config_fuzz_hyb.log 2>&1; tail -1 $O/config_fuzz_hyb.log | cut -c1-80
FUZZ_OSF=47,83.3,139 python tools/config_fuzz.py 400 $((S+11)) > $O/config_fuzz_gather.log 2>&1; tail -1 $O/config_fuzz_gather.log | cut -c1-80
FUZZ_OSF=0.13,0.3,0.55,1.0 python tools/config_fuzz.py 400 $((S+12)) > $O/config_fuzz_sub.log 2>&1; tail -1 $O/config_fuzz_sub.log | cut -c1-80
python tools/api_fuzz.py 3000 $((S+20)) > $O/api_fuzz.log 2>&1; tail -1 $O/api_fuzz.log
python tools/recording_fuzz.py 500 $((S+30)) > $O/recording_fuzz.log 2>&1; tail -2 $O/recording_fuzz.log
FUZZ_SYMBOLS=42e6,83e6 FUZZ_RAMPS=0,1.5,-2 python tools/recording_fuzz.py 16 $((S+31)) > $O/recording_fuzz_long.log 2>&1; tail -2 $O/recording_fuzz_long.log
python tools/cli_fuzz.py 400 $((S+40)) > $O/cli_fuzz.log 2>&1; tail -1 $O/cli_fuzz.log
python tools/recording_opts_fuzz.py 200 $((S+50)) > $O/recording_opts_fuzz.log 2>&1; tail -1 $O/recording_opts_fuzz.log
for k in 1 2; do python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done
