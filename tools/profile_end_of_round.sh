cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r01f; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/c1_stats -o x -- python3 bench.py --config c1 --steps 5 --warmup 2 --no-cpu-baseline --no-check > $O/c1_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/c1_FETCH -o x -- python3 bench.py --config c1 --steps 2 --warmup 1 --no-cpu-baseline --no-check > $O/c1_F.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/c1_WRITE -o x -- python3 bench.py --config c1 --steps 2 --warmup 1 --no-cpu-baseline --no-check > $O/c1_W.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/c1_sq -o x -- python3 bench.py --config c1 --steps 2 --warmup 1 --no-cpu-baseline --no-check > $O/c1_sq.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c3_stats -o x -- python3 bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-check > $O/c3_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c4_stats -o x -- python3 bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-check > $O/c4_stats.log 2>&1
python bench.py > $O/bench_c1.json 2> $O/bench_c1.err
for d in c1_stats c1_FETCH c1_WRITE c1_sq c3_stats c4_stats; do python tools/rocpd_summary.py $(find $O/$d -name "*.db" | head -1) > $O/$d.md 2>&1; done
grep -h "demod_kernel_rw" $O/*.md | cut -c1-200 | head -30
tail -c 600 $O/bench_c1.json
