# the dynamic count of packed FIR instructions per wave-firing on the BASELINE std-window kernels.
# One --pmc pass (SQ_INSTS_VALU ...) of the product and one of gpurun_exp/noskip.so (tools/build_exp_rot.sh noskip "-DROT_EXP_NOSKIP": the skip
# flags zeroed behind hipcc's back, everything else identical), then wall-clock A/B of the two.  usage: tools/price_fir_padding.sh <dir under gpurun_out>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_fir}; mkdir -p $O
for c in c1 c3; do
  B="python3 bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-check"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH -d $O/${c}_ship -o x -- $B > $O/${c}_ship.log 2>&1
  export MDEMOD_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_exp/noskip.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH -d $O/${c}_noskip -o x -- $B > $O/${c}_noskip.log 2>&1
  unset MDEMOD_LIB_PATH
  for d in ship noskip; do python tools/rocpd_summary.py $(find $O/${c}_$d -name "*.db" | head -1) > $O/${c}_$d.md 2>&1; rm -rf $O/${c}_$d; done
done
for rep in 1 2 3; do
  for lib in "" gpurun_exp/noskip.so; do
    for c in c1 c3; do
      MDEMOD_LIB_PATH=$lib python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib': '${lib:-product}', 'config': '$c', 'msps': d['value'], 'kernel_ms': d['roofline']['kernel_ms']}))" >> $O/ab.jsonl
    done
  done
done
grep -h "demod_kernel_rot" $O/*.md | cut -c1-160; cat $O/ab.jsonl
