# rocprofv3 A/B of kernel generations on one box: kernel stats + two SQ counter passes per (kernel, config).
# usage: tools/profile_ab.sh <outdir-name> "<configs>" "<MDEMOD_KERNEL values, '-' = default>"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; mkdir -p $O
CFGS=${2:-"c1 c3"}
KERNELS=${3:-"- v2"}
for k in $KERNELS; do
  kk=$k; [ "$k" = "-" ] && kk=""
  export MDEMOD_KERNEL=$kk
  for c in $CFGS; do
    B="python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-check"
    T=${k}_${c}
    rocprofv3 --kernel-trace --stats -d $O/${T}_stats -o x -- $B > $O/${T}_stats.log 2>&1
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/${T}_sq -o x -- $B > $O/${T}_sq.log 2>&1
    rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH GRBM_GUI_ACTIVE -d $O/${T}_sq2 -o x -- $B > $O/${T}_sq2.log 2>&1
    for d in stats sq sq2; do python tools/rocpd_summary.py $(find $O/${T}_$d -name "*.db" | head -1) > $O/${T}_$d.md 2>&1; done
    rm -rf $O/${T}_stats $O/${T}_sq $O/${T}_sq2
  done
done
grep -h "demod_kernel" $O/*.md | cut -c1-60,100-200 | head -80
