# rocprofv3 passes behind profiles/rNN_kernels.md (tools/profile_round.sh <dir under gpurun_out> ["c1 c3 c4"]; then tools/make_profile_md.py): kernel stats and PMC (separate passes, no trace domains with --pmc)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03}; mkdir -p $O
CFGS=${2:-"c1 c3 c4"}
for c in $CFGS; do
  B="python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-check"
  rocprofv3 --kernel-trace --stats -d $O/${c}_stats -o x -- $B > $O/${c}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/${c}_FETCH -o x -- $B > $O/${c}_F.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/${c}_WRITE -o x -- $B > $O/${c}_W.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/${c}_sq -o x -- $B > $O/${c}_sq.log 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH GRBM_GUI_ACTIVE -d $O/${c}_sq2 -o x -- $B > $O/${c}_sq2.log 2>&1
  for d in stats FETCH WRITE sq sq2; do python tools/rocpd_summary.py $(find $O/${c}_$d -name "*.db" | head -1) > $O/${c}_$d.md 2>&1; done
  rm -rf $O/${c}_stats $O/${c}_FETCH $O/${c}_WRITE $O/${c}_sq $O/${c}_sq2
done
grep -h "demod_kernel" $O/*.md | cut -c1-200 | head -60
