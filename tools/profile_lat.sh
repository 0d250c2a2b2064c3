cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/lat; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o x -- python3 tools/lat_profile.py c1 > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/sq -o x -- python3 tools/lat_profile.py c1 > $O/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d $O/sq2 -o x -- python3 tools/lat_profile.py c1 > $O/sq2.log 2>&1
for d in stats sq sq2; do python tools/rocpd_summary.py $(find $O/$d -name "*.db" | head -1) > $O/$d.md 2>&1; done
grep -h "demod_kernel_lat" $O/*.md | cut -c1-220
tail -2 $O/sq2.log
