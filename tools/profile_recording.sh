cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02rec; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o x -- python3 tools/recording_native_profile.py 64 > $O/stats.log 2>&1
python tools/rocpd_summary.py $(find $O/stats -name "*.db" | head -1) > $O/recording_stats.md 2>&1
rm -rf $O/stats
head -40 $O/recording_stats.md
