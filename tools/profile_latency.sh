R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lat_trace
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/t -o trace -- python3 $R/tools/time_frame_chain.py > $OUT/log.txt 2>&1
python3 $R/tools/rocpd_stats.py $(find $OUT/t -name "*.db" | head -1) grid > $OUT/stats.txt
rm -rf $OUT/t
tail -5 $OUT/log.txt
