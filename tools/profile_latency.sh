#!/bin/bash
# Kernel trace of the one-frame-per-call path only (step 5 of tools/profile_round.sh):  gpurun -- bash tools/profile_latency.sh <tag>
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/latency_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "rocprofv3 --kernel-trace -- python3 tools/time_frame_chain.py" > $OUT/cmd_latency.txt
rocprofv3 --kernel-trace -d $OUT/latency_trace -o trace -- python3 $R/tools/time_frame_chain.py > $OUT/latency.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $OUT/latency_trace -name "*.db" | head -1) grid > $OUT/latency_kernel_stats.txt
rm -rf $OUT/latency_trace
grep "ms per frame\|chain alone" $OUT/latency.log
