#!/bin/bash
# kernel trace of the multi-GPU BA schedule at one rank (bench.py's global_ba leg, 32 cameras x 16 objects):  gpurun -- bash tools/prof_global_ba.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/global_ba
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gba.py <<PY
import sys, os
sys.path.insert(0, "$R")
import bench
print(bench.global_ba_leg(1, 16))
PY
rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python3 /tmp/gba.py > $OUT/run.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $OUT/trace -name "*.db" | head -1) grid > $OUT/kernel_stats.txt
rm -rf $OUT/trace
tail -2 $OUT/run.log
