#!/bin/bash
# Per-kernel serial timings of one backbone pass at L crops (no side streams, no graph): tools/profile_serial.sh [L]
L=${1:-32}
cd /tmp && export TMPDIR=/tmp
export SUO_SERIAL=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/serial_L$L
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_backbone.py $L 6 0 > $OUT/run.log 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB grid > $OUT/stats.txt
tail -2 $OUT/run.log
