#!/bin/bash
# Per-kernel serial timings of the bench's network calls (no side streams, no graph, one forward in flight):
#   tools/profile_bench_serial.sh [frames_per_forward]
F=${1:-32}
cd /tmp && export TMPDIR=/tmp
export SUO_SERIAL=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/bench_serial_F$F
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-latency-leg --no-global-ba-leg --no-graph --only cnn --depth 1 --frames-per-step $F --steps 6 --warmup 2 > $OUT/run.log 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB grid > $OUT/stats.txt
tail -1 $OUT/run.log | cut -c1-200
