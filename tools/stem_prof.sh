#!/bin/bash
# Per-kernel durations of the network's first launches, fused stem on / off, 256 crops per call and one frame per call (rocprofv3 --kernel-trace).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/stem_prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  export SUO_STEM_X3=$m
  SUO_SERIAL=1 rocprofv3 --kernel-trace -d $OUT/t$m -o trace -- python3 $R/bench.py --no-legs --no-graph --only cnn --depth 1 --steps 6 --warmup 2 > $OUT/serial$m.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $OUT/t$m -name "*.db" | head -1) grid | grep -i "stem\|roi_align\|convk\|Total\|total" > $OUT/serial_stem$m.txt
  rm -rf $OUT/t$m
  rocprofv3 --kernel-trace -d $OUT/l$m -o trace -- python3 $R/tools/time_frame_chain.py > $OUT/lat$m.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $OUT/l$m -name "*.db" | head -1) grid | grep -i "stem\|roi_align\|convk" > $OUT/lat_stem$m.txt
  rm -rf $OUT/l$m
done
tail -n 20 $OUT/*_stem*.txt
