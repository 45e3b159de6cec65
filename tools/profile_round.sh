#!/bin/bash
# Round profiles on the GPU box (one gpurun call): bench line, bench kernel stats, dominant-kernel stats, PMC passes.
#   bash tools/profile_round.sh <round-tag>      -> gpurun_out/round_<tag>/...
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line itself (with the CPU baseline)
python3 $R/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
# 2. kernel trace of the same command (shorter run)
rocprofv3 --kernel-trace -d $OUT/bench_trace -o trace -- python3 $R/bench.py --no-cpu-baseline --no-latency-leg --steps 20 --warmup 5 > $OUT/bench_trace.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $OUT/bench_trace -name "*.db" | head -1) grid > $OUT/bench_kernel_stats.txt
# 3. dominant kernel alone at the bench launch shape (256 crops)
rocprofv3 --kernel-trace -d $OUT/dom_trace -o trace -- python3 $R/tools/bench_dominant.py 100 256 > $OUT/dom.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $OUT/dom_trace -name "*.db" | head -1) grid > $OUT/dominant_kernel_stats.txt
# 4. PMC passes (separate runs per counter group, no other tracing)
mkdir -p $R/gpurun_out/pmc && rm -f $R/gpurun_out/pmc/*
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc -o $n -- python3 $R/tools/bench_dominant.py 20 256 > /dev/null 2>&1
done
python3 $R/tools/pmc_to_json.py 256 > $OUT/pmc.txt 2>&1
cp $R/profiles/pmc_dominant_conv.json $OUT/ 2>/dev/null
rm -rf $OUT/bench_trace $OUT/dom_trace
ls $OUT
