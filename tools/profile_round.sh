#!/bin/bash
# Round profiles on the GPU box (one gpurun call): bench lines, kernel traces, dominant-kernel stats, PMC passes.
#   bash tools/profile_round.sh <round-tag>      -> gpurun_out/round_<tag>/...   (then: bash tools/install_profiles.sh <tag> <prefix>)
# Every rocprofv3 command is written next to its output (cmd_*.txt) so that the installed files carry the command that made them.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round_$TAG
[ -z "${SUO_PROFILE_ONLY_PMC:-}" ] && rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_traced() {   # name, then the program and its arguments
  local name=$1; shift
  echo "rocprofv3 --kernel-trace -- python3 $*" | sed "s#$R/##g" > $OUT/cmd_$name.txt
  rocprofv3 --kernel-trace -d $OUT/${name}_trace -o trace -- python3 "$@" > $OUT/$name.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $OUT/${name}_trace -name "*.db" | head -1) grid > $OUT/${name}_kernel_stats.txt
  rm -rf $OUT/${name}_trace
}
if [ -z "${SUO_PROFILE_ONLY_PMC:-}" ]; then
# 1. the bench line itself (all legs), and at the driver's flags
python3 $R/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_flags.json 2>> $OUT/bench.err
python3 $R/bench.py --no-cpu-baseline --no-slam-leg --no-latency-leg --objects 16 --frames-per-step 16 > $OUT/bench_line_objects16.json 2>> $OUT/bench.err
# 2. kernel trace of the timed region (two steps in flight: per-kernel durations are concurrent-execution times)
run_traced bench $R/bench.py --no-legs --steps 20 --warmup 5
# 3. the same network calls one at a time (one stream, no graph): non-overlapped per-kernel durations at 256 crops per call
SUO_SERIAL=1 run_traced cnn_serial $R/bench.py --no-legs --no-graph --only cnn --depth 1 --steps 6 --warmup 2
# 4. dominant kernel alone at the bench launch shape (256 crops) and at the latency-mode shape (8 crops)
run_traced dominant $R/tools/bench_dominant.py 100 256
run_traced dominant_latency $R/tools/bench_dominant.py 100 8
# 5. one frame per call, one in flight: every kernel of a frame (network + device geometry chain)
run_traced latency $R/tools/time_frame_chain.py
# 6. one SLAM sequence through ObjectSLAM.process_view (BASELINE configs[2])
run_traced slam $R/tools/profile_slam_view.py
# 6b. the one-launch Residual blocks of the one-frame call shape against the three per-layer launches (HIP events), and the in-kernel phase times
python3 $R/tools/bench_res_block.py 8 2>&1 | grep -v amdgpu.ids > $OUT/res_block.txt
python3 $R/tools/bench_res_block.py 32 2>&1 | grep -v amdgpu.ids >> $OUT/res_block.txt
# 6b'. the fused RoIAlign + stem launch (HIP events), and the stages of a SLAM view (perf_counter around the host methods, no profiler)
python3 $R/tools/bench_stem.py 256 2>&1 | grep -v amdgpu.ids > $OUT/stem.txt
python3 $R/tools/bench_stem.py 8 2>&1 | grep -v amdgpu.ids >> $OUT/stem.txt
python3 $R/tools/time_slam_stages.py 1 2>&1 | grep -v amdgpu.ids > $OUT/slam_stages.txt
# 6c. the multi-GPU bundle adjustment schedule at one rank (bench.py's global_ba leg): kernels of the phase units
echo "rocprofv3 --kernel-trace -- python3 -c 'import bench; bench.global_ba_leg(1, 16)'" > $OUT/cmd_global_ba.txt
printf 'import sys\nsys.path.insert(0, "%s")\nimport bench\nprint(bench.global_ba_leg(1, 16))\n' $R > /tmp/gba.py
rocprofv3 --kernel-trace -d $OUT/global_ba_trace -o trace -- python3 /tmp/gba.py > $OUT/global_ba.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $OUT/global_ba_trace -name "*.db" | head -1) grid > $OUT/global_ba_kernel_stats.txt
rm -rf $OUT/global_ba_trace
fi
# 7. PMC passes (separate runs per counter group, nothing but --kernel-trace beside --pmc)
mkdir -p $R/gpurun_out/pmc && rm -f $R/gpurun_out/pmc/*
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc -o $n -- python3 $R/tools/bench_dominant.py 20 256 > /dev/null 2>&1
done
python3 $R/tools/pmc_to_json.py 256 > $OUT/pmc.txt 2>&1
cp $R/profiles/pmc_dominant_conv.json $OUT/ 2>/dev/null
SUO_PMC_GEMM_M=1048576 bash $R/tools/profile_gemm_pmc.sh conv1_f16 conv1_x3 conv1 > $OUT/pmc_gemm.txt 2>&1
cp $R/profiles/pmc_gemm.json $OUT/ 2>/dev/null
ls $OUT
if [ -z "${SUO_PROFILE_ONLY_PMC:-}" ]; then
# 8. round 5: the matrix-pipe forms side by side at the network's launch shapes, and the host side of the batched evaluation
python3 $R/tools/bench_f16x2.py 256 2>&1 | grep -v amdgpu.ids > $OUT/f16x2_ab.txt
python3 $R/tools/time_views_single_host.py 16 2>&1 | grep -v amdgpu.ids | head -40 > $OUT/views_single_host.txt
python3 $R/tools/time_frame_chain.py 2>&1 | grep -v amdgpu.ids > $OUT/frame_chain.txt
# 9. round 5, late: lin -> head as one launch; a network call of L = 1 .. 9 crops with the round's launch-size thresholds against the former ones
python3 $R/tools/bench_chain_head.py 256 2>&1 | grep -v amdgpu.ids > $OUT/chain_head.txt
python3 $R/tools/bench_chain_head.py 8 2>&1 | grep -v amdgpu.ids >> $OUT/chain_head.txt
ms() { python3 $R/bench.py --no-legs --only cnn --depth 1 --objects $1 --frames-per-step 1 --steps 200 --warmup 20 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for L in 1 2 3 4 5 6 7 8 9; do
  echo "crops per call $L: $(ms $L) ms | four-wave Winograd kernels on small launches too (SUO_WINO_W8=0, rounds 1-5): $(SUO_WINO_W8=0 ms $L) ms"
done > $OUT/network_by_crops.txt
python3 $R/tools/bench_global_ba.py 60 8 2>&1 | grep -v amdgpu.ids > $OUT/global_ba_tool.txt
python3 $R/tools/bench_global_ba.py 32 16 2>&1 | grep -v amdgpu.ids >> $OUT/global_ba_tool.txt
python3 $R/tools/bench_global_ba.py 120 8 2>&1 | grep -v amdgpu.ids >> $OUT/global_ba_tool.txt
fi
# gpurun merges gpurun_out/ back only when it is <= 64 MiB: the raw counter databases stay on the box (their summaries are in $OUT and profiles/*.json)
rm -rf $R/gpurun_out/pmc
find $R/gpurun_out -name "*.db" -delete 2>/dev/null
du -sh $R/gpurun_out
