#!/bin/bash
# Copy the artifacts of one tools/profile_round.sh run (gpurun_out/round_<tag>/) into profiles/ under the round's names; every
# installed file starts with the command that produced it, taken from the cmd_*.txt the profiling script wrote (not retyped here).
#   bash tools/install_profiles.sh <tag> [round-prefix, default r03]
set -eu
TAG=$1; P=${2:-r06}
cd "$(dirname "$0")/.."
SRC=gpurun_out/round_$TAG
for extra in "" _driver_flags _objects16; do
  if [ -s $SRC/bench_line$extra.json ]; then tail -1 $SRC/bench_line$extra.json > profiles/${P}_bench_line$extra.json; fi
done
for name in bench cnn_serial dominant dominant_latency latency slam global_ba; do
  if [ -s $SRC/${name}_kernel_stats.txt ]; then
    { echo "# $(cat $SRC/cmd_$name.txt)   (1x MI355X; tools/profile_round.sh $TAG; per kernel x grid, tools/rocpd_stats.py)"; cat $SRC/${name}_kernel_stats.txt; } > profiles/${P}_${name}_kernel_stats.txt
  fi
done
{ echo "# rocprofv3 --pmc <counter group> --kernel-trace -- python3 tools/bench_dominant.py 20 256   (one pass per counter group; tools/profile_round.sh $TAG)"
  echo "# FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE as reported"
  cat $SRC/pmc.txt; } > profiles/${P}_pmc_dominant_conv.txt
{ echo "# SUO_PMC_GEMM_M=1048576 bash tools/profile_gemm_pmc.sh conv1_f16 conv1_x3 conv1   (rocprofv3 --pmc <group> --kernel-trace -- python3 tools/pmc_gemm.py run <shape> 10, one pass per group; conv1_f16 = gemm_bf16x3_kernel with two fp16 planes, what the network launches; conv1_x3 = the same kernel with three bf16 planes; conv1 = the fp32-pipe kernel)"
  cat $SRC/pmc_gemm.txt; } > profiles/${P}_pmc_gemm.txt
if [ -s $SRC/res_block.txt ]; then { echo "# python3 tools/bench_res_block.py 8; python3 tools/bench_res_block.py 32   (1x MI355X; tools/profile_round.sh $TAG; HIP events, 50 launches each)"; cat $SRC/res_block.txt; } > profiles/${P}_res_block.txt; fi
if [ -s $SRC/stem.txt ]; then { echo "# python3 tools/bench_stem.py 256; python3 tools/bench_stem.py 8   (1x MI355X; tools/profile_round.sh $TAG; HIP events, 30 launches each; it replaces roi_align_concat_kernel<0,4> + convk_kernel<7,2,4,...>: 171 + 864 us at 256 crops, 8 + 29 at 8)"; cat $SRC/stem.txt; } > profiles/${P}_stem.txt; fi
if [ -s $SRC/slam_stages.txt ]; then { echo "# python3 tools/time_slam_stages.py 1   (1x MI355X; tools/profile_round.sh $TAG; bench.py's slam leg configuration, perf_counter around the host methods)"; cat $SRC/slam_stages.txt; } > profiles/${P}_slam_stages.txt; fi
cp $SRC/pmc_dominant_conv.json profiles/pmc_dominant_conv.json
[ -s $SRC/pmc_gemm.json ] && cp $SRC/pmc_gemm.json profiles/pmc_gemm.json
for f in latency.log slam.log; do [ -s $SRC/$f ] && grep -v "amdgpu.ids\|rocprofv3\|simple_timer\|^W2026" $SRC/$f > profiles/${P}_${f%.log}_run.txt || true; done
ls -la profiles/
if [ -s $SRC/f16x2_ab.txt ]; then { echo "# python3 tools/bench_f16x2.py 256   (1x MI355X; tools/profile_round.sh $TAG; HIP events, 30 launches each after 8 warm-up)"; cat $SRC/f16x2_ab.txt; } > profiles/${P}_f16x2_ab.txt; fi
if [ -s $SRC/views_single_host.txt ]; then { echo "# python3 tools/time_views_single_host.py 16   (1x MI355X; tools/profile_round.sh $TAG; cProfile of 8 batches, two in flight)"; cat $SRC/views_single_host.txt; } > profiles/${P}_views_single_host.txt; fi
if [ -s $SRC/frame_chain.txt ]; then { echo "# python3 tools/time_frame_chain.py   (1x MI355X; tools/profile_round.sh $TAG; one frame per call, one in flight)"; cat $SRC/frame_chain.txt; } > profiles/${P}_frame_chain.txt; fi
if [ -s $SRC/chain_head.txt ]; then { echo "# python3 tools/bench_chain_head.py 256; python3 tools/bench_chain_head.py 8   (1x MI355X; tools/profile_round.sh $TAG; HIP events, 20 launches each)"; cat $SRC/chain_head.txt; } > profiles/${P}_chain_head.txt; fi
if [ -s $SRC/network_by_crops.txt ]; then { echo "# python3 bench.py --no-legs --only cnn --depth 1 --objects L --frames-per-step 1 --steps 200 --warmup 20   (1x MI355X; tools/profile_round.sh $TAG; one network call of L crops at a time: H2D of boxes + network + decode + masks)"; cat $SRC/network_by_crops.txt; } > profiles/${P}_network_by_crops.txt; fi
if [ -s $SRC/global_ba_tool.txt ]; then { echo "# python3 tools/bench_global_ba.py 60 8 | 32 16 | 120 8   (1x MI355X; tools/profile_round.sh $TAG; a global adjustment through suo_optimize = the phase kernels driven from C, and through ba_dist.py on one rank)"; cat $SRC/global_ba_tool.txt; } > profiles/${P}_global_ba_tool.txt; fi
