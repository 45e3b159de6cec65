#!/bin/bash
# Copy the artifacts of one tools/profile_round.sh run (gpurun_out/round_<tag>/, optionally gpurun_out/bench_serial_F32/)
# into profiles/ under the round's names, each with a header saying which command produced it.
#   bash tools/install_profiles.sh <tag> [round-prefix, default r01]
set -eu
TAG=$1; P=${2:-r02}
cd "$(dirname "$0")/.."
SRC=gpurun_out/round_$TAG
tail -1 $SRC/bench_line.json > profiles/${P}_bench_line.json
for extra in driver_flags objects16; do
  if [ -s $SRC/bench_line_$extra.json ]; then tail -1 $SRC/bench_line_$extra.json > profiles/${P}_bench_line_$extra.json; fi
done
{
  echo "# rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-latency-leg --steps 20 --warmup 5   (MI355X; tools/profile_round.sh $TAG)"
  echo "# one step = 32 frames (256 crops) per network call + their PnP / LM, 2 steps in flight; kernels of the two calls overlap,"
  echo "# so per-kernel durations here are concurrent-execution times (summarised per kernel x grid with tools/rocpd_stats.py)."
  echo "# The dominant kernel's 8192-workgroup row mixes the steps' launches (concurrent, longer) with the 41 isolated launches of bench.py's"
  echo "# roofline loop (its min_us column is the isolated duration); the isolated trace is r02_dominant_kernel_stats.txt."
  cat $SRC/bench_kernel_stats.txt
} > profiles/${P}_bench_kernel_stats_final.txt
{
  echo "# rocprofv3 --kernel-trace -- python3 tools/bench_dominant.py 100 256   (the dominant kernel (Winograd fused tail), the Winograd 3x3 alone and the two direct-form kernels at the bench launch shape: 256 crops; tools/profile_round.sh $TAG)"
  cat $SRC/dominant_kernel_stats.txt
} > profiles/${P}_dominant_kernel_stats.txt
{
  echo "# rocprofv3 --pmc <counter group> --kernel-trace -- python3 tools/bench_dominant.py 20 256   (tools/profile_round.sh $TAG; one pass per counter group)"
  echo "# dominant kernel: fused Residual tail wino3x3_kernel<true> (3x3 128->128 in Winograd F(2x2,3x3) form + ReLU, 1x1 128->256 + skip) @64x64, 256 crops per launch"
  echo "# FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE as reported"
  cat $SRC/pmc.txt
} > profiles/${P}_pmc_dominant_conv.txt
cp $SRC/pmc_dominant_conv.json profiles/pmc_dominant_conv.json
if [ -f gpurun_out/bench_serial_F32/stats.txt ]; then
  {
    echo "# SUO_SERIAL=1 rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-graph --only cnn --depth 1 --frames-per-step 32 --steps 6 --warmup 2   (tools/profile_bench_serial.sh)"
    echo "# one network call at a time, one stream, no graph: per-kernel NON-overlapped durations at 256 crops per call (8 calls;"
    echo "# the dominant-kernel row also contains the launches of bench.py's roofline loop)."
    cat gpurun_out/bench_serial_F32/stats.txt
  } > profiles/${P}_cnn_serial_kernel_stats.txt
fi
ls -la profiles/
